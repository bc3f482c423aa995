# usage: bash profiles/ab3.sh "<bench args>"  — current build vs variants/libcrt_head.so, two interleaved rounds
for round in 1 2; do
timeout -k 10 200 python bench.py $1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
CRT_AMD_LIB=$PWD/variants/libcrt_head.so timeout -k 10 200 python bench.py $1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HEAD', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
