for round in 1 2; do
for m in 4 8 16; do
CRT_GRID_MULT=$m timeout -k 10 200 python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seg mult $m', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
CRT_AMD_LIB=$PWD/variants/libcrt_head.so timeout -k 10 200 python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HEAD(global atomics)', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
