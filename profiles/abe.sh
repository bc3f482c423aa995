# usage: bash profiles/abe.sh "<bench args>" name[:FUSED] ...  — variants with CRT_FUSED per entry (default 1)
ARGS="$1"; shift
for round in 1 2; do
  for e in "$@"; do
    v=${e%%:*}; f=${e#*:}; [ "$f" = "$e" ] && f=1
    if [ "$v" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$v.so; fi
    CRT_FUSED=$f timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline 2>>gpurun_out/ab_stderr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
