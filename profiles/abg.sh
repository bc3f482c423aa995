# usage: bash profiles/abg.sh "<bench args>" name[:GRID_MULT] ...  — like abn.sh, with an optional CRT_GRID_MULT per entry
ARGS="$1"; shift
for round in 1 2; do
  for e in "$@"; do
    v=${e%%:*}; g=${e#*:}; [ "$g" = "$e" ] && g=3
    if [ "$v" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$v.so; fi
    CRT_GRID_MULT=$g timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline 2>>gpurun_out/ab_stderr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
