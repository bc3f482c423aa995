# usage: bash profiles/abn.sh "<bench args>" name1 name2 ...   — interleaved A/B of variant builds on one box.
# "cur" = the in-tree build; any other name = variants/<name>.so. Two rounds, so drift shows up as disagreement.
ARGS="$1"; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$v.so; fi
    timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline 2>>gpurun_out/ab_stderr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['bytes_per_ray'])"
  done
done
