#!/bin/bash
# bench.py's default workload through variant libraries, interleaved:  bash profiles/ab_bench.sh "base cur" [bench args]
V="$1"; shift
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'Mray/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel_ms'])"; }
for round in 1 2 3; do for v in $V; do
  if [ "$v" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$v.so; fi
  python bench.py --no-cpu-baseline --warmup 1 "$@" 2>>gpurun_out/ab_stderr.log | show $v
done; done
