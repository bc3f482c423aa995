#!/bin/bash
# A/B of variant libraries on the kernel probe incl. the 7 M-triangle scene: bash profiles/ab_probe.sh "base cur"
for round in 1 2; do
  for v in $1; do
    if [ "$v" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$v.so; fi
    timeout -k 10 300 python bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 4 2>>gpurun_out/ab_stderr.log | python -c "
import json,sys; d=json.load(sys.stdin)
print('$v', ' '.join('%s %.0f/%.0f' % (k, e['intersect']['gpu_mray_s'], e['occluded']['gpu_mray_s']) for k, e in d['scenes'].items()))"
  done
done
