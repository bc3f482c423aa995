#!/bin/bash
# A/B for the per-stage pipeline with 4-wave traversal kernels: variants x grid multipliers on three flat scenes.
run() { v=$1; shift; lib=${v%%@*}; envs=$(echo "${v#*@}" | tr ',' ' '); [ "$lib" = "$v" ] && envs=""
  if [ "$lib" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$lib.so; fi
  env $envs timeout -k 10 240 python profiles/quick_bench.py --tag $v "$@" 2>>gpurun_out/ab_stderr.log || { echo "$v FAILED $* (stderr kept in gpurun_out/ab_stderr.log); stopping: no further GPU step after a failed one"; tail -5 gpurun_out/ab_stderr.log; exit 1; }; }
for round in 1 2; do for v in $1; do
  run $v --scene cornellbox --steps 6; run $v --scene veach_mis --steps 4; run $v --scene sun_sky --steps 4
done; done
