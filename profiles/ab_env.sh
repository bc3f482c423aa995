#!/bin/bash
# Interleaved A/B of variant libraries with per-variant environment, any quick_bench arguments:
#   bash profiles/ab_env.sh "base cur v1@CRT_WIDE=0,CRT_FUSED=0" --scene cornellbox --spp 256 --steps 2
# "cur" = the in-tree build, others = variants/<name>.so; two interleaved rounds; stderr kept in gpurun_out/ab_stderr.log.
VARS="$1"; shift
for round in 1 2; do for v in $VARS; do
  lib=${v%%@*}; envs=""
  if [ "$lib" != "$v" ]; then envs=$(echo "${v#*@}" | tr ',' ' '); fi
  if [ "$lib" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$lib.so; fi
  env $envs timeout -k 10 300 python profiles/quick_bench.py --tag $v "$@" 2>>gpurun_out/ab_stderr.log || { echo "$v FAILED (gpurun_out/ab_stderr.log); stopping"; tail -5 gpurun_out/ab_stderr.log; exit 1; }
done; done
