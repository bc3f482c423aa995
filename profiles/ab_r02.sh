#!/bin/bash
# Interleaved A/B of variant builds on ONE box, two rounds (drift shows up as disagreement between rounds).
#   bash profiles/ab_r02.sh "cur v1 v2" [scene-set]      "cur" = the in-tree build, others = variants/<name>.so
#   a variant may carry env settings: "cur@CRT_PARTITION=0"
# scene-set: all (default) = cornellbox 1080p, veach_mis 1080p, openpbr_showcase 1080p, MedCity 4K; mc = MedCity 4K only;
#            fast = cornellbox + MedCity 1080p
VARS="$1"; SET="${2:-all}"
run() { # tag, args...
  local v=$1; shift
  local lib=${v%%@*} envs=""
  if [ "$lib" != "$v" ]; then envs=$(echo "${v#*@}" | tr ',' ' '); fi   # name@VAR=1,VAR2=x : env settings for this variant
  if [ "$lib" = cur ]; then unset CRT_AMD_LIB; else export CRT_AMD_LIB=$PWD/variants/$lib.so; fi
  env $envs timeout -k 10 240 python profiles/quick_bench.py --tag $v "$@" 2>>gpurun_out/ab_stderr.log || { echo "$v FAILED $* (stderr kept in gpurun_out/ab_stderr.log); stopping: no further GPU step after a failed one"; tail -5 gpurun_out/ab_stderr.log; exit 1; }
}
for round in 1 2; do
  for v in $VARS; do
    case $SET in
      mc) run $v --scene PointInstancedMedCity --width 3840 --height 2160 --steps 2 ;;
      fast) run $v --scene cornellbox --steps 4; run $v --scene PointInstancedMedCity --steps 4 ;;
      cbmc) run $v --scene cornellbox --steps 6; run $v --scene PointInstancedMedCity --width 3840 --height 2160 --steps 2 ;;
      lit) run $v --scene veach_mis --steps 4; run $v --scene openpbr_showcase --steps 4; run $v --scene cornellbox_guided --steps 4; run $v --scene sun_sky --steps 4 ;;
      *) run $v --scene cornellbox --steps 6; run $v --scene veach_mis --steps 4; run $v --scene openpbr_showcase --steps 4
         run $v --scene PointInstancedMedCity --width 3840 --height 2160 --steps 2 ;;
    esac
  done
done
