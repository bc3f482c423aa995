#!/bin/bash
# Round-3 PMC collection on one box: the seven separate passes of profiles/run_pmc_r02.sh for each workload named,
# then the per-kernel summary bench.py reads (profiles/r03_pmc_bench.json, written HERE on the profiled build).
#   bash profiles/pmc_r03.sh <tag> stress|cb|veach|showcase|mc ...
T=$1; shift; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export CRT_LANES=1   # whole-batch launches: per-launch counters are then one batch's launch (bench.py scales them per lane)
KEYS=()
for W in "$@"; do
  case $W in
    stress)   A="--scene stress --spp-per-step 256 --steps 2 --warmup 1"; K="stress 1920x1080 256spp";;
    cb)       A="--steps 2 --warmup 1"; K="cornellbox 1920x1080 512spp";;
    veach)    A="--scene veach_mis --steps 2 --warmup 1"; K="veach_mis 1920x1080 512spp";;
    showcase) A="--scene openpbr_showcase --steps 2 --warmup 1"; K="openpbr_showcase 1920x1080 512spp";;
    mc)       A="--scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 128 --steps 2 --warmup 1"; K="PointInstancedMedCity 3840x2160 128spp";;
  esac
  bash profiles/run_pmc_r02.sh ${T}$W bench.py --no-cpu-baseline $A > gpurun_out/${T}${W}_passes.log 2>&1
  python profiles/summarize_pmc.py ${T}$W > gpurun_out/${T}_pmc_$W.json
  KEYS+=("$K=${T}$W")
  echo "$W: passes done"
done
python profiles/summarize_pmc_bench.py --merge "${KEYS[@]}" > /dev/null
cp profiles/r03_pmc_bench.json gpurun_out/${T}_pmc_bench.json
echo pmc done
