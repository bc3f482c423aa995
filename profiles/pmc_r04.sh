#!/bin/bash
# Round-4 PMC collection on one box: separate rocprofv3 --kernel-trace --pmc passes (profiles/run_pmc_r04.sh; passes 1 3 4 7:
# wave time + VALU, DRAM reads + L2 hit, DRAM writes, instruction mix) for each workload named, then the per-kernel summary
# bench.py reads (profiles/r04_pmc_bench.json, written HERE on the profiled build).
#   bash profiles/pmc_r04.sh <tag> cb|stress|veach|showcase|mc|big ...
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
    big)      A="--scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1"; K="synthetic:big 1920x1080 128spp";;
  esac
  PASSES="1 3 4 7" bash profiles/run_pmc_r04.sh ${T}$W bench.py --no-cpu-baseline --no-other-configs $A > gpurun_out/${T}${W}_passes.log 2>&1
  python profiles/summarize_pmc.py ${T}$W > gpurun_out/${T}_pmc_$W.json
  KEYS+=("$K=${T}$W")
  echo "$W: passes done"
done
python profiles/summarize_pmc_bench.py --merge "${KEYS[@]}" > /dev/null
cp profiles/r04_pmc_bench.json gpurun_out/${T}_pmc_bench.json
echo pmc done
