#!/bin/bash
# Round-3 evidence on the shipped build, in three calls of < 20 minutes each (gpurun's limit):
#   bash profiles/evidence_r03.sh <tag> 1   GPU tests; PMC passes of the bench scene and the stress scene -> r03_pmc_bench.json
#                                           (written on the box, on the profiled build); the two bench lines that carry it
#   bash profiles/evidence_r03.sh <tag> 2   PMC passes of veach_mis, openpbr_showcase, MedCity 4K (merged into the same
#                                           file); their bench lines; rocprofv3 --kernel-trace --stats of bench.py and of the stress line
#   bash profiles/evidence_r03.sh <tag> 3   kernel probe (+ the 7 M-triangle scene and its PMC passes), published renders,
#                                           phase utilisation, per-bounce work
# Part 2 needs part 1's r03_pmc_bench.json of the SAME build in profiles/ (copy gpurun_out/<tag>_pmc_bench.json there first).
T=$1; PART=$2; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
stats() { # name, bench args...
  local n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats_$n -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${T}_stats_$n.log 2>&1)
  cp $R/gpurun_out/${T}_stats_$n/*/*kernel_stats.csv $R/gpurun_out/${T}_rocprofv3_kernel_stats_$n.csv
}
case $PART in
1)
  timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_tests.log
  rm -f profiles/r03_pmc_bench.json
  bash profiles/pmc_r03.sh $T cb stress
  python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench.err
  python bench.py --scene stress --spp-per-step 256 > gpurun_out/${T}_bench_stress.json 2>> gpurun_out/${T}_bench.err
  ;;
2)
  bash profiles/pmc_r03.sh $T veach showcase mc
  : > gpurun_out/${T}_bench_other_configs.json
  for sc in veach_mis openpbr_showcase; do
    python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline >> gpurun_out/${T}_bench_other_configs.json 2>> gpurun_out/${T}_bench.err
  done
  for sc in cornellbox_guided sun_sky; do
    python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline >> gpurun_out/${T}_bench_other_configs.json 2>> gpurun_out/${T}_bench.err
  done
  python bench.py --scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 128 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_medcity_3840x2160.json 2>> gpurun_out/${T}_bench.err
  stats bench_default
  stats stress --scene stress --spp-per-step 256 --steps 2 --warmup 1
  ;;
3)
  python bench_kernels.py > gpurun_out/${T}_kernel_probe.json 2> gpurun_out/${T}_kernel_probe.err
  python bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 4 > gpurun_out/${T}_kernel_probe_big.json 2>> gpurun_out/${T}_kernel_probe.err
  bash profiles/run_pmc_r02.sh ${T}big bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 2 > /dev/null
  python - <<PY > gpurun_out/${T}_pmc_kernel_probe_big.json
import json, subprocess, sys
d = json.load(open("gpurun_out/${T}_kernel_probe_big.json"))["scenes"]["tri_spheres_big"]
print(subprocess.run([sys.executable, "profiles/summarize_pmc_big.py", "${T}big", str(d["intersect"]["gpu_ms_per_launch"]), str(d["occluded"]["gpu_ms_per_launch"])], capture_output=True, text=True).stdout)
PY
  python bench_published.py > gpurun_out/${T}_published_default_renders.json 2> gpurun_out/${T}_published.err
  python profiles/phase_utilisation.py scenes/PointInstancedMedCity.usd > gpurun_out/${T}_phase_medcity.txt 2>&1
  python profiles/per_bounce.py work --spp 16 --max 4 > gpurun_out/${T}_per_bounce_work_cornellbox.txt 2>&1
  python profiles/per_bounce.py work --scene stress --spp 8 --max 3 > gpurun_out/${T}_per_bounce_work_stress.txt 2>&1
  ;;
esac
echo "evidence part $PART done"
