#!/bin/bash
# Round-4 evidence on the shipped build, in calls of < 20 minutes each (gpurun's limit):
#   bash profiles/evidence_r04.sh <tag> 1   GPU tests; PMC passes of the bench scene and the stress scene -> r04_pmc_bench.json (written on
#                                           the box, on the profiled build)
#   bash profiles/evidence_r04.sh <tag> 2   PMC passes of the synthetic 7 M-triangle scene (+ its per-launch table), veach_mis, openpbr_showcase
#   bash profiles/evidence_r04.sh <tag> 3   PMC passes of MedCity 4K; the bench line (headline + other_configs + both CPU baselines); the
#                                           synthetic scene's and the stress scene's lines
#   bash profiles/evidence_r04.sh <tag> 4   rocprofv3 --kernel-trace --stats of bench.py; kernel probe (+ the 7 M-triangle scene);
#                                           published renders; per-bounce work; the N = 4 gloo rehearsal at configs 4 and 5's shapes
# Parts 2 and 3 merge into the r04_pmc_bench.json of the SAME build: copy gpurun_out/<tag>_pmc_bench.json to profiles/ between calls.
T=$1; PART=$2; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
stats() { # name, bench args...
  local n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats_$n -- python3 $R/bench.py --no-cpu-baseline --no-other-configs "$@" > $R/gpurun_out/${T}_stats_$n.log 2>&1)
  cp $R/gpurun_out/${T}_stats_$n/*/*kernel_stats.csv $R/gpurun_out/${T}_rocprofv3_kernel_stats_$n.csv
}
case $PART in
1)
  timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_tests.log
  rm -f profiles/r04_pmc_bench.json
  bash profiles/pmc_r04.sh $T cb stress
  ;;
1p) # the PMC passes of part 1 alone (after a change that leaves the tests' outcome as it was, e.g. comments in the kernel sources)
  rm -f profiles/r04_pmc_bench.json
  bash profiles/pmc_r04.sh $T cb stress
  ;;
2)
  bash profiles/pmc_r04.sh $T big veach showcase
  python profiles/pmc_per_launch.py ${T}big k_extend k_shadow > gpurun_out/${T}_pmc_per_launch_big.json
  ;;
3)
  bash profiles/pmc_r04.sh $T mc
  python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench.err
  python bench.py --no-cpu-baseline --no-other-configs --scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1 > gpurun_out/${T}_bench_big.json 2>> gpurun_out/${T}_bench.err
  python bench.py --no-other-configs --scene stress --spp-per-step 256 --steps 2 --warmup 1 > gpurun_out/${T}_bench_stress.json 2>> gpurun_out/${T}_bench.err
  ;;
4)
  stats bench_default
  stats big --scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1
  python bench_kernels.py > gpurun_out/${T}_kernel_probe.json 2> gpurun_out/${T}_kernel_probe.err
  python bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 4 > gpurun_out/${T}_kernel_probe_big.json 2>> gpurun_out/${T}_kernel_probe.err
  python bench_published.py > gpurun_out/${T}_published_default_renders.json 2> gpurun_out/${T}_published.err
  python profiles/phase_utilisation.py scenes/PointInstancedMedCity.usd > gpurun_out/${T}_phase_medcity.txt 2>&1
  python profiles/per_bounce.py work --spp 16 --max 4 > gpurun_out/${T}_per_bounce_work_cornellbox.txt 2>&1
  # N = 4 over gloo on this box's one GPU (at most 6 processes may use it): configs 4 and 5's own shapes, the batch sized to the HBM the four ranks share
  python bench.py --gpus 4 --backend gloo --scene veach_mis --depth 8 --spp-per-step 128 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_rehearsal_veach_n4.json 2> gpurun_out/${T}_rehearsal.err
  python bench.py --gpus 4 --backend gloo --scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_rehearsal_medcity_n4.json 2>> gpurun_out/${T}_rehearsal.err
  ;;
esac
echo "evidence part $PART done"
