#!/bin/bash
# Round-2 evidence run (MI355X box): bench lines, kernel probe, PMC passes, phase and class profiles.
#   bash profiles/evidence_r02.sh <tag>        e.g. r02m  -> gpurun_out/<tag>_*
set -x
T=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
for sc in veach_mis openpbr_showcase cornellbox_guided sun_sky; do
  python bench.py --scene $sc --steps 4 --warmup 1 --no-cpu-baseline >> gpurun_out/${T}_bench_other_configs.json 2>> gpurun_out/${T}_bench_other.err
done
python bench.py --scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_medcity_3840x2160.json 2>> gpurun_out/${T}_bench_other.err
python bench_kernels.py > gpurun_out/${T}_kernel_probe.json 2> gpurun_out/${T}_kernel_probe.err
python bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 4 > gpurun_out/${T}_kernel_probe_big.json 2>> gpurun_out/${T}_kernel_probe.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${T}_stats.log 2>&1)
bash profiles/run_pmc_r02.sh ${T}cb bench.py --no-cpu-baseline --steps 4 --warmup 1
bash profiles/run_pmc_r02.sh ${T}mc bench.py --no-cpu-baseline --scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 64 --steps 2 --warmup 1
bash profiles/run_pmc_r02.sh ${T}big bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 2
python profiles/summarize_pmc.py ${T}cb > gpurun_out/${T}_pmc_cornellbox.json
python profiles/summarize_pmc.py ${T}mc > gpurun_out/${T}_pmc_medcity.json
python profiles/summarize_pmc.py ${T}big > gpurun_out/${T}_pmc_kernel_probe_big.json
python profiles/phase_utilisation.py scenes/cornellbox.usda > gpurun_out/${T}_phase_cornellbox.txt 2>&1
python profiles/phase_utilisation.py scenes/PointInstancedMedCity.usd > gpurun_out/${T}_phase_medcity.txt 2>&1
python profiles/shade_classes.py > gpurun_out/${T}_shade_classes.json 2> gpurun_out/${T}_shade_classes.err
python bench_published.py > gpurun_out/${T}_published_default_renders.json 2> gpurun_out/${T}_published.err
echo evidence done
