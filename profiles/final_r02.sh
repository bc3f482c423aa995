#!/bin/bash
# Final evidence of the round on one box: GPU tests, PMC passes of the two bench workloads, the profile bench.py reads
# (written HERE, on the build that was profiled), then the bench line that carries it, the kernel stats and the
# published-settings renders.   bash profiles/final_r02.sh <tag>
T=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
bash profiles/run_pmc_r02.sh ${T}cb bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null
bash profiles/run_pmc_r02.sh ${T}mc bench.py --no-cpu-baseline --scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 64 --steps 2 --warmup 1 > /dev/null
python profiles/summarize_pmc_bench.py "cornellbox 1920x1080 256spp=${T}cb" "PointInstancedMedCity 3840x2160 64spp=${T}mc" > /dev/null
cp profiles/r02_pmc_bench.json gpurun_out/${T}_pmc_bench.json
python profiles/summarize_pmc.py ${T}cb > gpurun_out/${T}_pmc_cornellbox.json
python profiles/summarize_pmc.py ${T}mc > gpurun_out/${T}_pmc_medcity.json
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
python bench.py --scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_medcity_3840x2160.json 2>> gpurun_out/${T}_bench_default.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${T}_stats.log 2>&1)
python bench_published.py > gpurun_out/${T}_published_default_renders.json 2> gpurun_out/${T}_published.err
echo final done
