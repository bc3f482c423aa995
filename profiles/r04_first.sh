#!/bin/bash
# Round 4, first measurement call on one box: the GPU suite, the bench line with its new other_configs legs, and the
# pipeline A/B on the large-tree workloads (fused against per-stage three-wave with lanes) incl. the synthetic 7 M-triangle scene.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04b}
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
grep -q " failed" gpurun_out/${T}_tests.log && exit 1
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "cur cur@CRT_FUSED=0" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_FUSED=0" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_FUSED=0 cur@CRT_FUSED=0,CRT_LANES=1" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
