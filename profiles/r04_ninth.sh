#!/bin/bash
# Round 4, ninth call: traversal parity subset on the branchless-push build, then base / cur on five workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04j}
timeout -k 10 600 python -m pytest tests/test_gpu_traverse.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
grep -q " failed" gpurun_out/${T}_tests.log && exit 1
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "base cur" --scene cornellbox --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "base cur" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "base cur" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "base cur" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "base cur" --scene veach_mis --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo ninth done
