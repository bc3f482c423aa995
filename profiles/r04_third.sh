#!/bin/bash
# Round 4, third measurement call: GPU suite on the row-swap + wide-arena build, then A/B: node-loop row swap on / off on
# four workloads, four-wave kernels with the large-tree split against the three-wave default on the large flat trees.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04d}
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
grep -q " failed" gpurun_out/${T}_tests.log && exit 1
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "noswap cur" --scene cornellbox --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "noswap cur" --scene veach_mis --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "noswap cur noswap@CRT_FUSED=0,CRT_WIDE=1 cur@CRT_FUSED=0,CRT_WIDE=1" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "noswap cur" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "noswap cur cur@CRT_FUSED=0,CRT_WIDE=1" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "noswap cur" --scene openpbr_showcase --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo third done
