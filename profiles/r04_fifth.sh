#!/bin/bash
# Round 4, fifth call: traversal parity subset on the unrolled-spill build, then old / new deep-stack node step across splits.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04f}
timeout -k 10 600 python -m pytest tests/test_gpu_traverse.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
grep -q " failed" gpurun_out/${T}_tests.log && exit 1
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "oldspill cur" --scene cornellbox --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "oldspill cur oldspill@CRT_FUSED=0,CRT_WIDE=1 cur@CRT_FUSED=0,CRT_WIDE=1 oldspill@CRT_POOL_STACK_RT=6 cur@CRT_POOL_STACK_RT=6" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "oldspill cur cur@CRT_POOL_STACK_RT=6 cur@CRT_DIRECT_LEAVES=0,CRT_WIDE=1,CRT_POOL_STACK_RT=10 cur@CRT_DIRECT_LEAVES=0,CRT_WIDE=0" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "oldspill cur cur@CRT_FUSED=0,CRT_WIDE=1 cur@CRT_POOL_STACK_RT=6" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo fifth done
