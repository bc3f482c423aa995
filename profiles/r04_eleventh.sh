#!/bin/bash
# Round 4, eleventh call: the four-wave kernels' direct-engine instances (CRT_WIDE=2) — parity on the direct-leaf scenes, then A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04l}
CRT_WIDE=2 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1 timeout -k 10 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "image_and_counters or medcity or synthetic_city or random_world or lights_at_infinity" > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
grep -q " failed" gpurun_out/${T}_tests.log && exit 1
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "cur cur@CRT_WIDE=2" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_WIDE=2" --scene openpbr_showcase --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_WIDE=2" --scene synthetic:city:181 --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo eleventh done
