#!/bin/bash
# Round 4, seventh call: parity subset on the packet-free-kernel build; packet-free instance on / off on the sphere scene;
# the scheduler's thresholds re-swept on a large tree; the brightened synthetic scene's per-launch HBM traffic.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04h}
timeout -k 10 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_fuzz.py tests/test_gpu_traverse.py tests/test_golden.py tests/test_gpu_determinism.py -m gpu -x -q -p no:cacheprovider -k "not knobs and not partition" > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
grep -q " failed" gpurun_out/${T}_tests.log && exit 1
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "nopk0 cur cur@CRT_FUSED=0" --scene openpbr_showcase --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur sticky16 sticky32 rare16 rare32 fetch32 fetch48" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
export CRT_LANES=1
PASSES="3 4" bash profiles/run_pmc_r04.sh ${T}big bench.py --no-cpu-baseline --no-other-configs --scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1
unset CRT_LANES
python profiles/pmc_per_launch.py ${T}big k_extend k_shadow > gpurun_out/${T}_pmc_per_launch_big.json
python - <<PY
import json
d = json.load(open("gpurun_out/${T}_pmc_per_launch_big.json"))
for k, v in d.items():
    print(k, "all launches:", v["all_launches_hbm_frac"], [(l["ms"], l["hbm_frac"]) for l in v["launches"][:7]])
PY
echo seventh done
