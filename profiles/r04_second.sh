#!/bin/bash
# Round 4, second measurement call: the fourth wave on large / instanced trees gauged with existing kernels (flat engine,
# direct forms off), the new per-stage default, and the first PMC passes of the synthetic 7 M-triangle scene in the integrator.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04c}
: > gpurun_out/${T}_ab.txt
MC="--scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2"
bash profiles/ab_env.sh "cur cur@CRT_FUSED=1 cur@CRT_DIRECT_LEAVES=0,CRT_FUSED=0,CRT_WIDE=0 cur@CRT_DIRECT_LEAVES=0,CRT_FUSED=0,CRT_WIDE=1,CRT_POOL_STACK_RT=6 cur@CRT_DIRECT_LEAVES=0,CRT_FUSED=0,CRT_WIDE=0,CRT_POOL_STACK_RT=6" $MC >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_FUSED=1 cur@CRT_FUSED=0,CRT_WIDE=1 cur@CRT_FUSED=0,CRT_POOL_STACK_RT=6" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_FUSED=0,CRT_WIDE=1" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
export CRT_LANES=1
PASSES="1 3 4" bash profiles/run_pmc_r04.sh ${T}big bench.py --no-cpu-baseline --no-other-configs --scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1
python profiles/summarize_pmc.py ${T}big > gpurun_out/${T}_pmc_big.json
unset CRT_LANES
python bench.py --no-cpu-baseline --no-other-configs --scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1 > gpurun_out/${T}_bench_big.json 2> gpurun_out/${T}_bench_big.err
python - <<PY
import json
d = json.load(open("gpurun_out/${T}_pmc_big.json"))
for k, e in d.items():
    if e.get("SQ_WAVE_CYCLES", 0) > 0 and k.startswith(("k_extend", "k_shadow", "k_shade", "k_path")):
        print(k, "launches", e["launches"], "read GB/launch %.2f write %.2f l2_hit %.3f" % (e.get("ea_dram_read_bytes_per_launch", 0) / 1e9, e.get("ea_dram_write_bytes_per_launch", 0) / 1e9, e.get("l2_hit_rate", 0)))
PY
echo second done
