#!/bin/bash
# Round 4, sixth call: render parity subset on the new engine choice, the stress / synthetic bench lines with their stage
# split, PMC passes (wave time, DRAM reads + L2 hit, DRAM writes) of the synthetic 7 M-triangle scene in the integrator.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04g}
timeout -k 10 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_stress.py -m gpu -x -q -p no:cacheprovider -k "not knobs and not partition" > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_tests.log
grep -q " failed" gpurun_out/${T}_tests.log && exit 1
python bench.py --no-cpu-baseline --no-other-configs --scene stress --spp-per-step 256 --steps 2 --warmup 1 > gpurun_out/${T}_bench_stress.json 2> gpurun_out/${T}_bench.err
python bench.py --no-cpu-baseline --no-other-configs --scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1 > gpurun_out/${T}_bench_big.json 2>> gpurun_out/${T}_bench.err
export CRT_LANES=1
PASSES="1 3 4" bash profiles/run_pmc_r04.sh ${T}big bench.py --no-cpu-baseline --no-other-configs --scene synthetic:big --spp-per-step 128 --steps 2 --warmup 1
python profiles/summarize_pmc.py ${T}big > gpurun_out/${T}_pmc_big.json
unset CRT_LANES
python - <<PY
import json
for n in ("stress", "big"):
    d = json.load(open("gpurun_out/${T}_bench_%s.json" % n)); r = d["roofline"]
    print(n, d["value"], d["ms_per_step"], r["pipeline"], "lanes", r["lanes"], "serial ms/step", r.get("serial_kernel_ms_per_step"), "timed", r["kernel_ms"])
d = json.load(open("gpurun_out/${T}_pmc_big.json"))
for k, e in d.items():
    if e.get("SQ_WAVE_CYCLES", 0) > 0 and k.startswith(("k_extend", "k_shadow", "k_shade", "k_path")):
        print(k, "launches", e["launches"], "read GB/launch %.2f write %.2f l2_hit %.3f" % (e.get("ea_dram_read_bytes_per_launch", 0) / 1e9, e.get("ea_dram_write_bytes_per_launch", 0) / 1e9, e.get("l2_hit_rate", 0)))
PY
echo sixth done
