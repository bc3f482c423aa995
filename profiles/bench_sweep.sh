#!/bin/bash
# bench.py's workload at other batch sizes (same 1024 spp in all) and grid multipliers: one line each.
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['spp_per_step'], 'spp/step x', d['steps'], d['value'], 'Mray/s', d['ms_per_step'], 'ms/step', d['config']['seconds_to_target_spp'], 's to 1024 spp', d['roofline']['kernel_ms'])"; }
for s in 128 192 256; do python bench.py --no-cpu-baseline --spp-per-step $s --steps $((1024 / s)) --warmup 1 2>>gpurun_out/ab_stderr.log | show batch; done
for g in 6 10 12; do CRT_GRID_MULT=$g python bench.py --no-cpu-baseline --steps 8 --warmup 1 2>>gpurun_out/ab_stderr.log | show grid_x$g; done
