#!/bin/bash
# bench.py's default workload (256 spp per batch) at other grid multipliers of the per-stage pipeline, interleaved.
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['spp_per_step'], 'spp/step x', d['steps'], d['value'], 'Mray/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel_ms'])"; }
for g in ${1:-8 16 8 16 32 20}; do CRT_GRID_MULT=$g python bench.py --no-cpu-baseline --warmup 1 2>>gpurun_out/ab_stderr.log | show grid_x$g; done
