#!/bin/bash
# The fused pipeline's grid at large batches: openpbr_showcase 1080p x 256 spp (531 M paths) by workgroups per CU.
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['workload'][:28], d['config']['spp_per_step'], 'spp/step', d['value'], 'Mray/s', d['roofline']['pipeline'][:9])"; }
for r in 1 2; do for g in 3 6 12 24; do CRT_GRID_MULT=$g python bench.py --no-cpu-baseline --scene openpbr_showcase --steps 2 --warmup 1 2>>gpurun_out/ab_stderr.log | show grid_x$g; done; done
for g in 3 6 12; do CRT_GRID_MULT=$g python bench.py --no-cpu-baseline --scene openpbr_showcase --spp-per-step 64 --steps 4 --warmup 1 2>>gpurun_out/ab_stderr.log | show grid_x$g; done
