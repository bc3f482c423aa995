#!/bin/bash
# Where the per-stage pipeline starts to pay: bench.py's scene at small batches, fused against per-stage (forced).
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['spp_per_step'], 'spp/step', d['config']['paths_per_step_per_gpu'], 'paths', d['value'], 'Mray/s', d['roofline']['pipeline'][:9])"; }
for s in 8 16 32 64; do
  CRT_FUSED=1 python bench.py --no-cpu-baseline --spp-per-step $s --steps $((256 / s)) --warmup 1 2>>gpurun_out/ab_stderr.log | show fused
  CRT_FUSED=0 CRT_WIDE=1 python bench.py --no-cpu-baseline --spp-per-step $s --steps $((256 / s)) --warmup 1 2>>gpurun_out/ab_stderr.log | show stage
done
