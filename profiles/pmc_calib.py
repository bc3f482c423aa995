#!/usr/bin/env python3
"""Known-byte kernels for calibrating the TCC_EA0_* / FETCH_SIZE / WRITE_SIZE counters on gfx950 (run under
profiles/run_pmc_r02.sh): a 4 GiB coalesced streaming read + 4 GiB write, and 32 Mi random 128-byte record gathers
(= the traversal's node fetch pattern) from an 8 GiB table — both far past the 256 MiB Infinity Cache."""
import torch

torch.manual_seed(1)
n = 1 << 30
x = torch.ones(n, dtype=torch.float32, device="cuda")           # 4 GiB
y = torch.empty_like(x)
table = torch.ones((1 << 26, 32), dtype=torch.float32, device="cuda")   # 64 Mi records of 128 B = 8 GiB
idx = torch.randint(0, 1 << 26, (1 << 25,), device="cuda")       # 32 Mi gathers = 4 GiB read, 4 GiB written, 256 MiB of indices
torch.cuda.synchronize()
for _ in range(3):
    torch.add(x, 1.0, out=y)                                      # stream: reads 4 GiB, writes 4 GiB
for _ in range(3):
    out = table.index_select(0, idx)                              # gather
torch.cuda.synchronize()
print("calib done", float(y[0]), float(out[0, 0]))
