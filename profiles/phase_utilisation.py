import os, sys
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
crt = load_package()
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cornellbox.usda"
r, desc = crt.load_usda(scene, 1920, 1080, None)
ext, sh = r.render_samples_stats(0, 8)
print("rays", int(ext.rays), "bytes/ray", ext.algorithmic_bytes() / int(ext.rays))
print(ext.as_dict())
tot_w = sum(int(ext.phase_waves[k]) for k in range(1, 8))
for n, (w, u) in ext.utilisation().items():
    k = list(ext.utilisation()).index(n)
    cyc = int(ext.phase_cycles[k])
    print(f"{n:14s} wave-execs {w:12d}  per-ray {w*64/int(ext.rays):7.3f}  util {u:6.3f}  cycles {cyc/1e9:8.3f}G  cyc/exec {cyc/max(w,1):9.1f}")
print("total cycles", sum(int(ext.phase_cycles[k]) for k in range(8))/1e9, "G")
