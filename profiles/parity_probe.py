#!/usr/bin/env python3
"""Debug aid: render one scene under the environment given on the command line and count the pixels that differ
from the oracle.  python profiles/parity_probe.py <scene> <w> <h> <spp> <depth|-> [VAR=val ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import ora, ora_world
crt = load_package()
name, w, h, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
depth = None if depth == "-" else int(depth)
path = os.path.join(%r, "scenes", name + ".usda")
if not os.path.exists(path): path = os.path.join(%r, "scenes", name + ".usd")
r, desc = crt.load_usda(path, w, h, depth)
r.render_samples(0, spp); torch.cuda.synchronize()
img, st = r.image(), r.stats()
oimg, ost = ora_world.OracleRenderer(desc, crt.usda, max_depth=depth).render(spp, forward=1)
r2, _ = crt.load_usda(path, w, h, depth)
r2.render_samples(0, spp); torch.cuda.synchronize()
img2 = r2.image()
gg = (img.view(np.uint32) != img2.view(np.uint32)).any(axis=-1)
bad = (img.view(np.uint32) != oimg.view(np.uint32)).any(axis=-1)
if bad.any():
    rows = np.nonzero(bad.any(axis=1))[0]
    ys, xs = np.nonzero(bad)
    k = len(ys) // 2
    print("rows with differences %%d..%%d of %%d; GPU run 1 vs run 2 differ in %%d pixels; sample (%%d,%%d): gpu %%s oracle %%s" %% (
        rows.min(), rows.max(), h, gg.sum(), xs[k], ys[k], img[ys[k], xs[k]], oimg[ys[k], xs[k]]))
    rel = np.abs(img - oimg)[bad] / np.maximum(np.abs(oimg)[bad], 1e-6)
    print("relative diff of differing pixels: median %%.3g max %%.3g" %% (np.median(rel.max(axis=-1)), rel.max()))
same_counters = all(getattr(st, f) == getattr(ost, f) for f, _ in ora.RayStats._fields_)
print("differing pixels %%d / %%d, counters equal: %%s, max abs diff %%.3g" %% (bad.sum(), bad.size, same_counters, np.abs(img - oimg).max()))
""" % (ROOT, ROOT, ROOT, ROOT)
args, envs = [a for a in sys.argv[1:] if "=" not in a], dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
res = subprocess.run([sys.executable, "-c", CODE] + args, env=dict(os.environ, **envs), capture_output=True, text=True, timeout=900)
print(" ".join(sys.argv[1:]), "->", " | ".join(res.stdout.strip().splitlines()[-3:]) or res.stderr[-500:])
