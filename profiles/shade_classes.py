#!/usr/bin/env python3
"""Material-class lane utilisation of shade's vertex step, with and without the class partition.
    python profiles/shade_classes.py [scene ...]      (MI355X box; prints one block per scene)
For each scene (1920x1080, 8 spp, authored depth): renders once with CRT_PARTITION=0 (the natural mix of a wave's 64
vertices) and once with the partition on, and prints per class the wave executions that contained the class and the
fraction of their lanes that held it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
from __graft_entry__ import load_package
crt = load_package()
name = sys.argv[1]
path = os.path.join(%r, "scenes", name + ".usda")
if not os.path.exists(path): path = os.path.join(%r, "scenes", name + ".usd")
r, _ = crt.load_usda(path, 1920, 1080, None)
r.shade_class_stats(True)
r.render_samples(0, 8)
torch.cuda.synchronize()
print(json.dumps(r.shade_class_stats()))
""" % (ROOT, ROOT, ROOT)

out = {}
for scene in (sys.argv[1:] or ["openpbr_showcase", "veach_mis", "sun_sky", "cornellbox_guided"]):
    out[scene] = {}
    for part in ("0", "1"):
        env = dict(os.environ, CRT_PARTITION=part)
        res = subprocess.run([sys.executable, "-c", CODE, scene], env=env, capture_output=True, text=True, timeout=600)
        if res.returncode != 0:
            print(res.stderr[-2000:], file=sys.stderr)
            continue
        d = json.loads(res.stdout.strip().splitlines()[-1])
        out[scene]["partition=" + part] = {k: {"wave_execs": v[0], "lane_utilisation": round(v[1], 3)} for k, v in d.items()}
print(json.dumps(out, indent=1))
