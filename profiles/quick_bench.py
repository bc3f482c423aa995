#!/usr/bin/env python3
"""Lean throughput probe for A/B runs of variant libraries (CRT_AMD_LIB=variants/<name>.so): loads one scene, renders
`--warmup` + `--steps` batches, prints one line `<tag> <scene> <Mray/s> <ms/step>`. No oracle, no roofline legs.
    python profiles/quick_bench.py --tag cur --scene PointInstancedMedCity --width 3840 --height 2160 --steps 2"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="cur")
    ap.add_argument("--scene", default="cornellbox")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--reps", type=int, default=2, help="timed repetitions; the best is printed")
    a = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    crt = load_package()
    path = crt.scene_path(a.scene)
    r, _ = crt.load_usda(path, a.width, a.height, a.depth)
    for k in range(a.warmup):
        r.render_samples(k * a.spp, a.spp)
    torch.cuda.synchronize()
    best = None
    for _ in range(a.reps):
        r.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            r.render_samples(k * a.spp, a.spp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = r.stats()
        rate = st.total_rays() / dt / 1e6
        if best is None or rate > best[0]:
            best = (rate, dt / a.steps * 1e3)
    print("%s %s %dx%d %.1f Mray/s %.3f ms/step" % (a.tag, a.scene, a.width, a.height, best[0], best[1]), flush=True)


if __name__ == "__main__":
    main()
