#!/usr/bin/env python3
"""Would two half-batches in flight on two HIP streams beat one batch (round 3 probe)? Two renderers of the bench scene,
256 spp per batch each, launched alternately on two streams, against one renderer at 256 and at 512 spp per batch.
Throughput only: the two renderers keep separate films."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
crt = load_package()
path = crt.scene_path("cornellbox")

def one(spp, steps=4):
    r, _ = crt.load_usda(path, 1920, 1080)
    r.render_samples(0, spp); torch.cuda.synchronize(); r.clear(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): r.render_samples(k * spp, spp)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("one renderer, %d spp per batch: %.1f Mray/s" % (spp, r.stats().total_rays() / dt / 1e6), flush=True)
    del r; torch.cuda.empty_cache()

def two(spp, steps=4, n=2):
    rs = [crt.load_usda(path, 1920, 1080)[0] for _ in range(n)]
    ss = [torch.cuda.Stream() for _ in range(n)]
    for r, s in zip(rs, ss): r.render_samples(0, spp, s)
    torch.cuda.synchronize()
    for r in rs: r.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        for r, s in zip(rs, ss): r.render_samples(k * spp, spp, s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d renderers on %d streams, %d spp per batch each: %.1f Mray/s" % (n, n, spp, sum(r.stats().total_rays() for r in rs) / dt / 1e6), flush=True)
    del rs; torch.cuda.empty_cache()

two(256); two(384); two(192, n=3); two(256, n=3); two(128, n=4); two(192, n=4)
