#!/usr/bin/env python3
"""bench_kernels.py — the reference's own KERNEL benchmark on the MI355X backend.

crates/crust-rt/examples/ray_throughput.rs times Scene::intersect / Scene::occluded on three probe scenes
(tri_spheres 43 200 tris, sphere_grid 1 728 spheres, instances 125 x 576-tri mesh) with a 4 096-ray LCG batch, one
CPU thread, min of 40; docs/simd.md:206-211 publishes the times. Here:

  gpu          crt_intersect_n / crt_occluded_n (through the C ABI) on N rays of the same distribution (origin in
               +-2*extent, target in +-0.5*extent, numpy PCG64 seed 1; the LCG is sequential and 4 096 rays do not
               fill one MI355X), rays and hits resident in HBM, HIP events around `--reps` launches after a warm-up;
               algorithmic GB/s from the stats build of the same kernels (SURVEY §8d byte formula).
  oracle_1t    the CPU restatement on the reference's exact 4 096-ray LCG batch, one thread, min of 10 —
               the figure comparable with the published one.
  published    docs/simd.md:206-211 (Xeon E5-2699 v4, one thread).

Prints one JSON object; `python bench_kernels.py > profiles/<round>_kernel_probe.json` on the GPU box."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PUBLISHED_MS = {("tri_spheres", "intersect"): 3.426, ("tri_spheres", "occluded"): 1.925,
                ("sphere_grid", "intersect"): 1.655, ("sphere_grid", "occluded"): 0.931,
                ("instances", "intersect"): 3.787, ("instances", "occluded"): 2.313}


def random_rays(n, extent, seed=1):
    rng = np.random.default_rng(seed)
    o = ((rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(4.0 * extent)).astype(np.float32)
    t = ((rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(extent)).astype(np.float32)
    d = t - o
    d /= np.sqrt((d * d).sum(axis=1, dtype=np.float32))[:, None]
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3:6] = o, d
    rays[:, 7] = np.array([0xFFFFFFFF], dtype=np.uint32).view(np.float32)
    return rays


def uv_sphere_np(center, radius, segs, rings):
    """Vectorised tessellation with the probe's topology (ray_throughput.rs:18-48); values are numpy's sin/cos, so
    this scene is for throughput only, not parity."""
    r = np.arange(rings + 1, dtype=np.float64)[:, None] / rings * np.pi
    t = np.arange(segs + 1, dtype=np.float64)[None, :] / segs * 2.0 * np.pi
    d = np.stack([np.sin(r) * np.cos(t), np.cos(r) * np.ones_like(t), np.sin(r) * np.sin(t)], axis=-1).reshape(-1, 3)
    v = (np.asarray(center, dtype=np.float64) + radius * d).astype(np.float32)
    row = segs + 1
    rr, ss = np.meshgrid(np.arange(rings), np.arange(segs), indexing="ij")
    a, b, c, e = rr * row + ss, rr * row + ss + 1, (rr + 1) * row + ss + 1, (rr + 1) * row + ss
    idx = np.stack([np.stack([a, b, c], -1), np.stack([a, c, e], -1)], axis=2).reshape(-1, 3).astype(np.uint32)
    return v, idx


def big_tri_spheres(api, segs, rings):
    """The tri_spheres layout (27 spheres at 2.5 spacing) at a tessellation whose BVH does not fit the caches."""
    b = api.SceneBuilder()
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            for z in (-1, 0, 1):
                v, i = uv_sphere_np((2.5 * x, 2.5 * y, 2.5 * z), 1.0, segs, rings)
                b.attach_triangles(v, i)
    return b.commit()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--big", type=int, default=0, metavar="SEGS",
                    help="also run tri_spheres at SEGS x SEGS/2 quads per sphere (e.g. 512 -> 7.1 M triangles)")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()

    import torch
    from __graft_entry__ import load_package
    import fixtures as fx
    import scenes
    crt = load_package()
    inf = float("inf")
    out = {"device": torch.cuda.get_device_name(0), "rays": args.rays, "reps": args.reps, "scenes": {}}
    names = ["tri_spheres", "sphere_grid", "instances"] + (["tri_spheres_big"] if args.big else [])
    for name in names:
        if name == "tri_spheres_big":
            make, extent = (lambda api: big_tri_spheres(api, args.big, args.big // 2)), 6.0
            t0 = time.perf_counter()
            scene = make(crt)
            fp = scene.memory_footprint() if hasattr(scene, "memory_footprint") else None
            out["big_build_seconds"] = round(time.perf_counter() - t0, 2)
            out["big_triangles"] = 27 * args.big * (args.big // 2) * 2
            if fp is not None:
                out["big_device_bytes"] = int(sum(fp.values()))
        else:
            make, extent = scenes.ALL[name]
            scene = make(crt)
        d_rays = crt.rays_to_device(random_rays(args.rays, extent))
        d_hits = torch.empty(args.rays * 40, dtype=torch.uint8, device="cuda")
        d_occ = torch.empty(args.rays, dtype=torch.int32, device="cuda")
        entry = {}
        for query in ("intersect", "occluded"):
            def launch(stats=None):
                if query == "intersect":
                    scene.intersect_n(d_rays, 0.001, inf, d_hits, stats=stats)
                else:
                    scene.occluded_n(d_rays, 0.001, inf, d_occ, stats=stats)
            launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            st = crt.CrtTravStats()
            launch(st)
            torch.cuda.synchronize()
            n_hit = int((crt.hits_to_host(d_hits)["geom_id"] != 0xFFFFFFFF).sum()) if query == "intersect" else int(d_occ.sum().item())
            row = {"gpu_ms_per_launch": round(ms, 4), "gpu_mray_s": round(args.rays / ms / 1e3, 1),
                   "hits": n_hit, "bytes_per_ray": round(st.algorithmic_bytes() / args.rays, 1),
                   "algorithmic_gb_s": round(st.algorithmic_bytes() / (ms * 1e-3) / 1e9, 1),
                   "nodes_per_ray": round((st.nodes[0] + st.nodes[1]) / args.rays, 2),
                   "lane_utilisation": {k: round(u, 3) for k, (w, u) in st.utilisation().items() if w},
                   "published_cpu_1t_mray_s": round(4096 / PUBLISHED_MS[(name, query)] / 1e3, 2) if (name, query) in PUBLISHED_MS else None}
            if not args.no_oracle and name != "tri_spheres_big":
                import ora
                o_scene = make(ora)
                rays = fx.ray_batch(4096, extent)
                best = 1e9
                for _ in range(10):
                    t0 = time.perf_counter()
                    if query == "intersect":
                        o_scene.intersect_n(rays, 0.001, inf)
                    else:
                        o_scene.occluded_n(rays, 0.001, inf)
                    best = min(best, time.perf_counter() - t0)
                row["oracle_1t_mray_s"] = round(4096 / best / 1e6, 2)
            entry[query] = row
        out["scenes"][name] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
