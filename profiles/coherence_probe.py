#!/usr/bin/env python3
"""What ray ORDER is worth to the traversal kernel (round 3): bounce-like rays of a scene through crt_intersect_n in
the order a wavefront produces them (camera-pixel order of the vertex they leave from) and sorted by origin cell x
direction octant at several granularities.   python profiles/coherence_probe.py [--scene cornellbox] [--n 16777216]
Rays: camera rays of a WxH frame -> their hits -> one cosine-distributed direction per hit point (numpy, seed 1).
Throughput only (no parity claim: the directions are numpy's)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornellbox")
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    crt = load_package()
    d = crt.usda.load(crt.scene_path(a.scene), a.width, a.height)
    scene, _m, _p = crt.usda.build_world(d, crt, crt.default_material)
    cam = d.camera
    # camera rays, pixel order by 16x16 tiles like the renderer (coherent reference)
    w, h = a.width, a.height
    lf, la, vup = (np.array(cam[k], dtype=np.float64) for k in ("lookfrom", "lookat", "vup"))
    th = np.tan(np.radians(cam["vfov_deg"]) / 2)
    wv = lf - la; wv /= np.linalg.norm(wv); uv = np.cross(vup, wv); uv /= np.linalg.norm(uv); vv = np.cross(wv, uv)
    ys, xs = np.mgrid[0:h, 0:w]
    ty, tx = ys // 16, xs // 16
    order = np.lexsort((xs.ravel() % 16, ys.ravel() % 16, tx.ravel(), ty.ravel()))
    s = ((xs.ravel()[order] + 0.5) / w * 2 - 1) * th * cam["aspect"]
    t = ((ys.ravel()[order] + 0.5) / h * 2 - 1) * th
    dirs = (s[:, None] * uv + t[:, None] * vv - wv).astype(np.float32)
    n = dirs.shape[0]
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = lf.astype(np.float32); rays[:, 3:6] = dirs
    rays[:, 7] = np.array([0xFFFFFFFF], dtype=np.uint32).view(np.float32)

    def run(r8, tag):
        dr = crt.rays_to_device(r8)
        dh = scene.intersect_n(dr, 0.001, float("inf"))
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter()
            scene.intersect_n(dr, 0.001, float("inf"), dh)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%-44s %9d rays %8.3f ms %8.1f Mray/s" % (tag, r8.shape[0], best * 1e3, r8.shape[0] / best / 1e6), flush=True)
        return crt.hits_to_host(dh)

    hits = run(rays, "camera rays (tile order)")
    ok = hits["geom_id"] != 0xFFFFFFFF
    p = rays[ok, 0:3] + rays[ok, 3:6] * hits["t"][ok, None]
    nrm = hits["normal"][ok]
    nrm = np.where((np.einsum("ij,ij->i", nrm, rays[ok, 3:6]) > 0)[:, None], -nrm, nrm)
    rng = np.random.default_rng(1)
    m = p.shape[0]
    u1, u2 = rng.random(m), rng.random(m)
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    lx, ly, lz = r * np.cos(phi), r * np.sin(phi), np.sqrt(1 - u1)
    aa = np.where(np.abs(nrm[:, 0:1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    tb = np.cross(nrm, aa); tb /= np.linalg.norm(tb, axis=1)[:, None]; bb = np.cross(nrm, tb)
    bd = (lx[:, None] * tb + ly[:, None] * bb + lz[:, None] * nrm).astype(np.float32)
    b = np.zeros((m, 8), dtype=np.float32)
    b[:, 0:3] = (p + nrm * 1e-4).astype(np.float32); b[:, 3:6] = bd; b[:, 7] = rays[0, 7]
    run(b, "bounce rays, wavefront (pixel) order")
    # the renderer's order: 256-ray chunks of consecutive pixels dealt round-robin to the segments (balanced segments,
    # neighbouring pixels within a wave)
    for nseg in (1024, 4096):
        ch = np.arange(m) // 256
        dealt = np.lexsort((np.arange(m), ch // nseg, ch % nseg))
        run(b[dealt], "bounce rays, 256-chunks dealt to %d segments" % nseg)
        sh = rng.permutation(m // 64) if False else None
    w64 = np.arange(m) // 64
    perm_w = rng.permutation(w64.max() + 1)
    run(b[np.argsort(perm_w[w64], kind="stable")], "bounce rays, whole 64-ray groups shuffled")
    b = b[rng.permutation(m)]  # every segment below is a uniform sample of the frame, as the renderer's segments are
    bd = b[:, 3:6]
    run(b, "bounce rays, random order")
    lo, hi = b[:, 0:3].min(0), b[:, 0:3].max(0)
    octant = (bd[:, 0] < 0).astype(np.int64) | ((bd[:, 1] < 0).astype(np.int64) << 1) | ((bd[:, 2] < 0).astype(np.int64) << 2)
    ax = np.argmax(np.abs(bd), axis=1); sgn = (np.take_along_axis(bd, ax[:, None], 1)[:, 0] < 0).astype(np.int64)
    face = ax * 2 + sgn
    for seg_rays in (16 * 1024,):
        seg = np.arange(m) // seg_rays
        run(b[np.lexsort((octant, seg))], "balanced %dKi segments, sorted: octant" % (seg_rays // 1024))
        run(b[np.lexsort((face, seg))], "balanced %dKi segments, sorted: cube face" % (seg_rays // 1024))
        for cells in (4,):
            c = np.minimum(((b[:, 0:3] - lo) / (hi - lo + 1e-6) * cells).astype(np.int64), cells - 1)
            cell = (c[:, 0] * cells + c[:, 1]) * cells + c[:, 2]
            run(b[np.lexsort((cell, seg))], "balanced %dKi segments, sorted: %d^3 cells" % (seg_rays // 1024, cells))
            run(b[np.lexsort((cell * 8 + octant, seg))], "balanced %dKi segments, sorted: %d^3 cells x octant" % (seg_rays // 1024, cells))
            run(b[np.lexsort((octant * cells ** 3 + cell, seg))], "balanced %dKi segments, sorted: octant-major x %d^3 cells" % (seg_rays // 1024, cells))

if __name__ == "__main__":
    main()
