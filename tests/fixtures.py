"""Deterministic input generators restated from the reference's own tests / probes (data, not code paths).

Each generator cites the reference lines whose inputs it reproduces. All arithmetic is done in float32 in
the same order as the reference so the inputs are the same bit patterns.
"""
import math

import numpy as np

f32 = np.float32


class Lcg:
    """state*1664525+1013904223 (triangle.rs:527-531; ray_throughput.rs:104-108)."""

    def __init__(self, seed):
        self.state = seed & 0xFFFFFFFF

    def next_u32(self):
        self.state = (self.state * 1664525 + 1013904223) & 0xFFFFFFFF
        return self.state

    def unit(self):
        """(state >> 8) as f32 / (1 << 24) as f32 — in [0, 1)."""
        return f32(self.next_u32() >> 8) / f32(1 << 24)

    def centered(self):
        """triangle.rs:530: unit - 0.5."""
        return f32(self.unit() - f32(0.5))


def sphere_grid_centers(n, spacing=3.0):
    """bvh.rs:1423-1438 sphere_grid(n): centers (x,y,z)*3.0, radius 0.5, geom_id = x*n*n+y*n+z."""
    out = []
    for x in range(n):
        for y in range(n):
            for z in range(n):
                out.append((f32(x) * f32(spacing), f32(y) * f32(spacing), f32(z) * f32(spacing)))
    return np.array(out, dtype=np.float32)


def diagonal_shards(n):
    """bvh.rs:1443-1458: long thin diagonal triangles; returns (verts[3n,3], idx[n,3])."""
    verts = np.zeros((3 * n, 3), dtype=np.float32)
    for i in range(n):
        o = f32(i) * f32(0.35)
        verts[3 * i] = (o, o, o)
        verts[3 * i + 1] = (o + f32(10.0), o + f32(10.0), o + f32(10.2))
        verts[3 * i + 2] = (o + f32(10.0), o + f32(10.3), o + f32(10.0))
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return verts, idx


def _sincos32(x):
    # Rust f32::sin / f32::cos: correctly-rounded-ish libm; computing in double and rounding once matches
    # to within the last bit, which is all these *inputs* need (both sides consume the same arrays).
    return f32(math.sin(float(x))), f32(math.cos(float(x)))


def uv_sphere(center, radius, segs, rings):
    """ray_throughput.rs:18-48 (also bvh.rs:1686-1725 with center 0, radius 1)."""
    center = np.asarray(center, dtype=np.float32)
    verts = []
    for r in range(rings + 1):
        phi = f32(f32(r) / f32(rings)) * f32(math.pi)
        sp, cp = _sincos32(phi)
        for s in range(segs + 1):
            theta = f32(f32(s) / f32(segs)) * f32(2.0 * math.pi)
            st, ct = _sincos32(theta)
            d = np.array([sp * ct, cp, sp * st], dtype=np.float32)
            verts.append(center + f32(radius) * d)
    idx = []
    row = segs + 1
    for r in range(rings):
        for s in range(segs):
            a, b, c, d = r * row + s, r * row + s + 1, (r + 1) * row + s + 1, (r + 1) * row + s
            idx.append((a, b, c))
            idx.append((a, c, d))
    return np.array(verts, dtype=np.float32), np.array(idx, dtype=np.uint32)


def ray_batch(count, extent, seed=0x2545F491):
    """ray_throughput.rs:103-116: rays[count, 8] = o(3), d(3) normalised, time 0, mask ALL (as bits)."""
    g = Lcg(seed)
    rays = np.zeros((count, 8), dtype=np.float32)
    e4 = f32(4.0) * f32(extent)
    for i in range(count):
        o = np.array([g.unit() - f32(0.5), g.unit() - f32(0.5), g.unit() - f32(0.5)], dtype=np.float32) * e4
        t = np.array([g.unit() - f32(0.5), g.unit() - f32(0.5), g.unit() - f32(0.5)], dtype=np.float32) * f32(extent)
        d = (t - o).astype(np.float32)
        ln = np.sqrt(f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
        d = (d / ln).astype(np.float32)
        rays[i, 0:3] = o
        rays[i, 3:6] = d
    rays[:, 7] = np.array([0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]
    return rays


def fast_ray_batch(count, extent, seed=0x2545F491):
    """Vectorised ray_batch for large counts: same LCG stream, same float32 arithmetic."""
    n = count * 6
    # LCG jump via numpy uint64 loop-free recurrence is awkward; generate sequentially in chunks.
    states = np.empty(n, dtype=np.uint32)
    s = seed & 0xFFFFFFFF
    a, c = 1664525, 1013904223
    # unrolled python loop is ~1 us/iter; fine up to a few million values
    for i in range(n):
        s = (s * a + c) & 0xFFFFFFFF
        states[i] = s
    u = (states >> 8).astype(np.float32) / f32(1 << 24)
    u = u.reshape(count, 6) - f32(0.5)
    o = (u[:, 0:3] * (f32(4.0) * f32(extent))).astype(np.float32)
    t = (u[:, 3:6] * f32(extent)).astype(np.float32)
    d = (t - o).astype(np.float32)
    ln = np.sqrt(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32))
    d = (d / ln[:, None]).astype(np.float32)
    rays = np.zeros((count, 8), dtype=np.float32)
    rays[:, 0:3] = o
    rays[:, 3:6] = d
    rays[:, 7] = np.array([0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]
    return rays


def tri_spheres():
    """ray_throughput.rs:50-65: 27 UV spheres 40x20 at 2.5 spacing = 43 200 triangles; list of (verts, idx)."""
    out = []
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            for z in (-1, 0, 1):
                c = np.array([x, y, z], dtype=np.float32) * f32(2.5)
                out.append(uv_sphere(c, 1.0, 40, 20))
    return out


def sphere_grid_probe():
    """ray_throughput.rs:67-80: 12^3 spheres r=0.6 at (x,y,z)*2-12."""
    out = []
    for x in range(12):
        for y in range(12):
            for z in range(12):
                out.append(np.array([x, y, z], dtype=np.float32) * f32(2.0) - f32(12.0))
    return np.array(out, dtype=np.float32)


def instance_translations():
    """ray_throughput.rs:82-101: 5^3 placements at (x,y,z)*2.5 of a 24x12 UV sphere."""
    out = []
    for x in range(-2, 3):
        for y in range(-2, 3):
            for z in range(-2, 3):
                out.append(np.array([x, y, z], dtype=np.float32) * f32(2.5))
    return out
