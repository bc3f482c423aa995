"""Synthetic scenes built in code (SceneDesc, like the USDA reader produces).

`big`: the out-of-cache path-traced workload (27 tessellated spheres, 7.08 M triangles at the default tessellation, a
rect light) on which the traversal kernels are HBM-bound inside the integrator — see its docstring.

`city`: the stand-in for BASELINE config 5. `PointInstancedMedCity.usd` is a binary USDC crate (LZ4 sections) that
nothing here can read, so the instancing-heavy workload it stands for — thousands of placements of
a few prototype meshes under one top-level BVH, every camera ray descending into instances — is generated instead,
following the reference's own instancing probe (crates/crust-rt/examples/traversal_probe.rs:133-169: N instances of
one prototype on a regular grid, spacing > prototype size). It is LABELLED synthetic wherever it is reported.
"""
import math

import numpy as np

from . import usda

f32 = np.float32


class _Lcg:  # triangle.rs:527-531 generator, so the scene is a pure function of its arguments
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF

    def unit(self):
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return float(f32(self.s >> 8) / f32(1 << 24))


def _box(sx, sy, sz):
    """Axis-aligned box [-sx/2, sx/2] x [0, sy] x [-sz/2, sz/2]: 8 vertices, 12 triangles (outward winding)."""
    x, z = sx / 2.0, sz / 2.0
    v = np.array([(-x, 0, -z), (x, 0, -z), (x, 0, z), (-x, 0, z), (-x, sy, -z), (x, sy, -z), (x, sy, z), (-x, sy, z)],
                 dtype=np.float32)
    i = np.array([(0, 1, 2), (0, 2, 3), (4, 6, 5), (4, 7, 6), (0, 4, 5), (0, 5, 1), (1, 5, 6), (1, 6, 2),
                  (2, 6, 7), (2, 7, 3), (3, 7, 4), (3, 4, 0)], dtype=np.uint32)
    return v, i


def _tower(segs=12):
    """A round tower with a conical roof: segs*4 triangles."""
    ang = np.arange(segs, dtype=np.float64) / segs * 2.0 * math.pi
    ring0 = np.stack([0.5 * np.cos(ang), np.zeros(segs), 0.5 * np.sin(ang)], axis=1)
    ring1 = ring0 + np.array([0.0, 1.0, 0.0])
    v = np.concatenate([ring0, ring1, [[0.0, 1.5, 0.0]]]).astype(np.float32)
    tris = []
    for k in range(segs):
        a, b = k, (k + 1) % segs
        tris += [(a, segs + a, segs + b), (a, segs + b, b), (segs + a, 2 * segs, segs + b)]
    return v, np.array(tris, dtype=np.uint32)


def _l2w(scale, yaw, t):
    """Affine3A as 12 floats (matrix3 columns x, y, z, then translation): T * Ry(yaw) * S, composed in f32."""
    c, s = f32(math.cos(yaw)), f32(math.sin(yaw))
    sx, sy, sz = (f32(v) for v in scale)
    return np.array([c * sx, 0.0, -s * sx, 0.0, sy, 0.0, s * sz, 0.0, c * sz, t[0], t[1], t[2]], dtype=np.float32)


def city(width=640, height=360, side=64, seed=0x2545F491, max_depth=8):
    """side x side lots; every lot is ONE instance of one of four prototypes, scaled and turned. 64 -> 4 096
    instances; 181 -> 32 761. Lit by two sphere lights and the sky gradient."""
    d = usda.SceneDesc()
    g = _Lcg(seed)
    pitch = 3.0
    ext = pitch * side
    ground = np.array([(-ext, 0, -ext), (ext, 0, -ext), (ext, 0, ext), (-ext, 0, ext)], dtype=np.float32)
    d.geoms.append(dict(kind="mesh", verts=ground, idx=np.array([(0, 2, 1), (0, 3, 2)], np.uint32), mask=0xFFFFFFFF,
                        material={"_preset": "diffuse", "base_color": (0.35, 0.35, 0.33), "specular_weight": 0.0},
                        name="ground"))
    protos = [_box(1.0, 1.0, 1.0), _box(1.0, 1.0, 0.6), _tower(12), _tower(20)]
    d.protos = [dict(verts=v, idx=i) for v, i in protos]
    palette = [
        {"base_color": (0.62, 0.60, 0.55), "specular_roughness": 0.6},
        {"base_color": (0.30, 0.33, 0.38), "specular_roughness": 0.25},
        {"base_color": (0.70, 0.35, 0.25), "specular_roughness": 0.5},
        {"base_color": (0.85, 0.85, 0.88), "base_metalness": 1.0, "specular_roughness": 0.2},
        {"base_color": (0.25, 0.45, 0.30), "specular_roughness": 0.7, "coat_weight": 0.5},
    ]
    half = pitch * side / 2.0
    for j in range(side):
        for i in range(side):
            proto = int(g.unit() * len(protos)) % len(protos)
            foot = 1.2 + 1.2 * g.unit()
            tall = 1.0 + 9.0 * g.unit() * g.unit()
            yaw = (int(g.unit() * 4) % 4) * (math.pi / 2.0) + (g.unit() - 0.5) * 0.2
            t = (f32(i * pitch - half + 0.5 * pitch), f32(0.0), f32(j * pitch - half + 0.5 * pitch))
            mat = dict(palette[int(g.unit() * len(palette)) % len(palette)])
            d.geoms.append(dict(kind="instance", proto=proto, l2w=_l2w((foot, tall, foot), yaw, t), mask=0xFFFFFFFF,
                                material=mat, name="lot_%d_%d" % (i, j)))
    for center, radius, rad in (((0.3 * ext, 0.9 * ext, -0.2 * ext), 0.08 * ext, (60.0, 56.0, 50.0)),
                                ((-0.5 * ext, 0.4 * ext, 0.6 * ext), 0.03 * ext, (20.0, 24.0, 40.0))):
        gid = len(d.geoms)
        c = np.array(center, dtype=np.float32)
        d.geoms.append(dict(kind="sphere", center=c, radius=f32(radius), mask=0xFFFFFFFF & ~2,  # lights cast no shadows
                            material={"_preset": "emissive", "emission_color": rad}, name="light%d" % gid))
        d.lights.append(dict(kind="sphere", geom_id=gid, radiance=np.array(rad, dtype=np.float32), center=c, radius=f32(radius)))
    lookfrom = np.array([0.0, 0.35 * ext, 0.75 * ext], dtype=np.float32)
    lookat = np.array([0.0, 0.0, 0.0], dtype=np.float32)
    d.camera = dict(lookfrom=lookfrom, lookat=lookat, vup=np.array([0, 1, 0], dtype=np.float32), vfov_deg=f32(40.0),
                    aspect=f32(f32(width) / f32(height)), aperture=f32(0.0), focus_dist=f32(10.0))
    d.settings = dict(usda.DEFAULTS, strategy="power", filter="triangle", filter_radius=1.0, width=width, height=height,
                      max_depth=max_depth)
    return d



def _uv_sphere(center, radius, segs, rings):
    """The reference probe's UV-sphere topology (crates/crust-rt/examples/ray_throughput.rs:18-48), vectorised: values are
    numpy's sin / cos in f64 rounded to f32 — the SAME arrays reach the device library and the oracle, so parity holds
    whatever the last bit of a vertex is."""
    r = np.arange(rings + 1, dtype=np.float64)[:, None] / rings * np.pi
    t = np.arange(segs + 1, dtype=np.float64)[None, :] / segs * 2.0 * np.pi
    d = np.stack([np.sin(r) * np.cos(t), np.cos(r) * np.ones_like(t), np.sin(r) * np.sin(t)], axis=-1).reshape(-1, 3)
    v = (np.asarray(center, dtype=np.float64) + radius * d).astype(np.float32)
    row = segs + 1
    rr, ss = np.meshgrid(np.arange(rings), np.arange(segs), indexing="ij")
    a, b, c, e = rr * row + ss, rr * row + ss + 1, (rr + 1) * row + ss + 1, (rr + 1) * row + ss
    idx = np.stack([np.stack([a, b, c], -1), np.stack([a, c, e], -1)], axis=2).reshape(-1, 3).astype(np.uint32)
    return v, idx


def big(width=640, height=360, side=512, max_depth=6):
    """SYNTHETIC (generated in code, not a reference file; labelled so wherever it is reported). The out-of-cache
    path-traced workload: the reference's `tri_spheres` probe layout (ray_throughput.rs:50-65: 27 UV spheres on a 3x3x3
    grid at 2.5 spacing) at `side` x `side`/2 quads per sphere — 512 -> 7 077 888 triangles, a 1.03 GB device image, four
    times the Infinity Cache — in a closed room under a 10 x 10 RectLight on its ceiling, the cluster filling the frame. Diffuse and rough-metal
    looks only (simple material table, flat shading: the leanest kernel instances), depth 6: camera rays, incoherent bounce
    rays between the spheres and one shadow ray per vertex all traverse a tree that does not fit any cache — the
    north_star's ">= 40 % of HBM peak in the traversal kernel" is physically reachable only on a scene like this
    (SURVEY §8d). side = 128 (442 368 triangles) is the size the parity tests render against the oracle."""
    d = usda.SceneDesc()
    # bright looks (albedo 0.8-0.9). Paths still thin out fast from the fourth vertex on: the reference's convention
    # `value = brdf * cos` with the tracer multiplying by the cosine again (material.rs:10-12) takes a mean factor of
    # albedo * 2/3 per diffuse bounce, and Russian roulette (tracer.rs:1478-1490) follows the throughput
    looks = [
        {"base_color": (0.90, 0.82, 0.80), "specular_weight": 0.0},
        {"base_color": (0.80, 0.90, 0.82), "specular_weight": 0.0},
        {"base_color": (0.80, 0.84, 0.92), "specular_weight": 0.5, "specular_roughness": 0.4},
        {"base_color": (0.92, 0.90, 0.84), "base_metalness": 1.0, "specular_roughness": 0.35},
        {"base_color": (0.88, 0.88, 0.88), "specular_weight": 0.0},
    ]
    k = 0
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            for z in (-1, 0, 1):
                v, i = _uv_sphere((2.5 * x, 2.5 * y, 2.5 * z), 1.0, side, side // 2)
                d.geoms.append(dict(kind="mesh", verts=v, idx=i, mask=0xFFFFFFFF, material=dict(looks[k % len(looks)]),
                                    name="ball_%d" % k))
                k += 1
    # a closed room around the cluster (floor, four walls, ceiling: 12 triangles, normals facing in): paths keep bouncing
    # to the depth limit instead of leaving for the sky, so most of a batch's rays are incoherent bounce and shadow rays
    lo, hi = np.array([-9.0, -3.9, -9.0]), np.array([9.0, 9.0, 13.0])
    c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], dtype=np.float32)
    quads = [(0, 1, 5, 4), (2, 6, 7, 3), (0, 2, 3, 1), (4, 5, 7, 6), (0, 4, 6, 2), (1, 3, 7, 5)]  # floor, ceiling, -x, +x, -z, +z
    room = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], dtype=np.uint32)
    d.geoms.append(dict(kind="mesh", verts=c, idx=room, mask=0xFFFFFFFF,
                        material={"base_color": (0.85, 0.85, 0.83), "specular_weight": 0.0}, name="room"))
    o = np.array([-5.0, 8.9, -5.0], dtype=np.float32)
    eu, ev = np.array([10.0, 0.0, 0.0], dtype=np.float32), np.array([0.0, 0.0, 10.0], dtype=np.float32)
    rad = (9.0, 8.6, 8.0)
    gid = len(d.geoms)
    quad = np.stack([o, o + eu, o + eu + ev, o + ev]).astype(np.float32)
    d.geoms.append(dict(kind="mesh", verts=quad, idx=np.array([(0, 1, 2), (0, 2, 3)], np.uint32), mask=0xFFFFFFFF & ~2,
                        material={"_preset": "emissive", "emission_color": rad}, name="light"))
    d.lights.append(dict(kind="rect", geom_id=gid, radiance=np.array(rad, np.float32), origin=o, edge_u=eu, edge_v=ev,
                         normal=np.array([0, -1, 0], np.float32)))
    d.camera = dict(lookfrom=np.array([5.5, 3.0, 10.0], np.float32), lookat=np.array([0, -0.2, 0], np.float32),
                    vup=np.array([0, 1, 0], np.float32), vfov_deg=f32(38.0), aspect=f32(f32(width) / f32(height)),
                    aperture=f32(0.0), focus_dist=f32(10.0))
    d.settings = dict(usda.DEFAULTS, strategy="power", filter="triangle", filter_radius=1.0, width=width, height=height,
                      max_depth=max_depth)
    return d
