"""Synthetic scenes built in code (SceneDesc, like the USDA reader produces).

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
