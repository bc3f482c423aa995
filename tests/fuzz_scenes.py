"""Random scene recipes shared by the randomised parity tests (GPU traversal, CPU build parity)."""
import math

import numpy as np

f32 = np.float32


def _rot(rng):
    a, b = rng.uniform(0, 2 * math.pi, 2)
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rz = np.array([[math.cos(b), -math.sin(b), 0], [math.sin(b), math.cos(b), 0], [0, 0, 1]])
    return (ry @ rz).astype(np.float32)


def _soup(rng, n, spread, size):
    c = rng.uniform(-spread, spread, (n, 1, 3))
    v = (c + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)
    k = max(n // 16, 1)
    v[:k, 2] = v[:k, 1]                 # degenerate slivers (zero area): must be rejected the same way
    v[k:2 * k] = v[2 * k:3 * k]         # exactly coincident triangles: ties go to the later one
    q = np.round(v[3 * k:4 * k] * 2) / 2  # vertices on a coarse lattice: shared edges, axis-aligned faces
    v[3 * k:4 * k] = q
    return v.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def recipe(seed):
    """A scene as a list of build steps, replayable through either API."""
    rng = np.random.default_rng(seed)
    steps = []
    n_proto = int(rng.integers(1, 4))
    protos = []
    for p in range(n_proto):
        v, i = _soup(rng, int(rng.integers(8, 200)), 1.0, 0.4)
        spheres = [(rng.uniform(-1, 1, 3).astype(np.float32), float(rng.uniform(0.1, 0.5))) for _ in range(int(rng.integers(0, 3)))]
        protos.append(dict(v=v, i=i, spheres=spheres, nested=[]))
    for p in range(1, n_proto):  # some prototypes instance earlier ones (nesting)
        if rng.random() < 0.7:
            protos[p]["nested"].append((int(rng.integers(0, p)), _rot(rng) * f32(rng.uniform(0.3, 1.2)), rng.uniform(-1, 1, 3).astype(np.float32)))
    top = []
    v, i = _soup(rng, int(rng.integers(16, 600)), 5.0, 0.8)
    top.append(("mesh", v, i, [0xFFFFFFFF, 1, 2, 4, 3][int(rng.integers(0, 5))]))
    for _ in range(int(rng.integers(2, 10))):
        top.append(("sphere", rng.uniform(-5, 5, 3).astype(np.float32), float(rng.uniform(0.2, 1.2)), [0xFFFFFFFF, 4, 6][int(rng.integers(0, 3))]))
    for _ in range(int(rng.integers(2, 12))):
        m = _rot(rng) * np.array([rng.uniform(0.4, 2.0), rng.uniform(0.4, 2.0), rng.uniform(0.4, 2.0)], dtype=np.float32)[None, :]
        t = rng.uniform(-5, 5, 3).astype(np.float32)
        end = (m, (t + rng.uniform(-1, 1, 3)).astype(np.float32)) if rng.random() < 0.3 else None
        top.append(("instance", int(rng.integers(0, n_proto)), m.astype(np.float32), t, end, [0xFFFFFFFF, 5][int(rng.integers(0, 2))]))
    return protos, top


def build(api, recipe):
    protos, top = recipe
    built = []
    for p in protos:
        b = api.SceneBuilder()
        b.attach_triangles(p["v"], p["i"])
        for c, r in p["spheres"]:
            b.attach_sphere(c, r)
        for (q, m, t) in p["nested"]:
            b.attach_instance(built[q], api.affine(m, t))
        built.append(b.commit())
    b = api.SceneBuilder()
    for s in top:
        if s[0] == "mesh":
            b.attach_triangles(s[1], s[2], mask=s[3])
        elif s[0] == "sphere":
            b.attach_sphere(s[1], s[2], mask=s[3])
        else:
            end = None if s[4] is None else api.affine(s[4][0], s[4][1])
            b.attach_instance(built[s[1]], api.affine(s[2], s[3]), end, mask=s[5])
    return b.commit(), built




def _affine12(m3, t):
    """glam Affine3A as 12 floats: matrix3 columns x, y, z then translation (include/crt.h, crt_attach_instance)."""
    m = np.asarray(m3, dtype=np.float32)
    return np.concatenate([m[:, 0], m[:, 1], m[:, 2], np.asarray(t, dtype=np.float32)]).astype(np.float32)


# ---------------------------------------------------------------- random renderable worlds (SceneDesc)
def _rand_material(rng):
    """Every OpenPBR knob, with each optional lobe switched on at random (openpbr.rs:66-121)."""
    u = rng.uniform
    col = lambda lo=0.05, hi=0.95: tuple(float(x) for x in u(lo, hi, 3))  # noqa: E731
    m = {"base_weight": float(u(0.3, 1.0)), "base_color": col(), "base_diffuse_roughness": float(u(0, 1)),
         "specular_weight": float(u(0, 1)), "specular_color": col(0.5, 1.0), "specular_roughness": float(u(0.02, 0.9)),
         "specular_ior": float(u(1.1, 2.2)), "specular_roughness_anisotropy": float(u(0, 0.9)) if rng.random() < 0.4 else 0.0}
    if rng.random() < 0.3:
        m["base_metalness"] = float(u(0.3, 1.0))
    if rng.random() < 0.35:
        m.update(transmission_weight=float(u(0.4, 1.0)), transmission_color=col(0.3, 1.0))
        if rng.random() < 0.6:
            m["transmission_depth"] = float(u(0.2, 3.0))
        if rng.random() < 0.4:
            m.update(transmission_scatter=col(0.0, 0.6), transmission_scatter_anisotropy=float(u(-0.7, 0.7)))
        if rng.random() < 0.4:
            m.update(transmission_dispersion_scale=float(u(0.2, 1.0)), transmission_dispersion_abbe_number=float(u(15, 60)))
        if rng.random() < 0.3:
            m["thin_walled"] = True
    if rng.random() < 0.3:
        m.update(subsurface_weight=float(u(0.3, 1.0)), subsurface_color=col(), subsurface_radius=float(u(0.05, 1.0)),
                 subsurface_radius_scale=col(0.2, 1.0), subsurface_scatter_anisotropy=float(u(-0.5, 0.8)))
    if rng.random() < 0.3:
        m.update(fuzz_weight=float(u(0.2, 1.0)), fuzz_color=col(), fuzz_roughness=float(u(0.1, 0.9)))
    if rng.random() < 0.35:
        m.update(coat_weight=float(u(0.2, 1.0)), coat_color=col(0.3, 1.0), coat_roughness=float(u(0.0, 0.5)),
                 coat_ior=float(u(1.2, 2.0)), coat_darkening=float(u(0, 1)),
                 coat_roughness_anisotropy=float(u(0, 0.8)) if rng.random() < 0.3 else 0.0)
    if rng.random() < 0.25:
        m.update(thin_film_weight=float(u(0.3, 1.0)), thin_film_thickness=float(u(0.1, 1.0)), thin_film_ior=float(u(1.2, 1.8)))
    if rng.random() < 0.15:
        m.update(emission_luminance=float(u(0.5, 4.0)), emission_color=col())
    return m


def random_world(usda, seed, width=40, height=28):
    """A SceneDesc with random OpenPBR materials on spheres, a triangle soup and a ground, one to three area lights of both
    kinds, sometimes a distant light and a uniform dome, a random strategy / filter and a random thin-lens camera."""
    rng = np.random.default_rng(seed)
    d = usda.SceneDesc()
    g = np.array([(-8, 0, -8), (8, 0, -8), (8, 0, 8), (-8, 0, 8)], dtype=np.float32)
    d.geoms.append(dict(kind="mesh", verts=g, idx=np.array([(0, 2, 1), (0, 3, 2)], np.uint32), mask=0xFFFFFFFF,
                        material=_rand_material(rng), name="ground"))
    for k in range(int(rng.integers(3, 8))):
        c = np.array([rng.uniform(-3, 3), rng.uniform(0.4, 2.0), rng.uniform(-3, 3)], dtype=np.float32)
        d.geoms.append(dict(kind="sphere", center=c, radius=f32(rng.uniform(0.3, 1.0)), mask=0xFFFFFFFF,
                            material=_rand_material(rng), name="s%d" % k))
    v, i = _soup(rng, int(rng.integers(10, 60)), 2.5, 0.7)
    v[:, 1] = np.abs(v[:, 1]) + f32(0.2)
    d.geoms.append(dict(kind="mesh", verts=v, idx=i, mask=0xFFFFFFFF, material=_rand_material(rng), name="soup"))
    for k in range(int(rng.integers(1, 4))):
        rad = tuple(float(x) for x in rng.uniform(2, 30, 3))
        gid = len(d.geoms)
        if rng.random() < 0.6:
            c = np.array([rng.uniform(-4, 4), rng.uniform(3, 6), rng.uniform(-4, 4)], dtype=np.float32)
            r = f32(rng.uniform(0.1, 0.8))
            d.geoms.append(dict(kind="sphere", center=c, radius=r, mask=0xFFFFFFFF & ~2,
                                material={"_preset": "emissive", "emission_color": rad}, name="L%d" % k))
            d.lights.append(dict(kind="sphere", geom_id=gid, radiance=np.array(rad, np.float32), center=c, radius=r))
        else:
            o = np.array([rng.uniform(-3, 1), rng.uniform(4, 6), rng.uniform(-3, 1)], dtype=np.float32)
            eu = np.array([rng.uniform(0.5, 2), 0, 0], dtype=np.float32)
            ev = np.array([0, 0, rng.uniform(0.5, 2)], dtype=np.float32)
            verts = np.stack([o, o + eu, o + eu + ev, o + ev]).astype(np.float32)
            d.geoms.append(dict(kind="mesh", verts=verts, idx=np.array([(0, 1, 2), (0, 2, 3)], np.uint32), mask=0xFFFFFFFF & ~2,
                                material={"_preset": "emissive", "emission_color": rad}, name="L%d" % k))
            d.lights.append(dict(kind="rect", geom_id=gid, radiance=np.array(rad, np.float32), origin=o, edge_u=eu, edge_v=ev,
                                 normal=np.array([0, -1, 0], np.float32)))
    # Lights at infinity (own stream, so the rest of a seed's world stays what it was): a sun in about half of the
    # worlds, a uniform dome in a third, placed anywhere in the light list.
    inf = np.random.default_rng(seed ^ 0x5EED1234)
    if inf.random() < 0.5:
        sun = usda.distant_light((inf.uniform(-1, 1), inf.uniform(-1, -0.2), inf.uniform(-1, 1)),
                                 inf.uniform(0.5, 4, 3), [0.0, 0.53, 3.0, 25.0][int(inf.integers(0, 4))])
        d.lights.insert(int(inf.integers(0, len(d.lights) + 1)), sun)
    if inf.random() < 0.33:
        d.lights.insert(int(inf.integers(0, len(d.lights) + 1)), usda.dome_light(inf.uniform(0.05, 0.8, 3)))
    # Instances, some of them MOVING (own stream again: the worlds of seeds that draw none stay what they were): one or
    # two prototypes — a small triangle soup, sometimes with a sphere beside it in a nested scene — placed two to five
    # times with random scale / rotation, a third of the placements with a second transform (prim.rs:285-331). A moving
    # placement makes the scene draw a shutter time per camera sample (tracer.rs:579-583) and carry it along the path.
    mv = np.random.default_rng(seed ^ 0x0B1E55ED)
    if mv.random() < 0.5:
        for _ in range(int(mv.integers(1, 3))):
            pv, pi = _soup(mv, int(mv.integers(6, 40)), 0.6, 0.35)
            d.protos.append(dict(verts=pv, idx=pi))
            proto = len(d.protos) - 1
            if mv.random() < 0.4:  # a nested scene: the soup and a sphere, placed once each (instance depth 2)
                d.protos.append(dict(radius=float(mv.uniform(0.15, 0.4))))
                d.protos.append(dict(instances=[dict(proto=proto, l2w=_affine12(np.eye(3, dtype=np.float32), (0, 0, 0)), mask=0xFFFFFFFF),
                                                dict(proto=proto + 1, l2w=_affine12(np.eye(3, dtype=np.float32), (0.0, 0.9, 0.0)), mask=0xFFFFFFFF)]))
                proto = len(d.protos) - 1
            mat = _rand_material(mv)
            for _k in range(int(mv.integers(2, 6))):
                m = (_rot(mv) * f32(mv.uniform(0.5, 1.6))).astype(np.float32)
                t = np.array([mv.uniform(-3.5, 3.5), mv.uniform(0.6, 2.5), mv.uniform(-3.5, 3.5)], dtype=np.float32)
                g = dict(kind="instance", proto=proto, l2w=_affine12(m, t), mask=0xFFFFFFFF, material=mat, name="inst")
                if mv.random() < 0.34:
                    g["l2w_end"] = _affine12(m, (t + mv.uniform(-0.8, 0.8, 3)).astype(np.float32))
                d.geoms.append(g)
    lookfrom = np.array([rng.uniform(-2, 2), rng.uniform(1.5, 4), rng.uniform(6, 9)], dtype=np.float32)
    d.camera = dict(lookfrom=lookfrom, lookat=np.array([0, 1, 0], np.float32), vup=np.array([0, 1, 0], np.float32),
                    vfov_deg=f32(rng.uniform(30, 70)), aspect=f32(f32(width) / f32(height)),
                    aperture=f32(rng.uniform(0, 0.3)) if rng.random() < 0.5 else f32(0.0), focus_dist=f32(rng.uniform(5, 9)))
    d.settings = dict(usda.DEFAULTS, strategy=["power", "balance", "light", "bsdf"][int(rng.integers(0, 4))],
                      filter=["triangle", "box"][int(rng.integers(0, 2))], filter_radius=float(rng.uniform(0.5, 1.5)),
                      width=width, height=height, max_depth=int(rng.integers(3, 10)), frame=int(rng.integers(0, 5)))
    return d
