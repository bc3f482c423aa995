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


