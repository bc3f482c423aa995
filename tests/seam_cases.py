"""Seeded inputs for the function-level parity tests of the shading seam: materials per lobe class,
Material-method queries and Light-method queries as numpy record arrays in the C ABI's layouts (include/crt.h), plus the
oracle's side of the comparison (oracle/ora_shade.c, ora_t_*_n drivers). Used by the GPU test (tests/
test_gpu_shading_seam.py: crt_material_*_n / crt_light_*_n vs the oracle) and by its CPU twin (tests/
test_shading_seam_host.py: the SAME device functions compiled as host C++ vs the oracle)."""
import ctypes as C

import numpy as np

SHADE_QUERY = np.dtype([("ray_dir", np.float32, 3), ("material", np.uint32), ("p", np.float32, 3), ("t", np.float32),
                        ("normal", np.float32, 3), ("front_face", np.uint32), ("wi", np.float32, 3),
                        ("cos_theta_o", np.float32), ("sampler_pattern", np.uint32), ("sampler_index", np.uint32),
                        ("_pad", np.uint32, 2)])
SCATTER_SAMPLE = np.dtype([("origin", np.float32, 3), ("some", np.uint32), ("dir", np.float32, 3), ("pdf", np.float32),
                           ("value", np.float32, 3), ("flags", np.uint32)])
BSDF_EVAL = np.dtype([("value", np.float32, 3), ("pdf", np.float32), ("some", np.uint32), ("_pad", np.uint32, 3)])
LIGHT_QUERY = np.dtype([("from", np.float32, 3), ("light", np.uint32), ("u", np.float32), ("v", np.float32),
                        ("_pad", np.uint32, 2), ("point", np.float32, 3), ("_pad2", np.uint32)])
LIGHT_SAMPLE = np.dtype([("direction", np.float32, 3), ("distance", np.float32), ("radiance", np.float32, 3),
                         ("pdf", np.float32), ("some", np.uint32), ("_pad", np.uint32, 3)])

# CrtMaterial / OraMaterial as a numpy record (same 224-byte layout on both sides)
MATERIAL = np.dtype([
    ("kind", np.uint32), ("thin_walled", np.uint32),
    ("base_weight", np.float32), ("base_color", np.float32, 3), ("base_diffuse_roughness", np.float32),
    ("base_metalness", np.float32),
    ("specular_weight", np.float32), ("specular_color", np.float32, 3), ("specular_roughness", np.float32),
    ("specular_ior", np.float32), ("specular_roughness_anisotropy", np.float32),
    ("transmission_weight", np.float32), ("transmission_color", np.float32, 3), ("transmission_depth", np.float32),
    ("transmission_scatter", np.float32, 3), ("transmission_scatter_anisotropy", np.float32),
    ("transmission_dispersion_scale", np.float32), ("transmission_dispersion_abbe_number", np.float32),
    ("subsurface_weight", np.float32), ("subsurface_color", np.float32, 3), ("subsurface_radius", np.float32),
    ("subsurface_radius_scale", np.float32, 3), ("subsurface_scatter_anisotropy", np.float32),
    ("fuzz_weight", np.float32), ("fuzz_color", np.float32, 3), ("fuzz_roughness", np.float32),
    ("coat_weight", np.float32), ("coat_color", np.float32, 3), ("coat_roughness", np.float32),
    ("coat_roughness_anisotropy", np.float32), ("coat_ior", np.float32), ("coat_darkening", np.float32),
    ("thin_film_weight", np.float32), ("thin_film_thickness", np.float32), ("thin_film_ior", np.float32),
    ("emission_luminance", np.float32), ("emission_color", np.float32, 3),
    ("geometry_opacity", np.float32),
])
LIGHT = np.dtype([("kind", np.uint32), ("geom_id", np.uint32), ("radiance", np.float32, 3), ("center", np.float32, 3),
                  ("radius", np.float32), ("origin", np.float32, 3), ("edge_u", np.float32, 3), ("edge_v", np.float32, 3),
                  ("normal", np.float32, 3)])
assert MATERIAL.itemsize == 224 and LIGHT.itemsize == 84

# Lobe classes of the OpenPBR übershader (openpbr.rs:1026-1136) + Emissive: what each class switches on.
CLASSES = ("base", "metal", "anisotropic", "coat", "fuzz", "thin_film", "transmission", "dispersion", "thin_wall",
           "subsurface_medium", "everything", "emissive")


def _unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def materials(cls, n, rng):
    """n random materials of one lobe class (OpenPBR::default, openpbr.rs:130-173, with the class's inputs randomised)."""
    u = lambda lo=0.0, hi=1.0, shape=None: rng.uniform(lo, hi, size=(n,) if shape is None else (n, shape)).astype(np.float32)
    m = np.zeros(n, dtype=MATERIAL)
    m["base_weight"] = u(0.2, 1.0); m["base_color"] = u(0.02, 1.0, 3); m["base_diffuse_roughness"] = u()
    m["specular_weight"] = u(0.0, 1.0); m["specular_color"] = u(0.2, 1.0, 3); m["specular_roughness"] = u(0.02, 1.0)
    m["specular_ior"] = u(1.05, 2.5)
    m["transmission_color"] = u(0.05, 1.0, 3); m["transmission_dispersion_abbe_number"] = u(10.0, 60.0)
    m["subsurface_color"] = u(0.05, 1.0, 3); m["subsurface_radius"] = u(0.05, 2.0); m["subsurface_radius_scale"] = u(0.1, 1.0, 3)
    m["fuzz_color"] = u(0.05, 1.0, 3); m["fuzz_roughness"] = u(0.02, 1.0)
    m["coat_color"] = u(0.05, 1.0, 3); m["coat_roughness"] = u(0.0, 1.0); m["coat_ior"] = u(1.1, 2.0)
    m["thin_film_thickness"] = u(0.05, 1.5); m["thin_film_ior"] = u(1.1, 2.0)
    m["emission_color"] = u(0.0, 4.0, 3); m["emission_luminance"] = np.where(rng.uniform(size=n) < 0.3, u(0.0, 3.0), 0.0)
    m["geometry_opacity"] = 1.0
    on = lambda p=0.7: (rng.uniform(size=n) < p)
    if cls in ("metal", "everything"):
        m["base_metalness"] = np.where(on(), u(0.1, 1.0), np.float32(1.0 if cls == "metal" else 0.0))
    if cls in ("anisotropic", "everything"):
        m["specular_roughness_anisotropy"] = u(0.0, 1.0); m["coat_roughness_anisotropy"] = u(0.0, 1.0)
    if cls in ("coat", "everything"):
        m["coat_weight"] = u(0.05, 1.0); m["coat_darkening"] = u(0.0, 1.0)
    if cls in ("fuzz", "everything"):
        m["fuzz_weight"] = u(0.05, 1.0)
    if cls in ("thin_film", "everything"):
        m["thin_film_weight"] = u(0.05, 1.0)
        if cls == "thin_film":
            m["base_metalness"] = np.where(on(0.5), u(0.0, 1.0), np.float32(0.0))
            m["coat_weight"] = np.where(on(0.3), u(0.05, 1.0), np.float32(0.0))
    if cls in ("transmission", "dispersion", "thin_wall", "subsurface_medium", "everything"):
        m["transmission_weight"] = u(0.05, 1.0)
        m["transmission_depth"] = np.where(on(0.5), u(0.05, 3.0), np.float32(0.0))
        m["transmission_scatter"] = np.where(on(0.4)[:, None], u(0.0, 1.0, 3), np.float32(0.0))
        m["transmission_scatter_anisotropy"] = u(-0.9, 0.9)
    if cls in ("dispersion", "everything"):
        m["transmission_dispersion_scale"] = u(0.05, 1.0)
    if cls == "thin_wall" or cls == "everything":
        m["thin_walled"] = (on(1.0 if cls == "thin_wall" else 0.3)).astype(np.uint32)
    if cls in ("subsurface_medium", "everything"):
        m["subsurface_weight"] = u(0.05, 1.0); m["subsurface_scatter_anisotropy"] = u(-0.9, 0.9)
        if cls == "subsurface_medium":
            m["transmission_weight"] = np.where(on(0.5), u(0.0, 1.0), np.float32(0.0))
    if cls == "emissive":
        m["kind"] = 1
    return m


def shade_queries(n, n_materials, rng):
    """Random Material-method calls: a hit record whose normal faces the ray (one in 16 does not: the None arms), a
    direction to evaluate anywhere on the sphere, a sampler domain."""
    q = np.zeros(n, dtype=SHADE_QUERY)
    nrm = _unit(rng, n)
    rd = (_unit(rng, n) * rng.uniform(0.3, 3.0, size=(n, 1))).astype(np.float32)  # unnormalised directions are allowed
    facing = np.einsum("ij,ij->i", rd, nrm) < 0
    flip = facing ^ (rng.uniform(size=n) < 1.0 / 16.0)
    rd[~flip] *= np.float32(-1.0)
    q["ray_dir"] = rd; q["normal"] = nrm
    q["material"] = rng.integers(0, n_materials, size=n, dtype=np.uint32)
    q["p"] = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32); q["t"] = rng.uniform(0.01, 20.0, size=n).astype(np.float32)
    q["front_face"] = rng.integers(0, 2, size=n, dtype=np.uint32)
    q["wi"] = _unit(rng, n)
    q["cos_theta_o"] = rng.uniform(0.0, 1.0, size=n).astype(np.float32)
    q["sampler_pattern"] = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    q["sampler_index"] = rng.integers(0, 4096, size=n, dtype=np.uint32)
    return q


def lights(rng, n_each=8):
    """Sphere, rect, distant and dome lights (derived record forms: include/crt.h, CrtLight)."""
    out = []
    for k in range(n_each):
        l = np.zeros(1, dtype=LIGHT)[0]
        l["kind"] = 0; l["geom_id"] = k; l["radiance"] = rng.uniform(0.5, 20, 3)
        l["center"] = rng.uniform(-4, 4, 3); l["radius"] = rng.uniform(0.05, 1.5)
        out.append(l)
    for k in range(n_each):
        l = np.zeros(1, dtype=LIGHT)[0]
        eu, ev = rng.uniform(-2, 2, 3).astype(np.float32), rng.uniform(-2, 2, 3).astype(np.float32)
        nr = np.cross(eu, ev).astype(np.float32)
        l["kind"] = 1; l["geom_id"] = 100 + k; l["radiance"] = rng.uniform(0.5, 20, 3)
        l["origin"] = rng.uniform(-4, 4, 3); l["edge_u"] = eu; l["edge_v"] = ev
        l["normal"] = nr / np.float32(np.sqrt(np.float32(nr @ nr)))
        out.append(l)
    for k in range(n_each):
        l = np.zeros(1, dtype=LIGHT)[0]
        d = rng.normal(size=3).astype(np.float32); d = d / np.float32(np.linalg.norm(d))
        cos_half = np.float32(np.cos(np.radians(rng.uniform(0.2, 25.0))))
        l["kind"] = 2; l["geom_id"] = 0xFFFFFFFF; l["radiance"] = rng.uniform(0.5, 5, 3); l["normal"] = d
        l["radius"] = cos_half; l["center"][0] = np.float32(2.0 * np.pi) * (np.float32(1.0) - cos_half)
        out.append(l)
    for k in range(2):
        l = np.zeros(1, dtype=LIGHT)[0]
        l["kind"] = 3; l["geom_id"] = 0xFFFFFFFF; l["radiance"] = rng.uniform(0.1, 2, 3)
        out.append(l)
    return np.array(out, dtype=LIGHT)


def light_queries(n, table, rng):
    q = np.zeros(n, dtype=LIGHT_QUERY)
    q["from"] = rng.uniform(-6, 6, size=(n, 3)).astype(np.float32)
    q["light"] = rng.integers(0, len(table), size=n, dtype=np.uint32)
    q["u"] = rng.uniform(size=n).astype(np.float32); q["v"] = rng.uniform(size=n).astype(np.float32)
    # pdf_at_point: a point ON the light for the area lights (what a bounce ray hits); escaped: a unit direction, half of
    # them inside the distant lights' cones
    pts = _unit(rng, n)
    kinds = table["kind"][q["light"]]
    for i in np.nonzero(kinds == 0)[0]:
        l = table[q["light"][i]]
        pts[i] = l["center"] + pts[i] * l["radius"]
    for i in np.nonzero(kinds == 1)[0]:
        l = table[q["light"][i]]
        pts[i] = l["origin"] + l["edge_u"] * np.float32(rng.uniform()) + l["edge_v"] * np.float32(rng.uniform())
    for i in np.nonzero(kinds == 2)[0][::2]:
        l = table[q["light"][i]]
        d = -l["normal"] + np.float32(0.05) * pts[i]
        pts[i] = d / np.float32(np.linalg.norm(d))
    q["point"] = pts.astype(np.float32)
    return q


class Drivers:
    """One side of the comparison: six batched functions (records in, records out) from a shared library that exports
    them under <prefix>_scatter_n ... — the oracle (liboracle.so, prefix ora_t) or the host-compiled device code
    (tests/host_shade, prefix host)."""

    def __init__(self, cdll, prefix):
        self.L, self.prefix = cdll, prefix
        for name in ("scatter_n", "eval_n", "emitted_n", "light_sample_n", "light_pdf_n", "light_escaped_n"):
            f = getattr(cdll, "%s_%s" % (prefix, name))
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]

    def _run(self, name, table, queries, out_dtype, per=1):
        table, queries = np.ascontiguousarray(table), np.ascontiguousarray(queries)
        out = np.zeros(len(queries) * per, dtype=out_dtype)
        getattr(self.L, "%s_%s" % (self.prefix, name))(table.ctypes.data, len(table), queries.ctypes.data, len(queries), out.ctypes.data)
        return out

    def scatter(self, mats, q): return self._run("scatter_n", mats, q, SCATTER_SAMPLE)
    def eval(self, mats, q): return self._run("eval_n", mats, q, BSDF_EVAL)
    def emitted(self, mats, q): return self._run("emitted_n", mats, q, np.float32, 3).reshape(-1, 3)
    def light_sample(self, ls, q): return self._run("light_sample_n", ls, q, LIGHT_SAMPLE)
    def light_pdf(self, ls, q): return self._run("light_pdf_n", ls, q, np.float32)
    def light_escaped(self, ls, q): return self._run("light_escaped_n", ls, q, LIGHT_SAMPLE)


def oracle_drivers():
    import ora
    return Drivers(ora.lib(), "ora_t")


def mismatches(got, want):
    """Records (or rows) whose bits differ; a NaN on both sides counts as equal whatever its payload (x86 and gfx950
    produce different default NaNs; the inputs above are built to avoid NaNs in the first place)."""
    g = np.ascontiguousarray(got).view(np.uint32).reshape(len(got), -1)
    w = np.ascontiguousarray(want).view(np.uint32).reshape(len(want), -1)
    gf, wf = g.view(np.float32), w.view(np.float32)
    same = (g == w) | (np.isnan(gf) & np.isnan(wf))
    return np.nonzero(~same.all(axis=1))[0]
