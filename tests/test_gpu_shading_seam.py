"""The shading seam as callable functions, on the GPU, function by function:
crt_material_scatter_n / crt_material_eval_n / crt_material_emitted_n (material.rs:40-45, :56-74, :112-115 ->
openpbr.rs:1026-1158, :1211-1218) and crt_light_sample_n / crt_light_pdf_n / crt_light_escaped_n (light.rs:126-146,
:180-213) through the C ABI, 100 000 seeded random calls per lobe class — base, metal, anisotropic, coat, fuzz, thin
film, rough transmission, dispersion, thin wall, subsurface / interior media, everything at once, Emissive — and all
four light kinds, compared BIT FOR BIT with the oracle's functions (tolerance zero; a NaN on both sides is equal).
Before this test a26-a33 were compared with the oracle only through whole renders, where a wrong pdf term that cancels
in value * cos / pdf on the scenes at hand would pass."""
import numpy as np
import pytest

import seam_cases as sc

pytestmark = pytest.mark.gpu
N = 100000


@pytest.fixture(scope="module")
def oracle():
    return sc.oracle_drivers()


def _host(d_out, dtype):
    return d_out.cpu().numpy().view(dtype)


@pytest.mark.parametrize("cls", sc.CLASSES)
def test_material_scatter_eval_emitted_match_the_oracle(crt, oracle, cls):
    rng = np.random.default_rng(2000 + sc.CLASSES.index(cls))
    mats = sc.materials(cls, 257, rng)
    q = sc.shade_queries(N, len(mats), rng)
    table = crt.shading.DeviceMaterials([crt.CrtMaterial.from_buffer_copy(m.tobytes()) for m in mats])
    d_q = crt.shading.to_device(q)
    got = {"scatter": _host(table.scatter_importance(d_q), sc.SCATTER_SAMPLE),
           "eval": _host(table.eval(d_q), sc.BSDF_EVAL),
           "emitted": _host(table.emitted_directional(d_q), np.float32).reshape(-1, 3)}
    for name, g in got.items():
        want = getattr(oracle, name)(mats, q)
        bad = sc.mismatches(g, want)
        assert len(bad) == 0, (cls, name, len(bad), g[bad[:2]], want[bad[:2]], q[bad[:2]])
    s = got["scatter"]
    if cls == "emissive":
        assert s["some"].sum() == 0 and got["eval"]["some"].sum() == 0
    else:
        assert s["some"].mean() > 0.5 and got["eval"]["some"].mean() > 0.8 and (s["pdf"][s["some"] == 1] >= 1e-4).all()


def test_light_sample_pdf_escaped_match_the_oracle(crt, oracle):
    rng = np.random.default_rng(88)
    table = sc.lights(rng)
    q = sc.light_queries(4 * N, table, rng)
    dl = crt.shading.DeviceLights((crt.CrtLight * len(table)).from_buffer_copy(table.tobytes()))
    d_q = crt.shading.to_device(q)
    got = {"light_sample": _host(dl.sample_li(d_q), sc.LIGHT_SAMPLE), "light_pdf": _host(dl.pdf_at_point(d_q), np.float32),
           "light_escaped": _host(dl.escaped(d_q), sc.LIGHT_SAMPLE)}
    for name, g in got.items():
        want = getattr(oracle, name)(table, q)
        bad = sc.mismatches(g, want)
        assert len(bad) == 0, (name, len(bad), g[bad[:2]], want[bad[:2]], q[bad[:2]])
    kinds = table["kind"][q["light"]]
    assert got["light_sample"]["some"].all() and np.isinf(got["light_sample"]["distance"][kinds >= 2]).all()
    assert got["light_escaped"]["some"][kinds == 3].all() and not got["light_escaped"]["some"][kinds < 2].any()


def test_a_host_integrator_vertex_on_the_seam_matches_the_oracle_render(crt, oracle):
    """What the seam is for: ONE path vertex of the reference's trace_path (tracer.rs:1321-1523) assembled on the host
    from crt_intersect_n + the Material / Light functions — camera rays of veach_mis, the hit record built as
    World::intersect does (rt_world.rs:207-232), NEE sample, BSDF eval, BSDF sample — agrees bit for bit with the same
    vertex assembled from the oracle's functions on the oracle's hit records."""
    import os
    import ora
    import ora_world
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r, desc = crt.load_usda(os.path.join(root, "scenes", "veach_mis.usda"), 96, 54, 8)
    o = ora_world.OracleRenderer(desc, crt.usda, max_depth=8)
    rng = np.random.default_rng(9)
    n = 8192
    cam = desc.camera
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = np.asarray(cam["lookfrom"], dtype=np.float32)
    d = (np.asarray(cam["lookat"], dtype=np.float32) - rays[0, 0:3])[None, :] + rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    rays[:, 3:6] = d
    rays[:, 7] = np.array([crt.MASK_CAMERA], dtype=np.uint32).view(np.float32)[0]
    hits = crt.hits_to_host(r.scene.intersect_n(crt.rays_to_device(rays), 0.001, float("inf")))
    torch.cuda.synchronize()
    hf, ids, front = o.scene.intersect_n(rays, 0.001, float("inf"))
    hit = hits["geom_id"] != 0xFFFFFFFF
    assert hit.sum() > n // 4 and np.array_equal(hits["geom_id"], ids[:, 0])
    q = np.zeros(int(hit.sum()), dtype=sc.SHADE_QUERY)
    q["ray_dir"] = rays[hit, 3:6]; q["material"] = hits["geom_id"][hit]
    q["t"] = hits["t"][hit]
    q["p"] = rays[hit, 0:3] + rays[hit, 3:6] * hits["t"][hit][:, None]  # ray.at(t): o + t * d, one rounding per op
    q["normal"] = hits["normal"][hit]; q["front_face"] = hits["front_face"][hit]
    q["sampler_pattern"] = rng.integers(0, 2 ** 32, size=len(q), dtype=np.uint64).astype(np.uint32)
    q["sampler_index"] = rng.integers(0, 64, size=len(q), dtype=np.uint32)
    lights = crt.make_lights(desc.lights)
    ltab = np.frombuffer(bytes(lights), dtype=sc.LIGHT)[:len(desc.lights)]
    lq = np.zeros(len(q), dtype=sc.LIGHT_QUERY)
    lq["from"] = q["p"]; lq["light"] = rng.integers(0, len(ltab), size=len(q), dtype=np.uint32)
    lq["u"] = rng.uniform(size=len(q)).astype(np.float32); lq["v"] = rng.uniform(size=len(q)).astype(np.float32)
    dl = crt.shading.DeviceLights(lights)
    ls = _host(dl.sample_li(crt.shading.to_device(lq)), sc.LIGHT_SAMPLE)
    q["wi"] = ls["direction"]
    mats = np.frombuffer(bytes(r._mats), dtype=sc.MATERIAL)[:len(r._mats)]
    table = crt.shading.DeviceMaterials(list(r._mats))
    d_q = crt.shading.to_device(q)
    ev, scat = _host(table.eval(d_q), sc.BSDF_EVAL), _host(table.scatter_importance(d_q), sc.SCATTER_SAMPLE)
    assert len(sc.mismatches(ls, oracle.light_sample(ltab, lq))) == 0
    assert len(sc.mismatches(ev, oracle.eval(mats, q))) == 0 and len(sc.mismatches(scat, oracle.scatter(mats, q))) == 0
    assert ev["some"].sum() > 0 and scat["some"].sum() > 0
