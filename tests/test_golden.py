"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must still produce them
(CPU), and the HIP path must produce them too (GPU) — fixed data, independent of the oracle build on the box."""
import os

import numpy as np
import pytest

import golden_inputs as gi
import ora
import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INF = float("inf")


def _g(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", list(scenes.ALL))
def test_oracle_traversal_matches_golden(name):
    make, extent = scenes.ALL[name]
    g = _g("traverse_" + name)
    s = make(ora)
    rays = gi.traverse_rays(name, extent)
    hf, ids, front = s.intersect_n(rays, 0.001, INF)
    assert np.array_equal(ids, g["ids"])
    hit = ids[:, 0] != 0xFFFFFFFF
    assert np.array_equal(hf.view(np.uint32)[hit], g["hit_bits"][hit])
    assert np.array_equal(front.astype(np.uint8)[hit], g["front"][hit])
    assert np.array_equal(s.occluded_n(rays, 0.001, INF).astype(np.uint8), g["occluded"])


@pytest.mark.parametrize("name,w,h,spp,depth", gi.RENDERS)
def test_oracle_render_matches_golden(crt, name, w, h, spp, depth):
    import ora_world
    g = _g("render_" + name)
    desc = crt.usda.load(os.path.join(ROOT, "scenes", name + ".usda"), w, h)
    img, st = ora_world.OracleRenderer(desc, crt.usda, max_depth=depth).render(spp, forward=1)
    assert np.array_equal(img.view(np.uint32), g["image_bits"])
    assert [getattr(st, f) for f, _ in ora.RayStats._fields_] == g["counters"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(scenes.ALL))
def test_gpu_traversal_matches_golden(crt, name):
    import torch
    make, extent = scenes.ALL[name]
    g = _g("traverse_" + name)
    s = make(crt)
    d_rays = crt.rays_to_device(gi.traverse_rays(name, extent))
    hits = crt.hits_to_host(s.intersect_n(d_rays, 0.001, INF))
    occ = s.occluded_n(d_rays, 0.001, INF)
    torch.cuda.synchronize()
    assert np.array_equal(hits["geom_id"], g["ids"][:, 0]) and np.array_equal(hits["prim_id"], g["ids"][:, 1])
    hit = g["ids"][:, 0] != 0xFFFFFFFF
    gb = g["hit_bits"]
    assert np.array_equal(hits["t"].view(np.uint32)[hit], gb[hit, 0])
    assert np.array_equal(hits["normal"].view(np.uint32)[hit], gb[hit, 1:4])
    assert np.array_equal(hits["u"].view(np.uint32)[hit], gb[hit, 4]) and np.array_equal(hits["v"].view(np.uint32)[hit], gb[hit, 5])
    assert np.array_equal(hits["front_face"][hit].astype(np.uint8), g["front"][hit])
    assert np.array_equal(occ.cpu().numpy().astype(np.uint8), g["occluded"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,spp,depth", gi.RENDERS)
def test_gpu_render_matches_golden(crt, name, w, h, spp, depth):
    import torch
    g = _g("render_" + name)
    r, _ = crt.load_usda(os.path.join(ROOT, "scenes", name + ".usda"), w, h, depth)
    r.render_samples(0, spp)
    torch.cuda.synchronize()
    assert np.array_equal(r.image().view(np.uint32), g["image_bits"])
    st = r.stats()
    assert [getattr(st, f) for f, _ in ora.RayStats._fields_] == g["counters"].tolist()


# ---- the shading seam's functions: result records of seeded calls, frozen ----
def _seam_sides(drivers):
    import seam_cases as sc
    g = _g("seam_calls")
    for cls in sc.CLASSES:
        mats, q = gi.seam_material_case(cls)
        for name in ("scatter", "eval", "emitted"):
            got = np.ascontiguousarray(getattr(drivers, name)(mats, q)).view(np.uint32).reshape(len(q), -1)
            assert len(sc.mismatches(got, g[cls + "_" + name].reshape(len(q), -1))) == 0, (cls, name)  # NaN == NaN, else bits
    table, q = gi.seam_light_case()
    for name in ("light_sample", "light_pdf", "light_escaped"):
        got = np.ascontiguousarray(getattr(drivers, name)(table, q)).view(np.uint32).reshape(len(q), -1)
        assert len(sc.mismatches(got, g[name].reshape(len(q), -1))) == 0, name


def test_oracle_seam_functions_match_golden():
    import seam_cases as sc
    _seam_sides(sc.oracle_drivers())


@pytest.mark.gpu
def test_gpu_seam_functions_match_golden(crt):
    import seam_cases as sc

    class Device:  # crt_material_*_n / crt_light_*_n on numpy records
        @staticmethod
        def _mats(mats):
            return crt.shading.DeviceMaterials([crt.CrtMaterial.from_buffer_copy(m.tobytes()) for m in mats])

        @staticmethod
        def _lights(table):
            return crt.shading.DeviceLights((crt.CrtLight * len(table)).from_buffer_copy(table.tobytes()))

        def scatter(self, mats, q): return self._mats(mats).scatter_importance(crt.shading.to_device(q)).cpu().numpy()
        def eval(self, mats, q): return self._mats(mats).eval(crt.shading.to_device(q)).cpu().numpy()
        def emitted(self, mats, q): return self._mats(mats).emitted_directional(crt.shading.to_device(q)).cpu().numpy()
        def light_sample(self, t, q): return self._lights(t).sample_li(crt.shading.to_device(q)).cpu().numpy()
        def light_pdf(self, t, q): return self._lights(t).pdf_at_point(crt.shading.to_device(q)).cpu().numpy()
        def light_escaped(self, t, q): return self._lights(t).escaped(crt.shading.to_device(q)).cpu().numpy()

    _seam_sides(Device())
