"""Randomised differential test: random scenes (triangle soups incl. degenerate and coincident triangles, spheres,
nested / scaled / rotated / moving instances, per-geometry masks) and random rays (random masks, shutter times, bounded
ranges), HIP kernels through the C ABI against the oracle, bit for bit. Fixed seeds: deterministic."""
import numpy as np
import pytest

import fuzz_scenes
import ora

pytestmark = pytest.mark.gpu
f32 = np.float32


import os

_EXTRA = int(os.environ.get("CRT_FUZZ_EXTRA", "0"))  # CRT_FUZZ_EXTRA=N: N more seeds per test (soak runs)
_BASE = int(os.environ.get("CRT_FUZZ_BASE", "0"))  # CRT_FUZZ_BASE=B: the extra seeds start B later (a soak run on seeds no earlier one saw)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18] + list(range(1000 + _BASE, 1000 + _BASE + _EXTRA)))
def test_random_scene_matches_oracle_bitwise(crt, seed):
    import torch
    recipe = fuzz_scenes.recipe(seed)
    o_scene, _ok = fuzz_scenes.build(ora, recipe)
    p_scene, _pk = fuzz_scenes.build(crt, recipe)
    rng = np.random.default_rng(1000 + seed)
    n = 6000
    o = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    d = tgt - o
    d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True)  # half normalised, half not (ray.rs: unnormalised allowed)
    d[::97, 1] = 0.0   # axis-parallel components exercise safe_inv3
    d[::89, 0] = 0.0
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3:6] = o, d
    rays[:, 6] = rng.choice(np.array([0.0, 0.25, 0.5, 1.0], dtype=np.float32), n)
    rays[:, 7] = rng.choice(np.array([0xFFFFFFFF, 1, 2, 4, 6], dtype=np.uint32), n).view(np.float32)
    d_rays = crt.rays_to_device(rays)
    for t_min, t_max in ((0.001, float("inf")), (0.5, 6.0)):
        hf, ids, front = o_scene.intersect_n(rays, t_min, t_max)
        occ = o_scene.occluded_n(rays, t_min, t_max)
        hits = crt.hits_to_host(p_scene.intersect_n(d_rays, t_min, t_max))
        got_occ = p_scene.occluded_n(d_rays, t_min, t_max)
        torch.cuda.synchronize()
        hit = ids[:, 0] != 0xFFFFFFFF
        assert hit.sum() > n // 50
        assert np.array_equal(hits["geom_id"], ids[:, 0]) and np.array_equal(hits["prim_id"], ids[:, 1])
        assert np.array_equal(hits["t"][hit].view(np.uint32), hf[hit, 0].view(np.uint32))
        assert np.array_equal(hits["normal"][hit].view(np.uint32), hf[hit, 1:4].view(np.uint32))
        assert np.array_equal(hits["u"][hit].view(np.uint32), hf[hit, 4].view(np.uint32))
        assert np.array_equal(hits["v"][hit].view(np.uint32), hf[hit, 5].view(np.uint32))
        assert np.array_equal(hits["front_face"][hit], front[hit].astype(np.uint32))
        assert np.array_equal(got_occ.cpu().numpy().astype(np.uint8), occ)


_CENSUS = {}  # engine instance the renderer's selection names (0 three-wave, 1 four-wave flat copy, 2 four-wave direct copy) -> worlds


@pytest.fixture(scope="module", autouse=True)
def _engine_census():
    """CRT_FUZZ_CENSUS=path: which engine instances a soak run's worlds ran on, written when the module is done."""
    yield
    path = os.environ.get("CRT_FUZZ_CENSUS")
    if path and _CENSUS:
        with open(path, "a") as f:
            f.write("worlds by engine instance (0 three-wave, 1 four-wave flat, 2 four-wave direct): %s\n" % dict(sorted(_CENSUS.items())))


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106, 107, 108, 109, 110] + list(range(2000 + _BASE, 2000 + _BASE + _EXTRA)))
def test_random_world_renders_identically(crt, seed):
    """Random OpenPBR materials (every lobe, interior media, thin walls, dispersion, thin film, emission), sphere and
    rect lights, all four sampling strategies, both filters, thin-lens cameras: image and counters identical."""
    import torch
    import ora_world
    desc = fuzz_scenes.random_world(crt.usda, seed)
    scene, mats, protos = crt.usda.build_world(desc, crt, crt.default_material)
    s = desc.settings
    settings = crt.RenderSettings(s["width"], s["height"], s["max_depth"], s["frame"], s["strategy"], s["filter"],
                                  s["filter_radius"], 0.0)
    r = crt.Renderer(scene, mats, desc.lights, crt.make_camera(**desc.camera), settings)
    r.render_samples(0, 6)
    torch.cuda.synchronize()
    img, st = r.image(), r.stats()
    w = scene.engine_select(-3)["wide"]  # the renderer's selection in this process (environment included)
    _CENSUS[w] = _CENSUS.get(w, 0) + 1
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda).render(6, forward=1)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), (seed, f, getattr(st, f), getattr(ost, f))
    assert np.isfinite(oimg).all()
    bad = np.argwhere(img.view(np.uint32) != oimg.view(np.uint32))
    assert bad.shape[0] == 0, (seed, bad.shape[0], bad[:3])


@pytest.mark.parametrize("split", ["6", "10", "wide"])
def test_both_lds_splits_of_the_engine_match_the_oracle(crt, split, monkeypatch):
    """The traversal engine divides its LDS between stack entries and the node window per scene (6 + 72 nodes, or
    10 + 16 for instance-heavy scenes, scene.cpp; 4 + 8 in the kernels that run four workgroups per CU). Forcing each
    split (CRT_POOL_STACK_RT, read when the device image is built; CRT_WIDE) over scenes with nested instances: hits and
    occlusion flags identical to the oracle."""
    import torch
    if split == "wide":  # the four-workgroups-per-CU kernels (4 LDS stack entries), which flat scenes get by default
        monkeypatch.setenv("CRT_WIDE", "1")
    else:
        monkeypatch.setenv("CRT_WIDE", "0")
        monkeypatch.setenv("CRT_POOL_STACK_RT", split)
    for seed in (13, 16, 18):
        recipe = fuzz_scenes.recipe(seed)
        o_scene, _ok = fuzz_scenes.build(ora, recipe)
        p_scene, _pk = fuzz_scenes.build(crt, recipe)
        rng = np.random.default_rng(77 + seed)
        n = 4000
        rays = np.zeros((n, 8), dtype=np.float32)
        rays[:, 0:3] = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
        rays[:, 3:6] = (rng.uniform(-5, 5, (n, 3)) - rays[:, 0:3]).astype(np.float32)
        rays[:, 6] = rng.choice(np.array([0.0, 0.5, 1.0], dtype=np.float32), n)
        rays[:, 7] = np.full(n, 0xFFFFFFFF, np.uint32).view(np.float32)
        d_rays = crt.rays_to_device(rays)
        hf, ids, _front = o_scene.intersect_n(rays, 0.001, float("inf"))
        occ = o_scene.occluded_n(rays, 0.001, float("inf"))
        hits = crt.hits_to_host(p_scene.intersect_n(d_rays, 0.001, float("inf")))
        got_occ = p_scene.occluded_n(d_rays, 0.001, float("inf"))
        torch.cuda.synchronize()
        hit = ids[:, 0] != 0xFFFFFFFF
        assert np.array_equal(hits["geom_id"], ids[:, 0]) and np.array_equal(hits["prim_id"], ids[:, 1])
        assert np.array_equal(hits["t"][hit].view(np.uint32), hf[hit, 0].view(np.uint32))
        assert np.array_equal(got_occ.cpu().numpy().astype(np.uint8), occ)
