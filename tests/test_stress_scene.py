"""The reference's BVH stress scene (scripts/gen_stress_scene.py:1-150; render time published at docs/simd.md:221) as
this repository carries it: scenes/stress.usda.xz, the generator's own output, packed (scenes/make_stress.py). CPU
checks: the importer sees what the generator says it wrote, and the oracle traces shadow rays on it."""
import os

import numpy as np
import pytest

import ora
import ora_world

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stress(crt):
    return crt.usda.load(crt.scene_path("stress"))


def test_stress_scene_is_what_the_generator_describes(crt, stress):
    d = stress
    s = d.settings  # gen_stress_scene.py:136-146
    assert (s["width"], s["height"], s["spp"], s["max_depth"], s["min_spp"], s["variance"]) == (640, 360, 16, 6, 16, 0.0)
    inst = [g for g in d.geoms if g["kind"] == "instance"]
    mesh = [g for g in d.geoms if g["kind"] == "mesh"]
    # 36 prims authoring the same points / indices / material: the importer's dedup places ONE shared mesh 36 times
    # (usd_import.rs:1025-1089: n_place >= 2 -> instanced), everything else is baked to world space
    assert len(inst) == 36 and len({g["proto"] for g in inst}) == 1
    assert len(d.protos[inst[0]["proto"]]["idx"]) == 9800 and len(d.protos[inst[0]["proto"]]["verts"]) == 4902
    tris = sorted(len(g["idx"]) for g in mesh)
    assert tris == [2, 2, 2, 30000]  # floor, occluder slab, the RectLight's own quad, the shard field
    assert 36 * 9800 + 30000 + 4 == 382804  # the generator's "world-baked" count (gen_stress_scene.py:148-150)
    assert len(d.lights) == 1 and d.lights[0]["kind"] == "rect"
    assert np.allclose(d.lights[0]["radiance"], 6.0) and np.allclose(np.abs(d.lights[0]["edge_u"]).max(), 12.0)
    grey = inst[0]["material"]  # UsdPreviewSurface -> OpenPBR (usd_import.rs:2658-2700)
    assert np.allclose(grey["base_color"], (0.55, 0.55, 0.6)) and abs(float(grey["specular_roughness"]) - 0.4) < 1e-7
    xs = sorted({round(float(g["l2w"][9]), 4) for g in inst})  # translation x of the 6 x 6 grid, spacing 2.5
    assert xs == [-6.25, -3.75, -1.25, 1.25, 3.75, 6.25]


def test_oracle_traces_shadow_rays_through_the_occluder(crt, stress, oracle):
    """A small frame through the oracle's integrator: NEE is on (one RectLight), a good part of the shadow rays is
    blocked by the slab and the shards, and two runs give the same bits (threads do not change the result)."""
    o = ora_world.OracleRenderer(stress, crt.usda, width=None, height=None)
    idx = np.sort(np.random.default_rng(5).choice(640 * 360, 2048, replace=False)).astype(np.uint32)
    px, st = o.render_pixels(idx, 4, forward=1)
    assert st.shadow_rays > 0.5 * st.vertices and st.closest_hit >= st.camera_rays == 2048 * 4
    assert np.isfinite(px).all() and float(px.mean()) > 0.05
    px2, st2 = o.render_pixels(idx, 4, forward=1, threads=3)
    assert np.array_equal(px.view(np.uint32), px2.view(np.uint32)) and st2.shadow_rays == st.shadow_rays
