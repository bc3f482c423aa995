"""The reference's BVH stress scene through the wavefront integrator at its authored settings (640x360, 16 spp, depth 6,
variance 0 — gen_stress_scene.py:136-146; the configuration docs/simd.md:221 times): 36 placements of a shared
9 800-triangle mesh, a 30 000-triangle shard field built with spatial splits (67 k Tri4 packets), one RectLight whose
shadow rays an occluder slab blocks — the path-traced workload that leaves the caches and calls `occluded`."""
import os

import numpy as np
import pytest

import ora
import ora_world

pytestmark = pytest.mark.gpu


def test_stress_scene_authored_settings_identical_to_oracle(crt):
    import torch
    r, desc = crt.load_usda(crt.scene_path("stress"))
    s = desc.settings
    assert (r.settings.width, r.settings.height, r.settings.max_depth) == (640, 360, 6)
    r.render_samples(0, s["spp"])
    torch.cuda.synchronize()
    img, st = r.image(), r.stats()
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda).render(s["spp"], forward=1)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), (f, getattr(st, f), getattr(ost, f))
    assert st.shadow_rays > 0 and st.camera_rays == 640 * 360 * 16
    bad = np.argwhere(img.view(np.uint32) != oimg.view(np.uint32))
    assert bad.shape[0] == 0, f"{bad.shape[0]} differing components, first {bad[:3]}"


def test_stress_scene_traversal_counters_identical_to_oracle(crt):
    """The traversal counters of both query kinds (bvh.rs:39-57: queries, node visits, leaves, packets, scalar primitives
    per level; accepted hits and instance descents) of one 2-spp batch at 320x180 — k_extend's launches against the
    oracle's closest-hit queries, k_shadow's against its any-hit queries, count for count."""
    r, desc = crt.load_usda(crt.scene_path("stress"), 320, 180)
    ext, sh = r.render_samples_stats(0, 2)
    o = ora_world.OracleRenderer(desc, crt.usda)
    _img, ost, closest, anyhit = o.render_serial_trav(2, forward=1)
    assert int(ext.rays) == ost.closest_hit and int(sh.rays) == ost.shadow_rays and ost.shadow_rays > 0
    for dev, ora_st, kind in ((ext, closest, "closest"), (sh, anyhit, "any")):
        for f in ("queries", "nodes", "leaves", "packets", "prims"):
            assert list(getattr(dev, f)) == list(getattr(ora_st, f)), (kind, f, list(getattr(dev, f)), list(getattr(ora_st, f)))
        assert int(dev.instance_descents) == int(ora_st.instance_descents), kind
    assert int(ext.accepted_hits) == int(closest.accepted_hits) and int(ext.instance_descents) > 0


def test_stress_scene_bench_shape_subset(crt):
    """The bench's shape for this scene (1920x1080, 64 spp in one batch, depth 6: 132.7 M paths) on a random subset of
    pixels against the oracle (per-pixel independence, tracer.rs:543, :559-560)."""
    import torch
    w, h, spp = 1920, 1080, 64
    r, desc = crt.load_usda(crt.scene_path("stress"), w, h)
    r.render_samples(0, spp)
    torch.cuda.synchronize()
    idx = np.sort(np.random.default_rng(11).choice(w * h, 2048, replace=False)).astype(np.uint32)
    opx, _ = ora_world.OracleRenderer(desc, crt.usda).render_pixels(idx, spp, forward=1)
    assert np.array_equal(r.image().reshape(-1, 3)[idx].view(np.uint32), opx.view(np.uint32))
