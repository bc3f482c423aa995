"""Parity at the BASELINE configs' OWN shapes (BASELINE.json configs[1..4]): resolution, authored depth and the
bench's batch shape, so the segment capacities, grid limits and staging-film offsets the bench exercises are the
ones compared with the oracle.

Two kinds of comparison per config, both bit-exact (tolerance 0):
  * full frame at a low sample count: image bits + all eight RayStats counters against `ora_render`;
  * the bench's batch shape — 256 spp of every pixel in ONE wavefront batch (530.8 M paths at 1920x1080; config 5
    at 3840x2160: 64 spp, 530.8 M paths) — against `ora_render_pixels` on a seeded random subset of pixels (per-pixel
    independence, tracer.rs:543, :559-560, makes a subset exact). cornellbox, the bench workload itself, is
    compared over the whole frame at 256 spp.
"""
import os

import numpy as np
import pytest

import ora
import ora_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH = 256  # bench.py --spp-per-step default
BATCH_4K = 64  # config 5 at 3840x2160: 530.8 M paths, 87 GB of path state per batch


def _path(name):
    p = os.path.join(ROOT, "scenes", name + ".usda")
    return p if os.path.exists(p) else os.path.join(ROOT, "scenes", name + ".usd")


def _counters_equal(st, ost, what):
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), (what, f, getattr(st, f), getattr(ost, f))


def _subset(n_pix, n, seed):
    return np.sort(np.random.default_rng(seed).choice(n_pix, n, replace=False)).astype(np.uint32)


def _render(crt, name, w, h, depth, spp, rank=0, world=1):
    import torch
    r, desc = crt.load_usda(_path(name), w, h, depth, rank=rank, world=world)
    r.render_samples(0, spp)  # ONE batch
    torch.cuda.synchronize()
    return r, desc


def test_config2_cornellbox_1080p_depth32_one_bench_batch_whole_frame(crt):
    """configs[1], the bench workload at the bench's step shape (256 spp of every pixel in ONE batch: 530.8 M paths,
    87 GB of path state, 4096 workgroup segments, the per-stage pipeline with the four-workgroups-per-CU traversal kernels):
    every pixel and every counter against the oracle's full frame."""
    w, h, depth = 1920, 1080, 32
    r, desc = _render(crt, "cornellbox", w, h, depth, BATCH)
    assert r.settings.max_depth == depth
    img, st = r.image(), r.stats()
    assert st.camera_rays == w * h * BATCH
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda, max_depth=depth).render(BATCH, forward=1)
    _counters_equal(st, ost, "cornellbox 1080p, one bench batch")
    bad = np.argwhere(img.view(np.uint32) != oimg.view(np.uint32))
    assert bad.shape[0] == 0, f"{bad.shape[0]} differing components, first {bad[:3]}"
    # and a second batch continues the same film exactly as the oracle's 512-sample pixel loop does (subset)
    import torch
    r.render_samples(BATCH, BATCH)
    torch.cuda.synchronize()
    idx = _subset(w * h, 4096, 2)
    opx, _ = ora_world.OracleRenderer(desc, crt.usda, max_depth=depth).render_pixels(idx, 2 * BATCH, forward=1)
    assert np.array_equal(r.image().reshape(-1, 3)[idx].view(np.uint32), opx.view(np.uint32))


def test_config2_cornellbox_1080p_the_shipped_bench_batch(crt):
    """bench.py's default step since round 3: 512 spp of every pixel in ONE batch — 1 061.7 M paths, 174 GB of path state,
    8 192 workgroup segments in lanes on their own streams, 16-byte camera paths, the fused tail from bounce 12 — on 4 096 random pixels against the
    oracle (per-pixel independence makes any subset exact), and the ray total against the 256-spp whole-frame run's law:
    every camera ray is traced."""
    import torch
    w, h, depth, spp = 1920, 1080, 32, 512
    r, desc = _render(crt, "cornellbox", w, h, depth, spp)
    st = r.stats()
    assert st.camera_rays == w * h * spp and st.closest_hit > st.camera_rays and st.shadow_rays == 0
    p = r.pipeline()
    # the batch runs as lanes (sub-batches of consecutive samples on their own streams): 8 192 segments in all
    assert p["fused"] is False and p["wide"] is True and r.lanes() >= 2 and p["grid"] * r.lanes() == 8192
    idx = _subset(w * h, 4096, 5)
    opx, _ = ora_world.OracleRenderer(desc, crt.usda, max_depth=depth).render_pixels(idx, spp, forward=1)
    assert np.array_equal(r.image().reshape(-1, 3)[idx].view(np.uint32), opx.view(np.uint32))
    del r
    torch.cuda.empty_cache()


def test_config3_openpbr_showcase_1080p_authored_depth(crt):
    """configs[2]: authored depth 32 (openpbr_showcase.usda:248-256), every lobe + interior media."""
    w, h = 1920, 1080
    r, desc = _render(crt, "openpbr_showcase", w, h, None, 4)
    assert r.settings.max_depth == 32 == desc.settings["max_depth"]
    o = ora_world.OracleRenderer(desc, crt.usda)
    oimg, ost = o.render(4, forward=1)
    _counters_equal(r.stats(), ost, "openpbr_showcase 1080p x4")
    assert np.array_equal(r.image().view(np.uint32), oimg.view(np.uint32))
    assert ost.shadow_rays > 0 and ost.ended_depth >= 0
    del r
    r, _ = _render(crt, "openpbr_showcase", w, h, None, BATCH)   # the bench's batch shape
    idx = _subset(w * h, 8192, 3)
    opx, _ = o.render_pixels(idx, BATCH, forward=1)
    assert np.array_equal(r.image().reshape(-1, 3)[idx].view(np.uint32), opx.view(np.uint32))


def test_config4_veach_mis_1080p_depth8_eight_shards(crt):
    """configs[3]: world = 8 renderers (one after another on this box's GPU) own disjoint 16x16 tiles; their union is
    the world = 1 frame and the oracle's frame, bit for bit, and their counters add up to the oracle's."""
    import torch
    w, h, depth, spp = 1920, 1080, 8, 4
    full, desc = _render(crt, "veach_mis", w, h, None, spp)
    assert full.settings.max_depth == depth and desc.settings["strategy"] == "balance"
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda).render(spp, forward=1)
    _counters_equal(full.stats(), ost, "veach_mis world=1")
    assert np.array_equal(full.image().view(np.uint32), oimg.view(np.uint32))
    img = np.zeros((w * h, 3), np.float32)
    seen = np.zeros(w * h, bool)
    total = {f: 0 for f, _t in ora.RayStats._fields_}
    sizes = []
    for rank in range(8):
        r, _ = _render(crt, "veach_mis", w, h, None, spp, rank=rank, world=8)
        idx = r.pixel_indices()
        assert np.array_equal(idx, crt.shard.shard_pixels(w, h, rank, 8))
        assert not seen[idx].any()
        seen[idx] = True
        img[idx] = r.film()
        st = r.stats()
        for f in total:
            total[f] += getattr(st, f)
        sizes.append(idx.size)
        del r
    assert seen.all() and max(sizes) - min(sizes) <= 256 * 2  # round-robin tiles: balanced to a tile or two
    assert np.array_equal(img.reshape(h, w, 3).view(np.uint32), oimg.view(np.uint32))
    for f in total:
        assert total[f] == getattr(ost, f), f
    # one shard at the bench's batch shape (weak scaling: 256 spp x 8 ranks per step), a subset of ITS pixels
    r, _ = _render(crt, "veach_mis", w, h, None, BATCH * 8, rank=5, world=8)  # 2048 spp of 1/8 of the pixels
    own = r.pixel_indices()
    pick = np.sort(np.random.default_rng(4).choice(own.size, 512, replace=False))
    opx, _ = ora_world.OracleRenderer(desc, crt.usda).render_pixels(own[pick], BATCH * 8, forward=1)
    assert np.array_equal(r.film()[pick].view(np.uint32), opx.view(np.uint32))


def test_config5_medcity_2160p(crt):
    """configs[4]'s own file at its own 3840x2160: the whole frame at 1 spp (image + counters), then ONE 64-spp batch
    — 530.8 M paths, 87 GB of path state — on a random subset of pixels."""
    w, h = 3840, 2160
    r, desc = _render(crt, "PointInstancedMedCity", w, h, None, 1)
    depth = r.settings.max_depth
    assert depth == desc.settings["max_depth"]
    o = ora_world.OracleRenderer(desc, crt.usda)
    oimg, ost = o.render(1, forward=1)
    st = r.stats()
    _counters_equal(st, ost, "MedCity 2160p x1")
    assert st.camera_rays == w * h and st.closest_hit > st.camera_rays
    assert np.array_equal(r.image().view(np.uint32), oimg.view(np.uint32))
    del r
    r, _ = _render(crt, "PointInstancedMedCity", w, h, None, BATCH_4K)
    st = r.stats()
    assert st.camera_rays == w * h * BATCH_4K
    idx = _subset(w * h, 3072, 5)
    opx, _ = o.render_pixels(idx, BATCH_4K, forward=1)
    assert np.array_equal(r.image().reshape(-1, 3)[idx].view(np.uint32), opx.view(np.uint32))


def test_synthetic_big_7m_triangles_at_the_bench_shape(crt):
    """`bench.py --scene synthetic:big` (LABELLED synthetic: 27 spheres at 512 x 256 quads = 7 077 888 triangles, a 1.03 GB
    device image in a closed room under a RectLight — the out-of-cache workload whose traversal kernels are HBM-bound inside
    the integrator): its own 1920x1080 x 128-spp batch (265.4 M paths) through the renderer's choice — one launch per stage
    on the four-wave kernels with the large-tree arena split — on 1 024 random pixels against the oracle, whose builder
    commits the same 7 M triangles on the CPU (per-pixel independence makes any subset exact)."""
    import torch
    w, h, spp = 1920, 1080, 128
    r, desc = crt.load_usda("synthetic:big", w, h, None)
    assert r.scene.primitive_breakdown()["triangles"] == 27 * 512 * 256 * 2 + 12 + 2
    assert r.scene.engine_select(-4)["wide"] == 1 and r.scene.engine_select(-4)["lds_stack"] == 5
    r.render_samples(0, spp)
    torch.cuda.synchronize()
    st, p = r.stats(), r.pipeline()
    assert p["fused"] is False and p["wide"] is True
    assert st.camera_rays == w * h * spp and st.shadow_rays > st.camera_rays and st.ended_escaped < st.camera_rays // 1000  # a closed room
    idx = _subset(w * h, 1024, 17)
    opx, _ = ora_world.OracleRenderer(desc, crt.usda).render_pixels(idx, spp, forward=1)
    assert np.array_equal(r.image().reshape(-1, 3)[idx].view(np.uint32), opx.view(np.uint32))
    del r
    torch.cuda.empty_cache()
