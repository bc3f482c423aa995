"""Parity of the wavefront path tracer with the oracle's integrator on the BASELINE scenes at reduced size.

The oracle's forward mode states the same estimator in the same accumulation order as the kernels, so the
images must be IDENTICAL bit for bit, and so must every RayStats counter. The oracle's reference-order
(backward gather, tracer.rs:1537-1557) image differs only by float association: bounded relative error."""
import os

import numpy as np
import pytest

import ora
import ora_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _both(crt, name, w, h, spp, depth, batch=None):
    import torch
    path = os.path.join(ROOT, "scenes", name + ".usda")
    r, desc = crt.load_usda(path, w, h, depth)
    o = ora_world.OracleRenderer(desc, crt.usda, max_depth=depth)
    batch = batch or spp
    for b in range(0, spp, batch):
        r.render_samples(b, min(batch, spp - b))
    torch.cuda.synchronize()
    img = r.image()
    st = r.stats()
    oimg, ost = o.render(spp, forward=1)
    return img, st, oimg, ost, o


SCENES = [("cornellbox", 64, 64, 8, 4), ("cornellbox", 96, 54, 4, 32), ("veach_mis", 96, 54, 8, 8),
          ("openpbr_showcase", 96, 54, 8, 12), ("cornellbox_guided", 48, 48, 8, 8), ("sun_sky", 96, 54, 8, 8),
          # the reference's remaining samples: transform motion blur at RENDER level — the K_TIME draw
          # (tracer.rs:579-583), the time plane, the cold-time kernel instances —, a RectLight, light geometry with per-light
          # camera visibility / ray masks, a PointInstancer, nested instancers, and (outside SURVEY §8) the dome sample
          ("motionblur", 96, 54, 8, 4), ("rectlight", 96, 54, 8, 4), ("light_visibility", 96, 54, 8, 4),
          ("instancing", 96, 54, 8, 8), ("nested_instancing", 96, 54, 8, 8), ("domelight", 96, 54, 8, 8)]


@pytest.mark.parametrize("name,w,h,spp,depth", SCENES)
def test_image_and_counters_identical_to_oracle(crt, name, w, h, spp, depth):
    img, st, oimg, ost, _ = _both(crt, name, w, h, spp, depth)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), (name, f, getattr(st, f), getattr(ost, f))
    assert st.camera_rays == w * h * spp
    assert np.isfinite(oimg).all()
    bad = np.argwhere(img.view(np.uint32) != oimg.view(np.uint32))
    assert bad.shape[0] == 0, f"{name}: {bad.shape[0]} differing components, first {bad[:3]}"


@pytest.mark.parametrize("strategy", ["balance", "light", "bsdf"])
def test_lights_at_infinity_under_every_strategy(crt, strategy):
    """sun_sky (DistantLight + uniform DomeLight + a SphereLight): escaped_emission's MIS weights and the NEE of
    lights without geometry (tracer.rs:966-1009, light.rs:285-298, :358-383) under the other three strategies."""
    import torch
    desc = crt.usda.load(os.path.join(ROOT, "scenes", "sun_sky.usda"), 80, 45)
    desc.settings["strategy"] = strategy
    desc.settings["max_depth"] = 6
    scene, mats, _protos = crt.usda.build_world(desc, crt, crt.default_material)
    s = desc.settings
    settings = crt.RenderSettings(s["width"], s["height"], s["max_depth"], s["frame"], s["strategy"], s["filter"],
                                  s["filter_radius"], 0.0)
    r = crt.Renderer(scene, mats, desc.lights, crt.make_camera(**desc.camera), settings)
    r.render_samples(0, 6)
    torch.cuda.synchronize()
    img, st = r.image(), r.stats()
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda).render(6, forward=1)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), (strategy, f)
    assert np.array_equal(img.view(np.uint32), oimg.view(np.uint32))


def test_medcity_crate_identical_to_oracle(crt):
    """BASELINE config 5's own file (binary USDC, PointInstancer with 40 000 instances of 8 prototypes), read by the
    crate reader: image and counters identical to the oracle."""
    import torch
    r, desc = crt.load_usda(os.path.join(ROOT, "scenes", "PointInstancedMedCity.usd"), 96, 54, 6)
    r.render_samples(0, 4)
    torch.cuda.synchronize()
    img, st = r.image(), r.stats()
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda, max_depth=6).render(4, forward=1)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), f
    assert st.closest_hit > st.camera_rays and np.array_equal(img.view(np.uint32), oimg.view(np.uint32))


def test_synthetic_city_identical_to_oracle(crt):
    """The labelled stand-in for BASELINE config 5 (instancing-heavy): 576 instances of four prototypes, two sphere
    lights, five material variants — image and counters identical to the oracle."""
    import torch
    r, desc = crt.load_usda("synthetic:city:24", 96, 54, 6)
    r.render_samples(0, 4)
    torch.cuda.synchronize()
    img, st = r.image(), r.stats()
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda, max_depth=6).render(4, forward=1)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), f
    assert st.shadow_rays > 0 and np.array_equal(img.view(np.uint32), oimg.view(np.uint32))


def test_adaptive_stopping_identical_to_oracle(crt):
    """render_pixel's adaptive early stop (tracer.rs:609-617): same image, and — with batches of 4 samples after
    the first min_spp, where the rule can fire — the same ray counters and per-pixel sample counts."""
    import torch
    path = os.path.join(ROOT, "scenes", "veach_mis.usda")
    w, h, spp, min_spp, var = 64, 36, 48, 8, 0.08
    r, desc = crt.load_usda(path, w, h, 8, variance=var, min_spp=min_spp)
    done = r.render_adaptive(spp)
    torch.cuda.synchronize()
    img, st = r.image(), r.stats()
    o = ora_world.OracleRenderer(desc, crt.usda, max_depth=8, variance=var, min_spp=min_spp)
    oimg, ost = o.render(spp, forward=1)
    counts = r.sample_counts()
    assert 0 < (counts < spp).sum() < counts.size, "the threshold should stop some pixels early and not others"
    assert (counts % 4 == 0).all() and counts.min() >= min_spp and done <= spp
    assert st.camera_rays == int(counts.sum()) == ost.camera_rays
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), f
    assert np.array_equal(img.view(np.uint32), oimg.view(np.uint32))
    # any other batching gives the same image (the rule is evaluated per sample inside the fold)
    r2, _ = crt.load_usda(path, w, h, 8, variance=var, min_spp=min_spp)
    r2.render_adaptive(spp, first=20, batch=28)
    torch.cuda.synchronize()
    assert np.array_equal(r2.image().view(np.uint32), oimg.view(np.uint32))
    assert np.array_equal(r2.sample_counts(), counts)


def test_batches_accumulate_in_sample_order(crt):
    """Rendering 8 spp as 1 batch or as 4 batches of 2 must give the same bits (sum += color in sample order)."""
    a, *_ = _both(crt, "veach_mis", 64, 36, 8, 8, batch=8)
    b, *_ = _both(crt, "veach_mis", 64, 36, 8, 8, batch=2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_forward_equals_reference_gather_within_float_association(crt):
    """The reference's estimator (backward gather) vs the forward accumulation: same samples, same decisions;
    stated tolerance: per-pixel L-inf <= 1e-5 * max(1, pixel)."""
    _, _, ofwd, ost, o = _both(crt, "veach_mis", 64, 36, 8, 8)
    oref, ost2 = o.render(8, forward=0)
    for f, _t in ora.RayStats._fields_:
        assert getattr(ost, f) == getattr(ost2, f)
    err = np.abs(ofwd - oref) / np.maximum(1.0, np.abs(oref))
    assert err.max() <= 1e-5, err.max()


def test_tile_shards_reassemble_the_single_gpu_image(crt):
    """Pixel-tile sharding: world=2 renderers own disjoint tiles; their union is bit-identical to world=1."""
    import torch
    path = os.path.join(ROOT, "scenes", "cornellbox.usda")
    full, _ = crt.load_usda(path, 80, 48, 4)
    full.render_samples(0, 4)
    parts = []
    for rank in range(2):
        r, _ = crt.load_usda(path, 80, 48, 4, rank=rank, world=2)
        r.render_samples(0, 4)
        parts.append(r)
    torch.cuda.synchronize()
    img = np.zeros((48 * 80, 3), np.float32)
    seen = np.zeros(48 * 80, bool)
    for r in parts:
        idx = r.pixel_indices()
        assert not seen[idx].any()
        seen[idx] = True
        img[idx] = r.film()
    assert seen.all()
    assert np.array_equal(img.reshape(48, 80, 3).view(np.uint32), full.image().view(np.uint32))


@pytest.mark.parametrize("scene,w,h,depth", [("openpbr_showcase", 96, 54, 12), ("cornellbox", 96, 54, 8),
                                             ("nested_instancing", 64, 36, 6), ("stress", 96, 54, 6),
                                             ("motionblur", 96, 54, 4)])
def test_the_three_pipelines_agree(crt, tmp_path, scene, w, h, depth):
    """The renderer picks its pipeline per scene and batch (pathtrace.hip, Renderer::fused): the fused path-loop kernel on
    three workgroups per CU, or one launch per stage with the four-workgroups-per-CU traversal kernels for large batches of
    small flat triangle scenes. All combinations (fused; per-stage with either engine split; switching between batches of
    one render) are the same device functions: identical image and counters on a lit scene with interior media, a
    triangle scene and nested instances."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np; sys.path.insert(0, %r); import torch\n"
        "from __graft_entry__ import load_package; crt = load_package()\n"
        "r, _ = crt.load_usda(crt.scene_path(%r), %d, %d, %d)\n"
        "r.render_samples(0, 5); p1 = r.pipeline(); r.render_samples(5, 3); p2 = r.pipeline(); torch.cuda.synchronize(); st = r.stats()\n"
        "np.save(sys.argv[1], r.image()); print(st.closest_hit, st.shadow_rays, st.vertices, st.rr_killed, int(p1['fused']), int(p1['wide']), int(p2['fused']))\n"
        % (ROOT, scene, w, h, depth))
    outs = []
    # every scene's renderer has four-wave traversal kernels to run: flat trees the plain ones, direct-leaf images
    # (openpbr_showcase: packet-free) their direct-engine instances — CRT_WIDE=1 on such an image falls back to that choice
    wide = 1
    for tag, env, want in (("fused", dict(CRT_FUSED="1"), (1, 0, 1)), ("stage3", dict(CRT_FUSED="0", CRT_WIDE="0"), (0, 0, 0)),
                           ("stage4", dict(CRT_FUSED="0", CRT_WIDE="1"), (0, wide, 0)),
                           # the batch decides: 5 spp of every pixel reach the threshold, 3 do not
                           ("by_batch", dict(CRT_WIDE="1", CRT_STAGE_MIN_PATHS=str(w * h * 4)), (0, wide, 1))):
        path = str(tmp_path / ("img_%s.npy" % tag))
        res = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), capture_output=True, text=True,
                             timeout=300)
        assert res.returncode == 0, (tag, res.stderr[-2000:])
        line = res.stdout.strip().splitlines()[-1].split()
        assert tuple(int(x) for x in line[4:]) == want, (tag, line)
        outs.append((np.load(path), line[:4]))
    for img, line in outs[1:]:
        assert line == outs[0][1]
        assert np.array_equal(img.view(np.uint32), outs[0][0].view(np.uint32))
    assert int(outs[0][1][0]) > w * h * 8  # bounces happened


def test_render_report_counts_the_scene_and_the_rays(crt, tmp_path):
    """render_with_report (main.rs:491-648 in one call): the report's scene block shows cornellbox as the reference's
    importer leaves it (822 baked triangles + 2 placements of one 760-triangle mesh), the ray block
    equals the renderer's counters, every phase is listed, and the EXR on disk is the image returned."""
    out = str(tmp_path / "cb.exr")
    img, st = crt.render_with_report(os.path.join(ROOT, "scenes", "cornellbox.usda"), spp=8, out_exr=out, width=64,
                                     height=36, variance=0.0)
    assert st.scene.top_level.triangles == 822 and st.scene.top_level.instances == 2
    assert st.scene.unique.triangles == 822 + 760 and st.scene.unique.instances == 2
    assert st.image.width == 64 and st.image.samples_per_pixel == 8
    assert st.rays.camera_rays == 64 * 36 * 8 and st.rays.total_rays() > st.rays.camera_rays
    text = st.report()
    for needle in ("Render Statistics", "primitives in memory", "kernel memory", "Ray Statistics", "throughput",
                   "Parse USD stage", "Commit acceleration structure", "Render", "Write image", "Profile by time"):
        assert needle in text, needle
    back = crt.exr.read_exr(out)
    assert np.array_equal(np.ascontiguousarray(back[::-1], dtype=np.float32).view(np.uint32), img.view(np.uint32))


def test_shade_class_partition_keeps_waves_pure_and_bits_unchanged(crt, tmp_path):
    """Shade's material-class partition (north_star: per-material sorting on wave primitives): with it the vertex step's
    waves hold one class each — lane utilisation of every class that occurs is far above the natural mix's — and the
    image and counters are the ones the unpartitioned path (CRT_PARTITION=0) produces (which the other tests of this file
    compare with the oracle). 960x540 x 16 spp: segments long enough (11 000 paths per workgroup) for full rings."""
    import subprocess
    import sys
    import json
    code = (
        "import os, sys, json, numpy as np; sys.path.insert(0, %r); import torch\n"
        "from __graft_entry__ import load_package; crt = load_package()\n"
        "r, _ = crt.load_usda(os.path.join(%r, 'scenes', 'openpbr_showcase.usda'), 960, 540, 12)\n"
        "r.shade_class_stats(True); r.render_samples(0, 16); torch.cuda.synchronize(); st = r.stats()\n"
        "np.save(sys.argv[1], r.image())\n"
        "print(json.dumps(dict(stats=[st.closest_hit, st.shadow_rays, st.vertices, st.rr_killed], cls=r.shade_class_stats())))\n" % (ROOT, ROOT))
    out = {}
    for part in ("0", "1"):
        path = str(tmp_path / ("crt_part%s.npy" % part))
        res = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, CRT_PARTITION=part), capture_output=True,
                             text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        out[part] = (np.load(path), json.loads(res.stdout.strip().splitlines()[-1]))
    assert out["0"][1]["stats"] == out["1"][1]["stats"]
    assert np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32))
    mixed, pure = out["0"][1]["cls"], out["1"][1]["cls"]
    seen = [k for k, (w, u) in pure.items() if w > 0]
    assert len(seen) >= 3, pure  # the showcase holds emissive lights, base, layered and transmissive looks
    for k in seen:
        assert pure[k][1] > mixed[k][1], (k, pure[k], mixed[k])
    assert min(pure[k][1] for k in seen) > 0.5 and pure["base"][1] > 0.95


@pytest.mark.parametrize("scene,res,base_env,knob", [
    ("PointInstancedMedCity.usd", (96, 54, 6), {}, {"CRT_MAT_DEDUP": "0"}),                 # 40 008 records instead of the distinct ones
    ("cornellbox.usda", (160, 90, 8), {"CRT_FUSED": "0"}, {"CRT_CAM_COMPACT": "0"}),       # camera paths as full records
    ("cornellbox.usda", (160, 90, 8), {"CRT_FUSED": "0"}, {"CRT_NOCLASSIFY_FROM": "1"}),   # shade without its CLASSIFY pass
    ("cornellbox.usda", (160, 90, 8), {"CRT_FUSED": "0"}, {"CRT_HOT_PACKETS": "0"}),       # packets in builder order
    ("stress", (96, 54, 6), {}, {"CRT_POOL_STACK_RT": "6"}),                               # a large tree on the flat LDS split
    # a batch as ONE lane against two / three lanes on their own streams (forced: these batches are far below the size
    # from which the renderer splits on its own) — per-stage unlit, fused lit, fused instanced, 8 samples into 3 lanes: 3 + 3 + 2
    ("cornellbox.usda", (160, 90, 8), {"CRT_FUSED": "0", "CRT_LANES": "1"}, {"CRT_LANES": "2", "CRT_LANE_MIN_PATHS": "1"}),
    ("stress", (96, 54, 6), {"CRT_LANES": "1"}, {"CRT_LANES": "3", "CRT_LANE_MIN_PATHS": "1"}),
    ("PointInstancedMedCity.usd", (96, 54, 6), {"CRT_LANES": "1"}, {"CRT_LANES": "4", "CRT_LANE_MIN_PATHS": "1"}),
    ("veach_mis.usda", (160, 90, 8), {"CRT_FUSED": "0", "CRT_LANES": "1"}, {"CRT_LANES": "3", "CRT_LANE_MIN_PATHS": "1"}),
    # render-level motion blur (the shutter-time plane travels with the path): fused and per-stage, one lane against three
    ("motionblur.usda", (160, 90, 4), {"CRT_LANES": "1"}, {"CRT_LANES": "3", "CRT_LANE_MIN_PATHS": "1"}),
    ("motionblur.usda", (160, 90, 4), {"CRT_FUSED": "0", "CRT_LANES": "1"}, {"CRT_FUSED": "0", "CRT_LANES": "3", "CRT_LANE_MIN_PATHS": "1"}),
    ("motionblur.usda", (160, 90, 4), {"CRT_FUSED": "1"}, {"CRT_FUSED": "0", "CRT_WIDE": "0"}),
])
def test_round3_knobs_change_no_bit(crt, tmp_path, scene, res, base_env, knob):
    """Round 3's layout and scheduling choices — deduplicated material table, 16-byte camera paths, hot-first packet order,
    the deep LDS split for large trees, shade's CLASSIFY pass — change WHEN and WHERE things are read, never a result: the
    same render with the choice turned off gives the same image bits and the same eight counters."""
    import subprocess
    import sys
    import json
    w, h, depth = res
    code = (
        "import os, sys, json, numpy as np; sys.path.insert(0, %r); import torch\n"
        "from __graft_entry__ import load_package; crt = load_package()\n"
        "r, _ = crt.load_usda(crt.scene_path(%r) if not %r.endswith(('.usd', '.usda')) else os.path.join(%r, 'scenes', %r), %d, %d, %d)\n"
        "r.render_samples(0, 8); torch.cuda.synchronize(); st = r.stats()\n"
        "np.save(sys.argv[1], r.image())\n"
        "print(json.dumps([getattr(st, f) for f, _t in st._fields_]))\n" % (ROOT, scene, scene, ROOT, scene, w, h, depth))
    out = []
    for k, extra in enumerate(({}, knob)):
        path = str(tmp_path / ("crt_knob%d.npy" % k))
        env = dict(os.environ)
        env.update(base_env)
        env.update(extra)
        res_ = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300)
        assert res_.returncode == 0, res_.stderr[-2000:]
        out.append((np.load(path), json.loads(res_.stdout.strip().splitlines()[-1])))
    assert out[0][1] == out[1][1], (knob, out[0][1], out[1][1])
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)), knob


def test_lanes_and_single_lane_batches_alternate_within_one_renderer(crt, monkeypatch):
    """A renderer that runs a batch as lanes, then a single-lane batch (the stats pass), then lanes again: the lanes a
    batch does not use give their buffers back and a whole-batch buffer is not kept next to lanes (Renderer::render,
    ensure_buffers) — and the image is the one a single-lane renderer produces for the same samples."""
    import torch
    monkeypatch.setenv("CRT_LANES", "3")
    monkeypatch.setenv("CRT_LANE_MIN_PATHS", "1")
    r, desc = crt.load_usda(crt.scene_path("veach_mis"), 96, 54, 8)
    r.render_samples(0, 6)
    assert r.lanes() == 3
    ext, sh = r.render_samples_stats(6, 6)   # single lane by construction
    assert r.lanes() == 1 and ext.rays > 0 and sh.rays > 0
    r.render_samples(12, 5)
    assert r.lanes() == 3
    torch.cuda.synchronize()
    img, st = r.image(), r.stats()
    monkeypatch.setenv("CRT_LANES", "1")
    r1, _ = crt.load_usda(crt.scene_path("veach_mis"), 96, 54, 8)
    r1.render_samples(0, 6); r1.render_samples(6, 6); r1.render_samples(12, 5)
    torch.cuda.synchronize()
    assert np.array_equal(img.view(np.uint32), r1.image().view(np.uint32))
    st1 = r1.stats()
    assert (st.closest_hit, st.shadow_rays, st.vertices) == (st1.closest_hit, st1.shadow_rays, st1.vertices)


def test_gather_plan_of_the_c_abi_assembles_what_the_indexed_stores_do(crt):
    """crt_gather_plan_assemble (the frame assembled on the device from the ranks' padded tile buffers — what a host that
    drives RCCL itself calls after its collective) against the per-rank indexed stores of the CPU path, for an odd frame
    and 3 and 8 ranks; the padded count is the C ABI's on both."""
    import torch
    for (w, h, world) in ((100, 52, 3), (1920, 1080, 8), (33, 17, 2)):
        n_max = crt.shard.padded_count(w, h, world)
        dev, cpu = crt.shard.GatherPlan(w, h, world, "cuda"), crt.shard.GatherPlan(w, h, world, "cpu")
        assert dev.n_max == cpu.n_max == n_max and n_max % 256 == 0
        rng = np.random.default_rng(w + world)
        recv = torch.zeros(world, n_max, 3, dtype=torch.float32)
        for r in range(world):
            n = crt.shard.shard_pixels(w, h, r, world).size
            assert n <= n_max
            recv[r, :n] = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
        # stand-in for the collective's output: every rank's padded buffer, in rank order
        dev.recv.copy_(recv.reshape(-1).cuda())
        cpu.recv.copy_(recv.reshape(-1))
        a = dev.gather(torch.zeros(0, 3, device="cuda"))  # world > 1 but dist None: assembles what recv holds
        b = cpu.gather(torch.zeros(0, 3))
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy().view(np.uint32), b.numpy().view(np.uint32)), (w, h, world)


@pytest.mark.parametrize("env,want", [({"CRT_STAGE_MIN_PATHS": "1"}, (0, 1)), ({"CRT_STAGE_MIN_PATHS": "1", "CRT_WIDE": "0"}, (0, 0)),
                                      ({"CRT_FUSED": "1"}, (1, 0))])
def test_synthetic_big_identical_to_oracle(crt, env, want, monkeypatch):
    """The labelled synthetic out-of-cache workload (synthetic.big: 27 tessellated spheres in a closed room under a rect
    light; here 96 x 48 quads per sphere = 248 832 triangles, the bench runs 512 x 256 = 7.08 M): a large flat tree. The
    renderer's own choice for it — one launch per stage on the four-wave kernels with their large-tree split (five stack
    entries, twelve-node window, no mask plane; here forced onto a small batch) —, the three-wave per-stage kernels on the
    deep split, and the fused kernel: image and all eight counters identical to the oracle."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r, desc = crt.load_usda("synthetic:big:96", 96, 54, 6)
    assert r.scene.engine_select(-4) == dict(r.scene.engine_select(-4), wide=1, direct=0, lds_stack=5, window=12)  # the renderer's preference
    assert r.scene.engine_select(-1)["wide"] == 0 and r.scene.engine_select(-1)["lds_stack"] == 10              # the batched queries'
    r.render_samples(0, 4)
    torch.cuda.synchronize()
    p = r.pipeline()
    assert (int(p["fused"]), int(p["wide"])) == want, p
    img, st = r.image(), r.stats()
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda, max_depth=6).render(4, forward=1)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(ost, f), (env, f)
    assert st.shadow_rays > 0 and st.ended_depth > 0 and np.array_equal(img.view(np.uint32), oimg.view(np.uint32))


def test_lane_count_set_through_the_abi_changes_no_bit(crt, monkeypatch):
    """crt_renderer_set_lanes (what bench.py's one-lane roofline legs call): the same renderer traces a batch as four
    lanes, then the next as one, then as three — the film is the one a single-lane renderer accumulates."""
    import torch
    monkeypatch.setenv("CRT_LANE_MIN_PATHS", "1")
    r, _ = crt.load_usda(crt.scene_path("rectlight"), 128, 72, 4)
    assert r.set_lanes(4) == 4
    r.render_samples(0, 8); assert r.lanes() == 4
    assert r.set_lanes(1) == 1
    r.render_samples(8, 8); assert r.lanes() == 1
    assert r.set_lanes(9) == 4 and r.set_lanes(3) == 3
    r.render_samples(16, 7); assert r.lanes() == 3
    torch.cuda.synchronize()
    monkeypatch.setenv("CRT_LANES", "1")
    ref, _ = crt.load_usda(crt.scene_path("rectlight"), 128, 72, 4)
    ref.render_samples(0, 8); ref.render_samples(8, 8); ref.render_samples(16, 7)
    torch.cuda.synchronize()
    assert np.array_equal(r.image().view(np.uint32), ref.image().view(np.uint32))
    a, b = r.stats(), ref.stats()
    assert [getattr(a, f) for f, _t in a._fields_] == [getattr(b, f) for f, _t in b._fields_]
