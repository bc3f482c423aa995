"""A host without Python, torch or HIP calls of its own renders through the C ABI: examples/host_c/crt_host.c (C99 over
include/crt.h: SceneBuilder -> commit -> Renderer -> film) reads the scene description build_world feeds the Python mirror
and writes the frame and the RayStats. Bit for bit the image of the Python host (same library, other binding) and of the
oracle; the reference's call sequence (scene.rs:152-341, tracer.rs:137-148, :405-470) is all a binding needs."""
import importlib
import os

import numpy as np
import pytest

import host_c_scene as hc
import ora
import ora_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# scene, width, height, depth, spp, batch, environment of the host process
CASES = [("cornellbox", 64, 64, 8, 8, 8, {}), ("veach_mis", 96, 54, 8, 6, 4, {}), ("openpbr_showcase", 96, 54, 12, 4, 4, {}),
         ("motionblur", 96, 54, 4, 8, 8, {}), ("nested_instancing", 96, 54, 8, 4, 2, {}), ("sun_sky", 96, 54, 8, 4, 4, {}),
         # one launch per stage on the four-wave kernels (a small batch runs the fused kernel by itself)
         ("veach_mis", 96, 54, 8, 6, 6, {"CRT_FUSED": "0", "CRT_STAGE_MIN_PATHS": "1"})]


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return hc.build_host(tmp_path_factory.mktemp("host_c"))


@pytest.mark.parametrize("name,w,h,depth,spp,batch,env", CASES)
def test_the_c_host_renders_what_the_python_host_and_the_oracle_render(crt, exe, tmp_path, name, w, h, depth, spp, batch, env):
    import torch
    desc = crt.usda.load(os.path.join(ROOT, "scenes", name + ".usda"), w, h)
    desc.settings["max_depth"] = depth
    scene, mats, _protos = crt.usda.build_world(desc, crt, crt.default_material)
    res, film = hc.run_host(exe, hc.scene_blob(crt, desc, mats, spp, batch), tmp_path, env)
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-2000:])
    if env:
        assert "per stage" in res.stdout, res.stdout
    raw = np.fromfile(film, dtype=np.uint8)
    assert raw.size == w * h * 12 + 64
    img = raw[:w * h * 12].view(np.float32).reshape(h, w, 3)
    st = raw[w * h * 12:].view(np.uint64)

    s = desc.settings
    settings = crt.RenderSettings(s["width"], s["height"], s["max_depth"], s["frame"], s["strategy"], s["filter"],
                                  s["filter_radius"], 0.0)
    r = crt.Renderer(scene, mats, desc.lights, crt.make_camera(**desc.camera), settings)
    for b in range(0, spp, batch):
        r.render_samples(b, min(batch, spp - b))
    torch.cuda.synchronize()
    pimg, pst = r.image(), r.stats()
    oimg, ost = ora_world.OracleRenderer(desc, crt.usda).render(spp, forward=1)
    for k, (f, _t) in enumerate(ora.RayStats._fields_):
        assert int(st[k]) == getattr(pst, f) == getattr(ost, f), (name, f, int(st[k]), getattr(pst, f), getattr(ost, f))
    assert np.array_equal(img.view(np.uint32), pimg.view(np.uint32)), name
    assert np.array_equal(img.view(np.uint32), oimg.view(np.uint32)), name


@pytest.mark.parametrize("name,w,h,depth,spp,batch", [("veach_mis", 100, 54, 8, 6, 4), ("instancing", 96, 54, 8, 4, 4)])
def test_the_rccl_host_runs_its_collectives_and_assembles_the_same_frame(crt, tmp_path, name, w, h, depth, spp, batch):
    """examples/host_rccl/crt_rccl_host.cpp at WORLD_SIZE = 1 (all a one-GPU box can run): ncclCommInitRank, the padded
    shard through ncclAllGather, crt_gather_plan_assemble, the counters through ncclAllReduce — RCCL executing on the
    device, driven by a native host through the C ABI; frame and counters equal the Python host's. (The multi-rank
    index plan itself is covered by test_gather_plan_on_device at world 2 / 3 / 8.)"""
    import torch
    exe_rccl = hc.build_rccl_host(tmp_path)
    desc = crt.usda.load(os.path.join(ROOT, "scenes", name + ".usda"), w, h)
    desc.settings["max_depth"] = depth
    scene, mats, _protos = crt.usda.build_world(desc, crt, crt.default_material)
    res, film = hc.run_host(exe_rccl, hc.scene_blob(crt, desc, mats, spp, batch), tmp_path,
                            {"RANK": "0", "WORLD_SIZE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "NCCL_SOCKET_IFNAME": "lo"},
                            timeout=150)
    if res.returncode == 3 and "ncclCommInitRank" in res.stderr:
        # the communicator is the environment's (network interface, IPC mode), not this repository's: without one there is
        # nothing to run the collectives on — the layout is still covered by the loopback test below
        pytest.skip("RCCL could not initialise on this box: " + res.stderr.strip()[-300:])
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-2000:])
    assert "world 1, RCCL" in res.stdout, res.stdout
    raw = np.fromfile(film, dtype=np.uint8)
    img = raw[:w * h * 12].view(np.float32).reshape(h, w, 3)
    st = raw[w * h * 12:].view(np.uint64)
    s = desc.settings
    settings = crt.RenderSettings(s["width"], s["height"], s["max_depth"], s["frame"], s["strategy"], s["filter"],
                                  s["filter_radius"], 0.0)
    r = crt.Renderer(scene, mats, desc.lights, crt.make_camera(**desc.camera), settings)
    for b in range(0, spp, batch):
        r.render_samples(b, min(batch, spp - b))
    torch.cuda.synchronize()
    pimg, pst = r.image(), r.stats()
    for k, (f, _t) in enumerate(ora.RayStats._fields_):
        assert int(st[k]) == getattr(pst, f), (name, f)
    assert np.array_equal(img.view(np.uint32), pimg.view(np.uint32)), name


@pytest.mark.parametrize("world", [2, 3, 8])
def test_the_rccl_hosts_multi_rank_layout_in_loopback(crt, exe, tmp_path, world):
    """RCCL does not take two ranks on one GPU (profiles/r04_rccl_two_ranks.txt), so the native host's multi-rank data
    layout — every rank's renderer, its shard at recv + rank x padded x 3, crt_gather_plan_assemble, the counters' sum —
    runs with ONE process playing every rank (CRT_RCCL_LOOPBACK=1: no communicator): frame and counters must equal the
    single-process C host's on an odd frame size."""
    exe_rccl = hc.build_rccl_host(tmp_path)
    desc = crt.usda.load(os.path.join(ROOT, "scenes", "veach_mis.usda"), 117, 61)
    desc.settings["max_depth"] = 8
    _scene, mats, _protos = crt.usda.build_world(desc, crt, crt.default_material)
    blob = hc.scene_blob(crt, desc, mats, 6, 3)
    one_dir, many_dir = tmp_path / "one", tmp_path / "many"
    one_dir.mkdir(); many_dir.mkdir()
    res1, film1 = hc.run_host(exe, blob, one_dir)
    resn, filmn = hc.run_host(exe_rccl, blob, many_dir, {"RANK": "0", "WORLD_SIZE": str(world), "CRT_RCCL_LOOPBACK": "1"}, timeout=150)
    assert res1.returncode == 0 and resn.returncode == 0, (res1.stderr[-500:], resn.stderr[-1500:])
    assert "loopback" in resn.stdout
    assert np.array_equal(np.fromfile(film1, dtype=np.uint8), np.fromfile(filmn, dtype=np.uint8)), world
