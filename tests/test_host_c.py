"""The plain-C host (examples/host_c/crt_host.c) WITHOUT a GPU: it builds against include/crt.h and libcrt_amd.so alone
(C99, -Werror), reads a scene description, commits it through the C ABI on the host — and then fails loudly, because
there is no device to render on and no CPU fallback (exit 3, crt_last_error's text). The GPU run: tests/test_gpu_host_c.py."""
import importlib
import os

import pytest

import host_c_scene as hc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return hc.build_host(tmp_path_factory.mktemp("host_c"))


def _blob(crt, name, w, h, depth, spp, batch):
    desc = crt.usda.load(os.path.join(ROOT, "scenes", name + ".usda"), w, h)
    desc.settings["max_depth"] = depth
    _scene, mats, _protos = crt.usda.build_world(desc, crt, crt.default_material)
    return hc.scene_blob(crt, desc, mats, spp, batch)


def test_the_c_host_builds_and_refuses_to_render_without_a_device(exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: tests/test_gpu_host_c.py renders")
    crt = importlib.import_module("crust-render_amd")
    res, film = hc.run_host(exe, _blob(crt, "instancing", 32, 18, 4, 2, 2), tmp_path)
    assert res.returncode == 3, (res.returncode, res.stdout, res.stderr)
    assert "crt_renderer_new failed" in res.stderr and not os.path.exists(film)


def test_the_c_host_rejects_a_truncated_scene_file(exe, tmp_path):
    crt = importlib.import_module("crust-render_amd")
    blob = _blob(crt, "cornellbox", 16, 16, 4, 1, 1)
    for cut in (3, 40, len(blob) // 2, len(blob) - 4):
        res, film = hc.run_host(exe, blob[:cut], tmp_path)
        assert res.returncode == 2 and not os.path.exists(film), (cut, res.returncode, res.stderr)


def test_the_rccl_host_builds_and_needs_a_device(tmp_path):
    """examples/host_rccl/crt_rccl_host.cpp (one process per GPU, ncclAllGather + crt_gather_plan_assemble driven by the
    host) compiles against include/crt.h + <rccl/rccl.h>; without a device it stops before any collective."""
    import torch
    exe_rccl = hc.build_rccl_host(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a device is present: tests/test_gpu_host_c.py runs it")
    crt = importlib.import_module("crust-render_amd")
    res, film = hc.run_host(exe_rccl, _blob(crt, "cornellbox", 16, 16, 4, 1, 1), tmp_path)
    assert res.returncode == 3 and "no HIP device" in res.stderr and not os.path.exists(film), (res.returncode, res.stderr)
    res, _ = hc.run_host(exe_rccl, _blob(crt, "cornellbox", 16, 16, 4, 1, 1), tmp_path, {"RANK": "2", "WORLD_SIZE": "2"})
    assert res.returncode == 2 and "bad RANK" in res.stderr
