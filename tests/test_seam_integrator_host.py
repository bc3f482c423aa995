"""The reference's per-pixel integrator on hooked seams WITHOUT a GPU (tests/seam_integrator.py; the GPU run is
tests/test_gpu_seam_integrator.py): the oracle's trace_path calls the kernel seam and the Material / Light traits through
OraSeamHooks; the shading backend is (a) the device's shading SOURCE (kernels/shade.hip.h) compiled as host C++ — the
functions crt_material_*_n / crt_light_*_n launch — and (b) the oracle's own batched drivers (plumbing check); traversal
stays the oracle's (a kernel cannot run here). Image bits and counters must equal the oracle on its own functions: the
record layouts of include/crt.h carry everything trace_path needs, and the device source composes to the same image."""
import ctypes as C
import os
import subprocess

import pytest

import ora_world
import seam_cases as sc
import seam_integrator as si

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_drivers(tmp_path_factory):
    out = tmp_path_factory.mktemp("seam_host") / "libseam_host.so"
    k = os.path.join(ROOT, "crust-render_amd", "csrc", "kernels")
    cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-attributes",
           "-I" + os.path.join(ROOT, "profiles", "host_shade"), "-I" + k,
           os.path.join(ROOT, "tests", "host_shade", "seam_host.cpp"), "-o", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return sc.Drivers(C.CDLL(str(out)), "host")


@pytest.mark.parametrize("name,w,h,spp,depth,must_call", si.CASES)
@pytest.mark.parametrize("backend", ["device_source", "oracle_drivers"])
def test_reference_integrator_on_hooked_seams(host_drivers, name, w, h, spp, depth, must_call, backend):
    import importlib
    crt = importlib.import_module("crust-render_amd")  # host-side scene description only: no kernel is launched
    if backend == "oracle_drivers" and name not in ("veach_mis", "sun_sky"):
        pytest.skip("plumbing check: two scenes are enough")
    desc = crt.usda.load(os.path.join(ROOT, "scenes", name + ".usda"), w, h)
    desc.settings["max_depth"] = depth
    o = ora_world.OracleRenderer(desc, crt.usda)
    drivers = host_drivers if backend == "device_source" else sc.oracle_drivers()
    host = si.SeamHost(si.OracleKernel(o.scene), si.DriversShade(drivers, o))
    for forward in (1, 0):
        si.check_against_own(name, host, o, spp, forward, must_call)
