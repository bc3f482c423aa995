"""Function-level parity of the shading functions WITHOUT a GPU: the device
functions of kernels/shade.hip.h — the code k_shade, k_path and the crt_material_*_n / crt_light_*_n kernels run —
compiled as host C++ (tests/host_shade/seam_host.cpp behind profiles/host_shade/hip/hip_runtime.h, a stand-in for the
HIP names they use) and compared bit for bit with the oracle's functions on seeded random calls per lobe class and
light kind. The GPU run of the same cases is tests/test_gpu_shading_seam.py. Neither side here is the product path:
this pins the SOURCE of the shading arithmetic against the oracle on every round, GPU or not."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import seam_cases as sc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 20000  # calls per lobe class and method (the GPU test runs 100 000)


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = tmp_path_factory.mktemp("seam_host") / "libseam_host.so"
    k = os.path.join(ROOT, "crust-render_amd", "csrc", "kernels")
    cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-attributes",
           "-I" + os.path.join(ROOT, "profiles", "host_shade"), "-I" + k,
           os.path.join(ROOT, "tests", "host_shade", "seam_host.cpp"), "-o", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return sc.Drivers(C.CDLL(str(out)), "host")


@pytest.fixture(scope="module")
def oracle():
    return sc.oracle_drivers()


@pytest.mark.parametrize("cls", sc.CLASSES)
def test_material_methods_of_the_device_source_match_the_oracle(host, oracle, cls):
    rng = np.random.default_rng(1000 + sc.CLASSES.index(cls))
    mats = sc.materials(cls, 257, rng)
    q = sc.shade_queries(N, len(mats), rng)
    for name in ("scatter", "eval", "emitted"):
        got, want = getattr(host, name)(mats, q), getattr(oracle, name)(mats, q)
        bad = sc.mismatches(got, want)
        assert len(bad) == 0, (cls, name, len(bad), got[bad[:2]], want[bad[:2]], q[bad[:2]])
    s, e = oracle.scatter(mats, q), oracle.eval(mats, q)
    if cls == "emissive":  # Emissive never scatters and has no continuous BSDF (emissive.rs:30-38)
        assert s["some"].sum() == 0 and e["some"].sum() == 0 and np.abs(oracle.emitted(mats, q)).sum() > 0
    else:  # the cases reach the arms they are meant to
        assert s["some"].mean() > 0.5 and e["some"].mean() > 0.8 and (e["value"].sum(axis=1) > 0).mean() > 0.2
        if cls in ("transmission", "dispersion", "subsurface_medium", "everything"):
            below = np.einsum("ij,ij->i", s["dir"], q["normal"]) < 0
            assert (below & (s["some"] == 1)).sum() > N // 200  # refracted continuations
        if cls in ("transmission", "subsurface_medium"):
            assert (s["flags"] & 2).sum() > 0  # rays that enter an interior medium
        if cls == "thin_wall":
            assert (s["flags"] & 1).sum() > N // 100  # delta pass-through samples


def test_light_methods_of_the_device_source_match_the_oracle(host, oracle):
    rng = np.random.default_rng(77)
    table = sc.lights(rng)
    q = sc.light_queries(4 * N, table, rng)
    for name in ("light_sample", "light_pdf", "light_escaped"):
        got, want = getattr(host, name)(table, q), getattr(oracle, name)(table, q)
        bad = sc.mismatches(got, want)
        assert len(bad) == 0, (name, len(bad), got[bad[:2]], want[bad[:2]], q[bad[:2]])
    kinds = table["kind"][q["light"]]
    s, esc, pdf = oracle.light_sample(table, q), oracle.light_escaped(table, q), oracle.light_pdf(table, q)
    assert s["some"].all() and np.isinf(s["distance"][kinds >= 2]).all() and np.isfinite(s["distance"][kinds < 2]).all()
    assert esc["some"][kinds == 3].all() and not esc["some"][kinds < 2].any() and 0 < esc["some"][kinds == 2].mean() < 1
    assert (pdf[kinds >= 2] == 0).all() and (pdf[kinds < 2] > 0).mean() > 0.9


def test_out_of_range_indices_answer_none(host, oracle):
    rng = np.random.default_rng(5)
    mats = sc.materials("base", 4, rng)
    q = sc.shade_queries(64, 4, rng)
    q["material"][::2] = 4 + rng.integers(0, 1000, size=32)
    for d in (host, oracle):
        assert d.scatter(mats, q)["some"][::2].sum() == 0 and d.eval(mats, q)["some"][::2].sum() == 0
        assert np.abs(d.emitted(mats, q)[::2]).sum() == 0
