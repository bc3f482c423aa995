"""The driver's bench.py contract: one JSON line with the required keys, roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "192",
                          "--height", "108", "--spp-per-step", "4", "--cpu-spp", "1"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mray/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # rays are counted as the reference counts them (stats.rs:150-152)
    assert d["config"]["rays_total"] == d["config"]["closest_hit"] + d["config"]["shadow_rays"]


@pytest.mark.gpu
def test_auxiliary_benchmarks_run():
    """bench_kernels.py (the reference's kernel probe) and bench_published.py print well-formed JSON."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench_kernels.py"), "--rays", "65536", "--reps", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout)
    assert set(d["scenes"]) == {"tri_spheres", "sphere_grid", "instances"}
    for e in d["scenes"].values():
        for q in ("intersect", "occluded"):
            assert e[q]["gpu_mray_s"] > 0 and e[q]["hits"] > 0 and e[q]["oracle_1t_mray_s"] > 0
