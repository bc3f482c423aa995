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
                          "--height", "108", "--spp-per-step", "4", "--cpu-spp", "1", "--other-configs-scale", "16"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
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
    # the bound is the tier the bytes are served from (bench.py, _roofline_entry): never a fraction above 1 of "hbm"
    assert r["bound"] in ("hbm", "latency/issue") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 34500.0) and (r["bound"] != "hbm" or r["frac"] <= 1.0)
    assert abs(r["hbm_frac_algorithmic"] - r["algorithmic_gb_s"] / 8000.0) < 1e-3 and abs(r["l2_frac"] - r["algorithmic_gb_s"] / 34500.0) < 1e-3
    assert r["frac"] <= 1.0 or r["bound"] != "hbm"  # never more than the peak of the bound it names
    for k, e in r["kernels"].items():  # the other stages of a per-stage pipeline, same fields
        assert k in ("k_extend", "k_shade", "k_shadow") and e["avg_launch_ms"] > 0 and e["achieved"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # the job's RayStats: all eight counters under their own names (a round-3 line carried them under the kernels' names)
    rs = d["config"]["ray_stats"]
    assert list(rs) == ["camera_rays", "closest_hit", "shadow_rays", "vertices", "rr_tested", "rr_killed", "ended_escaped", "ended_depth"]
    assert rs["closest_hit"] == d["config"]["closest_hit"] and rs["camera_rays"] == d["config"]["camera_rays"] == 192 * 108 * 4 * 2
    assert c["host_affinity"] >= c["cores"]
    # the same baseline on every hardware thread the process may use (what the reference's Rayon pool would take)
    a = d["cpu_baseline_all_cores"]
    quota = a["cgroup_cpu_quota"] or a["host_affinity"]
    assert a["kind"] == "port" and a["cores"] == min(a["host_affinity"], quota) and a["value"] > 0
    # the other BASELINE configurations on the same clock, after the headline: stress, MedCity at 2x its width, veach_mis,
    # openpbr_showcase — each with its rate, its shadow rays and the tier of its dominant kernel (here at 1/16 scale)
    oc = d["other_configs"]
    assert [o["workload"].split()[0] for o in oc] == ["stress", "PointInstancedMedCity", "veach_mis", "openpbr_showcase"]
    for o in oc:
        assert "error" not in o, o
        for k in ("workload", "value", "ms_per_step", "shadow_rays", "closest_hit", "roofline", "pipeline"):
            assert k in o, (o["workload"], k)
        assert o["value"] > 0 and o["ms_per_step"] > 0 and o["closest_hit"] > 0
        for k in ("bound", "tier", "frac", "hbm_measured_frac", "l2_hit", "kernel"):
            assert k in o["roofline"], (o["workload"], k)
        assert o["roofline"]["bound"] in ("hbm", "latency/issue") and o["roofline"]["kernel"] in ("k_path", "k_extend", "k_shade", "k_shadow")
    assert oc[0]["shadow_rays"] > 0 and oc[2]["shadow_rays"] > 0 and oc[3]["shadow_rays"] > 0 and oc[1]["shadow_rays"] == 0
    # what the job cost to set up and to hold (import once, per-rank commit, per-rank HBM)
    assert set(d["config"]["setup"]) == {"import_s", "commit_s", "hbm_used_gb_per_rank"} and d["config"]["setup"]["hbm_used_gb_per_rank"] > 0
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


@pytest.mark.gpu
def test_plain_invocation_with_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (no launcher, no WORLD_SIZE) starts two ranks itself and prints
    ONE line with n_gpus == 2; the gathered frame is bit-identical to the --gpus 1 frame of the same total spp.
    (gloo rehearsal: the two ranks share this box's one GPU; the driver's 8-GPU run takes the same path with RCCL.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--warmup", "1", "--width", "192", "--height", "108", "--no-cpu-baseline"]
    f2, f1 = str(tmp_path / "f2.npy"), str(tmp_path / "f1.npy")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                          "--spp-per-step", "4", "--dump-frame", f2] + common, capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and d2["value"] > 0
    assert "gather" in d2["config"]["sharding"] and d2["config"]["spp_per_step"] == 8
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--spp-per-step", "8",
                          "--dump-frame", f1] + common, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d1 = json.loads([l for l in res.stdout.splitlines() if l.strip()][0])
    assert d1["n_gpus"] == 1
    import numpy as np
    a, b = np.load(f1), np.load(f2)
    assert a.shape == (192 * 108, 3) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # the same rays were traced either way (stats.rs:150-152 counters, summed over ranks)
    assert d1["config"]["rays_total"] == d2["config"]["rays_total"]


def test_pmc_profile_is_reported_only_for_the_build_it_was_taken_on(tmp_path, monkeypatch):
    """roofline.traffic and the measured fractions come from committed PMC passes; a profile of OTHER kernel sources must
    never be reported for this build (the figure went stale with any kernel change). CPU-only."""
    sys.path.insert(0, ROOT)
    import bench
    here = bench.kernel_source_hash()
    assert len(here) == 16 and here == bench.kernel_source_hash()
    prof = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "PMC_PROFILE", str(prof))
    e, note = bench._pmc_for("cornellbox 1920x1080 64spp", "k_extend")
    assert e is None and "no PMC profile" in note
    entry = {"kernel": "k_extend<false, true>", "ea_dram_read_bytes_per_launch": 1.0, "ea_dram_write_bytes_per_launch": 2.0, "l2_hit_rate": 0.5,
             "valu_insts_per_launch": 3.0}
    wl = {"cornellbox 1920x1080 64spp": {"kernels": {"k_extend": entry}}}  # per workload, per kernel of the pipeline
    prof.write_text(json.dumps({"kernel_source_hash": "0" * 16, "workloads": wl}))
    e, note = bench._pmc_for("cornellbox 1920x1080 64spp", "k_extend")
    assert e is None and "not reported" in note
    prof.write_text(json.dumps({"kernel_source_hash": here, "git_commit": "abc", "workloads": wl}))
    e, note = bench._pmc_for("cornellbox 1920x1080 64spp", "k_extend")
    assert e == entry and "abc" in note
    assert bench._pmc_for("veach_mis 1920x1080 64spp", "k_extend")[0] is None    # another workload: nothing to report
    assert bench._pmc_for("cornellbox 1920x1080 64spp", "k_path")[0] is None     # the other pipeline's kernel: not what was profiled
    # the committed profile, if present, is well-formed
    real = bench._pmc_profile_path()  # the newest profiles/r*_pmc_bench.json
    if os.path.exists(real):
        d = json.load(open(real))
        assert len(d["kernel_source_hash"]) == 16 and d["workloads"]
        for w in d["workloads"].values():
            assert w["kernels"]
            for base, e in w["kernels"].items():
                for k in entry:
                    assert (e[k] > 0) if k != "kernel" else e[k].startswith(base), (base, k)

@pytest.mark.gpu
def test_a_batch_that_does_not_fit_is_halved_exactly_once(crt, monkeypatch):
    """bench.py's fallback for a part with less free HBM than the batch needs: the allocation of 8 spp fails
    (CRT_MAX_BATCH_SLOTS makes ensure_buffers ask for more than any device has above 500 000 slots: a genuine
    hipErrorOutOfMemory), 4 spp fit — fit_batch lands on exactly 4, not on 2: the failed allocation's error is not left
    behind for the next batch's launch check to find. The probe leaves nothing in the film or the counters."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("CRT_MAX_BATCH_SLOTS", "500000")
    r, _ = crt.load_usda(crt.scene_path("cornellbox"), 384, 216, 8)
    stream = torch.cuda.current_stream()
    with pytest.raises(crt.CrtError):
        r.render_samples(0, 8, stream)  # 663 552 paths -> 786 432 slots
    assert bench.fit_batch(crt, r, 8, 1, stream) == 4
    assert r.stats().camera_rays == 0 and not np.any(np.nan_to_num(r.film(), nan=1.0) != 1.0)  # 0 / 0 weight: cleared
    r.render_samples(0, 4, stream)
    torch.cuda.synchronize()
    assert r.stats().camera_rays == 384 * 216 * 4
    monkeypatch.delenv("CRT_MAX_BATCH_SLOTS")
    ref, _ = crt.load_usda(crt.scene_path("cornellbox"), 384, 216, 8)
    ref.render_samples(0, 4, stream)
    torch.cuda.synchronize()
    assert np.array_equal(r.image().view(np.uint32), ref.image().view(np.uint32))


def test_roofline_entry_names_the_tier_that_serves_the_bytes(monkeypatch):
    """bench.py's roofline object (CPU-only): the bound follows the MEASURED fabric traffic when PMC passes of this build
    exist — under 30 % of the HBM peak the kernel is priced against the L2 aggregate as "latency/issue", never as a
    fraction above 1 of "hbm" — and a lane's launch is charged its share of a whole batch's counters."""
    sys.path.insert(0, ROOT)
    import bench
    entry = {"kernel": "k_extend<false, true, 0>", "ea_dram_read_bytes_per_launch": 8.0e9, "ea_dram_write_bytes_per_launch": 6.0e9,
             "l2_hit_rate": 0.55, "valu_insts_per_launch": 7.0e9, "inst_active_frac": 0.42, "wait_any_frac": 0.41, "wait_inst_frac": 0.17}
    monkeypatch.setattr(bench, "_pmc_for", lambda key, kernel: (entry, "fake"))
    # one whole-batch launch: 160 GB algorithmic in 12.8 ms = 12.5 TB/s > the HBM peak; measured 14 GB = 0.137 of it
    e = bench._roofline_entry("k_extend", 160.0e9, 12.8 * 24, 24, 2, "w", 1)
    assert e["bound"] == "latency/issue" and e["peak"] == bench.L2_PEAK_GBS and 0.3 < e["frac"] < 0.4
    assert abs(e["hbm_measured_frac"] - 14.0e9 / 12.8e-3 / 1e9 / 8000.0) < 1e-3 and e["hbm_frac_algorithmic"] > 1.0
    # the same kernel as one of four overlapping lanes: a quarter of the counters per launch
    q = bench._roofline_entry("k_extend", 40.0e9, 11.0 * 96, 96, 2, "w", 1, traffic_scale=0.25)
    assert q["traffic"] == int(14.0e9 * 0.25) and q["bound"] == "latency/issue"
    # a streaming kernel whose measured traffic is above 30 % of the HBM peak is priced against HBM, fraction <= 1
    entry2 = dict(entry, ea_dram_read_bytes_per_launch=12.0e9, ea_dram_write_bytes_per_launch=6.0e9)
    monkeypatch.setattr(bench, "_pmc_for", lambda key, kernel: (entry2, "fake"))
    h = bench._roofline_entry("k_shade", 40.0e9, 6.8 * 24, 24, 2, "w", 1)
    assert h["bound"] == "hbm" and h["peak"] == bench.HBM_PEAK_GBS and h["frac"] <= 1.0
    # an HBM-bound kernel whose caches serve part of its algorithmic bytes (the 7 M-triangle scene's traversal: 10 TB/s
    # algorithmic, 0.35 of the peak measured): priced on the bytes that crossed the fabric, never above the peak
    entry3 = dict(entry, ea_dram_read_bytes_per_launch=99.8e9, ea_dram_write_bytes_per_launch=8.0e9)
    monkeypatch.setattr(bench, "_pmc_for", lambda key, kernel: (entry3, "fake"))
    b = bench._roofline_entry("k_extend", 384.0e9, 38.2 * 14, 14, 2, "w", 1)
    assert b["bound"] == "hbm" and b["frac"] <= 1.0 and abs(b["frac"] - b["hbm_measured_frac"]) < 1e-3
    assert b["hbm_frac_algorithmic"] > 1.0 and "measured" in b["achieved_basis"] and abs(b["frac"] - b["achieved"] / b["peak"]) < 1e-3
    # no PMC passes for this build: decided from the algorithmic rate alone
    monkeypatch.setattr(bench, "_pmc_for", lambda key, kernel: (None, "none"))
    n = bench._roofline_entry("k_extend", 160.0e9, 12.8 * 24, 24, 2, "w", 1)
    assert n["traffic"] is None and n["bound"] == "latency/issue" and n["frac"] < 1.0
