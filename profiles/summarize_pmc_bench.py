#!/usr/bin/env python3
"""Writes profiles/r04_pmc_bench.json — what bench.py reports as roofline.traffic / hbm_measured_frac / l2_hit /
valu_issue_frac, per kernel of the pipeline — from PMC passes collected with profiles/run_pmc_r04.sh:
    python profiles/summarize_pmc_bench.py "<workload key>=<tag>" ...      e.g. "cornellbox 1920x1080 256spp=cb"
Each tag names gpurun_out/pmc_<tag>_<pass>/; the workload key is bench.py's (`<scene> <w>x<h> <spp per step>spp`).
Per workload one entry per kernel FAMILY the run launched (k_path, k_extend, k_shade, k_shadow, k_generate): of a
family's template instances the one with the most wave cycles (bench.py also launches the OTHER pipeline for two untimed
steps and the stats build for one; their instances are the small ones).
The file is stamped with the kernel source hash of the tree it is run in (the same tree the passes were taken on)
and the git commit; bench.py reports the numbers only while its own kernel source hash is the same."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

OUT = os.path.join(ROOT, "profiles", "r04_pmc_bench.json")
FAMILIES = ("k_path", "k_extend", "k_shade", "k_shadow", "k_generate")

out = {"kernel_source_hash": kernel_source_hash(),
       "git_commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "?",
       "collected_with": "bash profiles/run_pmc_r04.sh <tag> bench.py --no-cpu-baseline ... (separate rocprofv3 --kernel-trace --pmc passes); "
                         "traffic = TCC_EA0_RDREQ_DRAM_32B_sum x 32 B + TCC_EA0_WRREQ_WRITE_DRAM_32B_sum x 32 B (calibration: profiles/r02_pmc_calibration.json)",
       "workloads": {}}
if os.path.exists(OUT) and "--merge" in sys.argv:  # keep the workloads of an earlier call on the same build
    old = json.load(open(OUT))
    if old.get("kernel_source_hash") == out["kernel_source_hash"]:
        out["workloads"] = old["workloads"]
for arg in [a for a in sys.argv[1:] if not a.startswith("--")]:
    key, tag = arg.rsplit("=", 1)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarize_pmc.py"), tag], cwd=ROOT, capture_output=True, text=True)
    d = json.loads(res.stdout)
    kernels = {}
    for fam in FAMILIES:
        inst = [n for n in d if n.split("<")[0] == fam and d[n].get("SQ_WAVE_CYCLES", 0.0) > 0]
        if not inst:
            continue
        k = max(inst, key=lambda n: d[n]["SQ_WAVE_CYCLES"])
        e = d[k]
        kernels[fam] = {
            "kernel": k, "launches_profiled": e["launches"],
            "ea_dram_read_bytes_per_launch": e["ea_dram_read_bytes_per_launch"],
            "ea_dram_write_bytes_per_launch": e["ea_dram_write_bytes_per_launch"],
            "fetch_size_x2_bytes_per_launch": e.get("hbm_read_bytes_x2_per_launch"),
            "write_size_bytes_per_launch": e.get("hbm_write_bytes_per_launch"),
            "l2_hit_rate": e["l2_hit_rate"], "valu_active_frac": e["valu_active_frac"], "wait_any_frac": e["wait_any_frac"],
            "wait_inst_frac": e.get("wait_inst_frac"), "inst_active_frac": e.get("inst_active_frac"),
            "valu_insts_per_launch": e["SQ_INSTS_VALU"] / e["launches"],
        }
    out["workloads"][key] = {"kernels": kernels}
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out, indent=1))
