#!/usr/bin/env python3
"""Writes profiles/r02_pmc_bench.json — what bench.py reports as roofline.traffic / hbm_measured_frac / l2_hit /
valu_issue_frac — from PMC passes collected with profiles/run_pmc_r02.sh:
    python profiles/summarize_pmc_bench.py "<workload key>=<tag>" ...      e.g. "cornellbox 1920x1080 256spp=cb"
Each tag names gpurun_out/pmc_<tag>_<pass>/; the workload key is bench.py's (`<scene> <w>x<h> <spp per step>spp`).
The kernel reported is the pipeline's dominant production kernel: k_path<...> when the run launched it more often than
once (bench.py launches the OTHER pipeline for two untimed steps), else k_extend<false, ...> (the per-stage pipeline).
The file is stamped with the kernel source hash of the tree it is run in (the same tree the passes were taken on)
and the git commit; bench.py reports the numbers only while its own kernel source hash is the same."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

out = {"kernel_source_hash": kernel_source_hash(),
       "git_commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "?",
       "collected_with": "bash profiles/run_pmc_r02.sh <tag> bench.py --no-cpu-baseline ... (separate rocprofv3 --kernel-trace --pmc passes); "
                         "traffic = TCC_EA0_RDREQ_DRAM_32B_sum x 32 B + TCC_EA0_WRREQ_WRITE_DRAM_32B_sum x 32 B (calibration: profiles/r02_pmc_calibration.json)",
       "workloads": {}}
for arg in sys.argv[1:]:
    key, tag = arg.rsplit("=", 1)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarize_pmc.py"), tag], cwd=ROOT, capture_output=True, text=True)
    d = json.loads(res.stdout)
    kp = [n for n in d if n.startswith("k_path")]
    ke = [n for n in d if n.startswith("k_extend<false")]
    # the production pipeline's kernel has the launches of warm-up + timed steps; the other pipeline's, two steps' worth
    k = max(kp + ke, key=lambda n: d[n].get("SQ_WAVE_CYCLES", 0.0))
    e = d[k]
    out["workloads"][key] = {
        "kernel": k, "launches_profiled": e["launches"],
        "ea_dram_read_bytes_per_launch": e["ea_dram_read_bytes_per_launch"],
        "ea_dram_write_bytes_per_launch": e["ea_dram_write_bytes_per_launch"],
        "fetch_size_x2_bytes_per_launch": e.get("hbm_read_bytes_x2_per_launch"),
        "write_size_bytes_per_launch": e.get("hbm_write_bytes_per_launch"),
        "l2_hit_rate": e["l2_hit_rate"], "valu_active_frac": e["valu_active_frac"], "wait_any_frac": e["wait_any_frac"],
        "wait_inst_frac": e.get("wait_inst_frac"), "inst_active_frac": e.get("inst_active_frac"),
        "valu_insts_per_launch": e["SQ_INSTS_VALU"] / e["launches"],
    }
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_pmc_bench.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
