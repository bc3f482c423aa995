#!/usr/bin/env python3
"""Copies the outputs of `bash profiles/final_r02.sh <tag>` (merged into gpurun_out/) to their committed names under
profiles/, stamping the PMC profile and the bench lines with the current git commit (the GPU box has no .git).
    python profiles/install_final.py <tag>"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
tag = sys.argv[1]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
g = lambda n: os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, n))
p = lambda n: os.path.join(ROOT, "profiles", n)
d = json.load(open(g("pmc_bench.json")))
assert d["kernel_source_hash"] == bench.kernel_source_hash(), "the profile is of other kernel sources than this tree's"
d["git_commit"] = head
json.dump(d, open(p("r02_pmc_bench.json"), "w"), indent=1)
for src, dst in (("bench_default.json", "r02_bench_default.json"), ("bench_medcity_3840x2160.json", "r02_bench_medcity_3840x2160.json")):
    line = [x for x in open(g(src)) if x.startswith("{")][-1]
    x = json.loads(line)
    x["roofline"]["pmc_source"] = "profiles/r02_pmc_bench.json (git %s)" % head
    open(p(dst), "w").write(json.dumps(x) + "\n")
    r = x["roofline"]
    print(dst, x["value"], x["ms_per_step"], r["achieved"], r["frac"], r["traffic"], r["hbm_measured_frac"], r["l2_hit"], r["valu_issue_frac"],
          r["wave_time"], r["avg_launch_ms"], r["kernel_ms"], r.get("other_pipeline"))
shutil.copy(glob.glob(g("stats") + "/*/*kernel_stats.csv")[0], p("r02_rocprofv3_kernel_stats_bench_default.csv"))
shutil.copy(g("pmc_cornellbox.json"), p("r02_pmc_summary_cornellbox_steps4.json"))
shutil.copy(g("pmc_medcity.json"), p("r02_pmc_summary_medcity_3840x2160_steps2.json"))
shutil.copy(g("published_default_renders.json"), p("r02_published_default_renders.json"))
print(open(p("r02_rocprofv3_kernel_stats_bench_default.csv")).read().splitlines()[1][-90:])
