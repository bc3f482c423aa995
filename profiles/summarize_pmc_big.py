#!/usr/bin/env python3
"""The out-of-cache probe's share of a `run_pmc_r02.sh <tag> bench_kernels.py --big 512 --no-oracle --rays N --reps 2` run:
the LAST two launches of each query kernel are the 7.08 M-triangle scene (bench_kernels.py runs it last).
    python profiles/summarize_pmc_big.py <tag> <intersect ms per launch> <occluded ms per launch>
The launch durations are the un-profiled ones of the same command (gpurun_out/<tag>_kernel_probe_big.json)."""
import csv, glob, collections, json, sys
tag, ms = sys.argv[1], {"intersect": float(sys.argv[2]), "occluded": float(sys.argv[3])}
VALU_PEAK = 256 * 4 * 2.4e9 / 2
res = {"command": "bash profiles/run_pmc_r02.sh %s bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 2 (the last two "
                  "launches of each query kernel = the 7.08 M-triangle scene, 8.4 M rays per launch)" % tag, "kernels": {}}
for kind in ("intersect", "occluded"):
    tot = collections.defaultdict(float)
    for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(d)) if kind + "_n_kernel<false" in r["Kernel_Name"]]
        last = sorted({int(r["Dispatch_Id"]) for r in rows})[-2:]
        for r in rows:
            if int(r["Dispatch_Id"]) in last:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]) / 2
    rd, wr = tot["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32, tot["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] * 32
    t = ms[kind] * 1e-3
    res["kernels"][kind] = {
        "launch_ms": ms[kind], "ea_dram_read_GB": round(rd / 1e9, 3), "ea_dram_write_GB": round(wr / 1e9, 3),
        "traffic_GB_s": round((rd + wr) / t / 1e9, 1), "hbm_frac": round((rd + wr) / t / 8e12, 3),
        "l2_hit": round(tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"], 1), 3),
        "valu_active_frac": round(tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1), 3),
        "inst_active_frac": round(tot["SQ_ACTIVE_INST_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1), 3),
        "wait_any_frac": round(tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1), 3),
        "valu_issue_frac": round(tot["SQ_INSTS_VALU"] / t / VALU_PEAK, 3),
        "read_by_size_GB": {"32B": round(tot["TCC_EA0_RDREQ_32B_sum"] * 32 / 1e9, 3), "64B": round(tot["TCC_EA0_RDREQ_64B_sum"] * 64 / 1e9, 3),
                            "128B": round(tot["TCC_EA0_RDREQ_128B_sum"] * 128 / 1e9, 3)},
    }
print(json.dumps(res, indent=1))
