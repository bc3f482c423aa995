#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (gpurun_out/pmc_<tag>_<n>/*/…_counter_collection.csv) per kernel.
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads
(MI355X_MICROARCH.md §HBM), so the corrected read bytes are reported as 2x next to the raw figure."""
import csv, glob, collections, json, sys, re

tag = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        name = r["Kernel_Name"]
        m = re.search(r"(k_\w+)(<[^>]*>)?", name)
        k = (m.group(1) + (m.group(2) or "")) if m else re.sub(r"[^A-Za-z0-9_:<>]", "", name)[:60]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add((d, r["Dispatch_Id"]))  # a counter may come as several rows per dispatch
res = {}
for k, c in out.items():
    n = max(len(v) for v in launches[k].values())
    e = dict(launches=n, **{kk: vv for kk, vv in sorted(c.items())})
    if "TCC_HIT_sum" in c:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes_raw"] = c["FETCH_SIZE"] * 1024
        e["hbm_read_bytes_x2"] = c["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    # round 2: the fabric-side request counters in 32-byte units need no access-width correction
    if "TCC_EA0_RDREQ_DRAM_32B_sum" in c:
        e["ea_dram_read_bytes"] = c["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32
    if "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" in c:
        e["ea_dram_write_bytes"] = c["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] * 32
    if "TCC_EA0_RDREQ_sum" in c and "TCC_EA0_RDREQ_32B_sum" in c:
        e["ea_read_bytes_by_size"] = (c["TCC_EA0_RDREQ_32B_sum"] * 32 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 +
                                      c.get("TCC_EA0_RDREQ_128B_sum", 0) * 128)
    if "TCC_MISS_sum" in c:
        e["l2_miss_x128_bytes"] = c["TCC_MISS_sum"] * 128
    for kk in [x for x in e if x.endswith("_bytes") or x.endswith("_x2") or x.endswith("_raw") or x.endswith("_size")]:
        e[kk + "_per_launch"] = e[kk] / n
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        e["valu_active_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
        e["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        e["wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_ANY" in c:
            e["inst_active_frac"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    res[k] = e
print(json.dumps(res, indent=1))
