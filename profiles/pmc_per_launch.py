#!/usr/bin/env python3
"""Per-LAUNCH HBM traffic of one kernel family from rocprofv3 --pmc passes (gpurun_out/pmc_<tag>_3 = DRAM reads + L2 hit,
pmc_<tag>_4 = DRAM writes; profiles/run_pmc_r04.sh): the launches of a per-stage batch in order — bounce 0 traces camera
rays, the later ones incoherent bounce rays — each with its own duration (the pass's own timestamps), fabric bytes
(TCC_EA0_RDREQ_DRAM_32B_sum x 32 + TCC_EA0_WRREQ_WRITE_DRAM_32B_sum x 32) and fraction of the 8 TB/s HBM peak.
    python profiles/pmc_per_launch.py <tag> k_extend k_shadow"""
import collections, csv, glob, json, re, sys

tag, fams = sys.argv[1], sys.argv[2:] or ["k_extend", "k_shadow"]
HBM = 8.0e12


def rows(i):
    out = collections.OrderedDict()
    for f in sorted(glob.glob("gpurun_out/pmc_%s_%d/*/*_counter_collection.csv" % (tag, i))):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
            if not m:
                continue
            e = out.setdefault(int(r["Dispatch_Id"]), {"kernel": m.group(1) + (m.group(2) or ""), "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                       "grid": int(r["Grid_Size"])})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


rd, wr = rows(3), rows(4)
res = {}
for fam in fams:
    # the production instance of the family: the one with the most launches that is not the stats build (<true, ...>)
    inst = collections.Counter(e["kernel"] for e in rd.values() if e["kernel"].split("<")[0] == fam and not e["kernel"].startswith(fam + "<true"))
    if not inst:
        continue
    name = inst.most_common(1)[0][0]
    a = [e for e in rd.values() if e["kernel"] == name]
    b = [e for e in wr.values() if e["kernel"] == name]
    launches = []
    for k, (x, y) in enumerate(zip(a, b)):
        rbytes = x.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0) * 32
        wbytes = y.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 0.0) * 32
        ms = (x["ns"] + y["ns"]) / 2e6
        hit = x.get("TCC_HIT_sum", 0.0) / max(x.get("TCC_HIT_sum", 0.0) + x.get("TCC_MISS_sum", 0.0), 1.0)
        launches.append({"launch": k, "ms": round(ms, 3), "read_gb": round(rbytes / 1e9, 2), "write_gb": round(wbytes / 1e9, 2),
                         "hbm_frac": round((rbytes + wbytes) / (ms * 1e-3) / HBM, 4), "l2_hit": round(hit, 3)})
    tot_b = sum(l["read_gb"] + l["write_gb"] for l in launches) * 1e9
    tot_s = sum(l["ms"] for l in launches) * 1e-3
    res[name] = {"launches": launches, "all_launches_hbm_frac": round(tot_b / tot_s / HBM, 4) if tot_s else None}
print(json.dumps(res, indent=1))
