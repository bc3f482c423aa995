#!/usr/bin/env python3
"""Per-bounce picture of one batch: rays traced at each bounce (closest_hit at depth k minus depth k-1) next to the
durations of the k_extend / k_shade / k_shadow launches of a rocprofv3 --kernel-trace of the same batch.
    python profiles/per_bounce.py counts --scene cornellbox --spp 256 [--max 8]          -> rays per bounce (GPU)
    python profiles/per_bounce.py trace <kernel_trace.csv> [--skip N]                    -> launch durations in order"""
import argparse, csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def counts(a):
    import torch
    from __graft_entry__ import load_package
    crt = load_package()
    prev = None
    for d in range(a.max + 1):
        r, _ = crt.load_usda(crt.scene_path(a.scene), a.width, a.height, d)
        r.render_samples(0, a.spp)
        torch.cuda.synchronize()
        st = r.stats()
        cur = (st.closest_hit, st.shadow_rays, st.vertices)
        if prev is None: print("depth %d: closest %d shadow %d vertices %d" % ((d,) + cur))
        else: print("depth %d: +closest %d +shadow %d +vertices %d" % (d, cur[0] - prev[0], cur[1] - prev[1], cur[2] - prev[2]))
        prev = cur
        del r

def work(a):
    """Traversal work per ray at each bounce: the stats build's counters at depth k minus depth k-1."""
    import torch
    from __graft_entry__ import load_package
    crt = load_package()
    prev = None
    for d in range(a.max + 1):
        r, _ = crt.load_usda(crt.scene_path(a.scene), a.width, a.height, d)
        ext, _sh = r.render_samples_stats(0, a.spp)
        torch.cuda.synchronize()
        cur = dict(rays=int(ext.rays), nodes=sum(ext.nodes), leaves=sum(ext.leaves), packets=sum(ext.packets), prims=sum(ext.prims),
                   accepted=int(ext.accepted_hits), descents=int(ext.instance_descents),
                   waves=[int(x) for x in ext.phase_waves], lanes=[int(x) for x in ext.phase_lanes], cyc=[int(x) for x in ext.phase_cycles])
        if prev is None: dlt = cur
        else: dlt = {k: (cur[k] - prev[k] if not isinstance(cur[k], list) else [x - y for x, y in zip(cur[k], prev[k])]) for k in cur}
        n = max(dlt["rays"], 1)
        util = ["%s %.2f" % (crt.CrtTravStats.PHASES[i][:6], dlt["lanes"][i] / (64.0 * dlt["waves"][i])) for i in range(7) if dlt["waves"][i]]
        execs = ["%s %.2f" % (crt.CrtTravStats.PHASES[i][:6], dlt["waves"][i] * 64.0 / n) for i in range(7) if dlt["waves"][i]]
        cyc = ["%s %.0f%%" % (crt.CrtTravStats.PHASES[i][:6], 100.0 * dlt["cyc"][i] / max(sum(dlt["cyc"]), 1)) for i in range(7) if dlt["waves"][i]]
        print("bounce %d: rays %d | per ray: nodes %.2f leaves %.2f packets %.2f scalar %.2f accepted %.2f descents %.3f" % (
            d, dlt["rays"], dlt["nodes"] / n, dlt["leaves"] / n, dlt["packets"] / n, dlt["prims"] / n, dlt["accepted"] / n, dlt["descents"] / n))
        print("   lane utilisation: " + ", ".join(util))
        print("   wave-executions x 64 per ray: " + ", ".join(execs))
        print("   cycles: " + ", ".join(cyc))
        prev = cur
        del r

def trace(a):
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = []
    for r in rows:
        n = r["Kernel_Name"]
        for k in ("k_generate", "k_extend", "k_shade", "k_shadow", "k_resolve", "k_path"):
            if k in n:
                out.append((k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    # batches start at k_generate (or k_path)
    b = -1
    for k, ms in out:
        if k in ("k_generate", "k_path"): b += 1; print("--- batch %d" % b)
        if ms > 0.02: print("  %-10s %8.3f ms" % (k, ms))

ap = argparse.ArgumentParser()
sub = ap.add_subparsers(dest="cmd")
c = sub.add_parser("counts"); c.add_argument("--scene", default="cornellbox"); c.add_argument("--spp", type=int, default=256)
c.add_argument("--width", type=int, default=1920); c.add_argument("--height", type=int, default=1080); c.add_argument("--max", type=int, default=8)
wk = sub.add_parser("work"); wk.add_argument("--scene", default="cornellbox"); wk.add_argument("--spp", type=int, default=16)
wk.add_argument("--width", type=int, default=1920); wk.add_argument("--height", type=int, default=1080); wk.add_argument("--max", type=int, default=4)
t = sub.add_parser("trace"); t.add_argument("csv")
a = ap.parse_args()
{"counts": counts, "work": work, "trace": trace}[a.cmd](a)
