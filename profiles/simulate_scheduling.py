#!/usr/bin/env python3
"""Offline study of wave scheduling policies for the BVH4 traversal kernel (not part of the product or tests).

Takes real per-ray work sequences from the oracle's phase trace (N node, P packet, s sphere, I instance entry,
X instance exit) for camera rays and for cosine-distributed bounce rays of the cornellbox scene, in the pixel
order the renderer uses (16x16 tiles, 64 consecutive pixels per wave), and replays them through models of
 A  the shipped kernel: while-while step() per lane, refill when >= 32 lanes idle
 B  majority-vote phase scheduling with register-resident state (one ray per lane)
 C  a per-wave pool of R rays in LDS, 64 rays of one phase gathered per step
Cost model: VALU instructions per wave-level execution of a phase (from the gfx950 disassembly, rounded).
Prints wave-instructions per ray and mean lane utilisation per policy."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import ora  # noqa: E402
import ora_world  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

COST = dict(F=150, N=140, P=300, s=80, I=250, X=200, E=120, loop=30, queue=50)


def traces(scene, rays):
    L = ora.lib()
    L.ora_set_trace.argtypes = [C.c_char_p, C.c_size_t]
    L.ora_trace_len.restype = C.c_size_t
    buf = C.create_string_buffer(4096)
    out = []
    for r in rays:
        L.ora_set_trace(buf, 4096)
        scene.intersect(ora.ray(r[0:3], r[3:6], mask=4))
        out.append(buf.raw[: L.ora_trace_len()].decode())
    L.ora_set_trace(None, 0)
    return out


def to_steps(tr):
    """Group a trace into the shipped kernel's step() calls: each call is one scalar prim / one instance exit /
    a run of N followed by the packets of one leaf."""
    steps, i = [], 0
    while i < len(tr):
        c = tr[i]
        if c in "sIX":
            steps.append(c); i += 1
        else:
            j = i
            while j < len(tr) and tr[j] == "N":
                j += 1
            k = j
            while k < len(tr) and tr[k] == "P":
                k += 1
            steps.append(tr[i:k]); i = k
    return steps


def sim_A(all_steps, refill=32):
    """while-while: each iteration every active lane runs one step(); wave cost = sum over the branch kinds
    present (+ max N run + max P run)."""
    instr = useful = 0
    nxt = 0
    lanes = [None] * 64  # (steps, pos)
    n = len(all_steps)
    while True:
        idle = [k for k in range(64) if lanes[k] is None]
        if nxt < n and (len(idle) == 64 or len(idle) >= refill):
            take = idle[: n - nxt]
            for k in take:
                lanes[k] = [all_steps[nxt], 0]; nxt += 1
            instr += COST["F"]; useful += COST["F"] * len(take)
        act = [k for k in range(64) if lanes[k] is not None]
        if not act:
            break
        instr += COST["loop"]; useful += COST["loop"] * len(act)
        maxN = maxP = 0
        kinds = {}
        emit = 0
        for k in act:
            st, pos = lanes[k]
            if pos >= len(st):
                emit += 1; lanes[k] = None; continue
            s = st[pos]; lanes[k][1] += 1
            if s in ("s", "I", "X"):
                kinds[s] = kinds.get(s, 0) + 1
            else:
                nN, nP = s.count("N"), s.count("P")
                maxN, maxP = max(maxN, nN), max(maxP, nP)
                useful += nN * COST["N"] + nP * COST["P"]
        instr += maxN * COST["N"] + maxP * COST["P"]
        for s, c in kinds.items():
            instr += COST[s]; useful += COST[s] * c
        if emit:
            instr += COST["E"]; useful += COST["E"] * emit
    return instr, useful


def sim_affine(all_tr, rows=2, rare_min=16, overhead=20, fetch_min=32, chain_min=0, retire=False, emit_min=None):
    """State in LDS but slot r is only ever processed by lane r % 64 (conflict-free LDS, no gather list): a lane
    runs the chosen phase for at most one of its `rows` slots per step. retire=True: ONE phase emits a finished ray's hit
    and, in the same step, fetches and sets up the next ray into the slot (while input remains) — the slot does not wait
    for a second scheduling round, at the price of a step that runs both pieces of code. emit_min: lanes that must be
    waiting to emit before the emit / retire phase is chosen over node / packet work (the shipped kernel: 40)."""
    instr = useful = 0
    n = len(all_tr); nxt = 0
    slots = [[None] * rows for _ in range(64)]
    while True:
        free = [(l, k) for l in range(64) for k in range(rows) if slots[l][k] is None]
        live = [slots[l][k] for l in range(64) for k in range(rows) if slots[l][k] is not None]
        lanes_free = len({l for l, _ in free})
        if nxt < n and ((lanes_free >= fetch_min and not (retire and live)) or not live):
            got = 0
            seen = set()
            for l, k in free:
                if l in seen or nxt >= n:
                    continue
                seen.add(l); slots[l][k] = [all_tr[nxt], 0]; nxt += 1; got += 1
            instr += COST["F"] + overhead; useful += COST["F"] * got
            continue
        if not live:
            break
        def ph_of(r):
            return r[0][r[1]] if r[1] < len(r[0]) else "E"
        cnt = {}
        for l in range(64):
            phs = {ph_of(r) for r in slots[l] if r is not None}
            for p in phs:
                cnt[p] = cnt.get(p, 0) + 1
        common = {p: v for p, v in cnt.items() if p in "NP"}
        rare = {p: v for p, v in cnt.items() if p not in "NP" and v >= (emit_min if (emit_min and p == "E") else rare_min)}
        cand = rare or common or cnt
        ph = max(cand.items(), key=lambda kv: kv[1])[0]
        k_run = 0
        ran = []
        for l in range(64):
            for k in range(rows):
                r = slots[l][k]
                if r is not None and ph_of(r) == ph:
                    if ph == "E":
                        slots[l][k] = None
                        if retire and nxt < n:
                            slots[l][k] = [all_tr[nxt], 0]; nxt += 1
                    else:
                        r[1] += 1
                        ran.append(r)
                    k_run += 1
                    break
        if ph == "E" and retire:
            instr += COST["F"]; useful += COST["F"] * k_run
        instr += COST[ph] + overhead; useful += COST[ph] * k_run
        # chaining: while enough of the rays that just ran agree on their next common phase, run it at once
        # (state stays in registers: no scheduling round, no LDS round trip)
        while chain_min and ran:
            follow = {}
            for r in ran:
                p2 = ph_of(r)
                if p2 in "NP":
                    follow.setdefault(p2, []).append(r)
            if not follow:
                break
            p2, rs = max(follow.items(), key=lambda kv: len(kv[1]))
            if len(rs) < chain_min:
                break
            for r in rs:
                r[1] += 1
            instr += COST[p2] + 10; useful += COST[p2] * len(rs)
            ran = rs
    return instr, useful


def sim_pool(all_tr, pool=128, lanes=64, rare_min=16, overhead=None, fetch_min=64):
    """Pool of `pool` rays per wave; each iteration runs ONE phase for up to 64 rays in that phase. pool == 64
    with overhead 0 models policy B (register state, majority vote): a ray never changes lane."""
    overhead = COST["queue"] if overhead is None else overhead
    instr = useful = 0
    n = len(all_tr)
    nxt = 0
    live = []  # [trace, pos]
    while True:
        if nxt < n and pool - len(live) >= min(fetch_min, n - nxt) and (pool - len(live) >= fetch_min or not live):
            take = min(lanes, n - nxt, pool - len(live))
            for _ in range(take):
                live.append([all_tr[nxt], 0]); nxt += 1
            instr += COST["F"] + overhead; useful += COST["F"] * take
            continue
        if not live:
            break
        by = {}
        for r in live:
            ph = r[0][r[1]] if r[1] < len(r[0]) else "E"
            by.setdefault(ph, []).append(r)
        # common phases by majority; rare phases wait until rare_min rays queue up or nothing else is runnable
        common = {p: v for p, v in by.items() if p in "NP"}
        rare = {p: v for p, v in by.items() if p not in "NP" and len(v) >= rare_min}
        cand = rare or common or by
        if common and not rare:
            cand = common
        ph, rs = max(cand.items(), key=lambda kv: len(kv[1]))
        rs = rs[:lanes]
        instr += COST[ph] + overhead; useful += COST[ph] * len(rs)
        for r in rs:
            if ph == "E":
                live.remove(r)
            else:
                r[1] += 1
    return instr, useful


def main():
    crt = load_package()
    w, h = 128, 64
    desc = crt.usda.load(os.path.join(ROOT, "scenes", "cornellbox.usda"), w, h)
    o = ora_world.OracleRenderer(desc, crt.usda)
    cam = o.job.camera
    pix = crt.shard.shard_pixels(w, h, 0, 1)  # tile order, as the renderer traces
    rng = np.random.default_rng(1)
    prim = np.zeros((len(pix), 8), dtype=np.float32)
    L = ora.lib()
    for k, p in enumerate(pix):
        i, j = int(p) % w, int(p) // w
        r = ora.Ray()
        L.ora_camera_get_ray(C.byref(cam), (i + 0.5) / w, (j + 0.5) / h, 0.5, 0.5, 0.0, C.byref(r))
        prim[k, 0:3] = r.origin.np(); prim[k, 3:6] = r.dir.np()
    prim[:, 7] = np.array([1], dtype=np.uint32).view(np.float32)
    hf, ids, front = o.scene.intersect_n(prim, 0.001, float("inf"))
    hit = ids[:, 0] != 0xFFFFFFFF
    sec = []
    for k in np.nonzero(hit)[0]:
        nrm = hf[k, 1:4].astype(np.float64)
        pnt = prim[k, 0:3] + prim[k, 3:6] * hf[k, 0]
        a = np.array([1.0, 0, 0]) if abs(nrm[0]) < 0.9 else np.array([0, 1.0, 0])
        t = np.cross(nrm, a); t /= np.linalg.norm(t); b = np.cross(nrm, t)
        u1, u2 = rng.random(2)
        rr, ph = np.sqrt(u1), 2 * np.pi * u2
        d = t * rr * np.cos(ph) + b * rr * np.sin(ph) + nrm * np.sqrt(1 - u1)
        sec.append(np.concatenate([pnt, d]))
    sec = np.array(sec, dtype=np.float32)
    for label, rays in (("camera rays", prim[:, :6]), ("bounce rays (cosine about the hit normal)", sec)):
        tr = traces(o.scene, rays)
        steps = [to_steps(t) for t in tr]
        nN = sum(t.count("N") for t in tr) / len(tr); nP = sum(t.count("P") for t in tr) / len(tr)
        nI = sum(t.count("I") for t in tr) / len(tr)
        print(f"== {label}: {len(tr)} rays, {nN:.2f} nodes, {nP:.2f} packets, {nI:.3f} instance entries per ray")
        for name, (ins, use) in (
            ("A while-while, refill 32", sim_A(steps)),
            ("B majority vote, registers", sim_pool(tr, pool=64, overhead=10)),
            ("C pool 96 in LDS", sim_pool(tr, pool=96)),
            ("C pool 128 in LDS", sim_pool(tr, pool=128)),
            ("C pool 192 in LDS", sim_pool(tr, pool=192)),
            ("C pool 96, fetch>=32", sim_pool(tr, pool=96, fetch_min=32)),
            ("C pool 96, fetch>=16", sim_pool(tr, pool=96, fetch_min=16)),
            ("C pool 128, fetch>=32", sim_pool(tr, pool=128, fetch_min=32)),
            ("C pool 128, fetch>=32, rare>=32", sim_pool(tr, pool=128, fetch_min=32, rare_min=32)),
            ("C pool 128, fetch>=32, rare>=8", sim_pool(tr, pool=128, fetch_min=32, rare_min=8)),
            ("C pool 160, fetch>=32", sim_pool(tr, pool=160, fetch_min=32)),
            # round 4: the wave-shared pool with >= 3 rays per lane's worth of slots and a ballot / prefix
            # gather list per phase, at the gather's realistic price (the list, the indexed LDS reads of the state groups at
            # 2-4-way bank conflicts, the cold state that must then live in LDS too: ~90 instructions per step instead of 50),
            # and its limit — one pool shared by the workgroup's four waves (768 slots)
            ("C pool 192, gather 90", sim_pool(tr, pool=192, fetch_min=32, overhead=90)),
            ("C pool 256, gather 90", sim_pool(tr, pool=256, fetch_min=32, overhead=90)),
            ("C pool 768 (workgroup), gather 90", sim_pool(tr, pool=768, fetch_min=32, overhead=90)),
            ("C pool 768 (workgroup), gather 50", sim_pool(tr, pool=768, fetch_min=32, overhead=50)),
            ("D lane-affine 2 rows, ovh 50", sim_affine(tr, rows=2, overhead=50)),
            ("D lane-affine 2 rows", sim_affine(tr, rows=2)),
            ("D lane-affine 3 rows", sim_affine(tr, rows=3)),
            ("D lane-affine 4 rows", sim_affine(tr, rows=4)),
            ("D 2 rows, overhead 90", sim_affine(tr, rows=2, overhead=90)),
            ("D 2 rows, ovh 90, chain>=24", sim_affine(tr, rows=2, overhead=90, chain_min=24)),
            ("D 2 rows, ovh 90, chain>=32", sim_affine(tr, rows=2, overhead=90, chain_min=32)),
            ("D 2 rows, ovh 90, chain>=16", sim_affine(tr, rows=2, overhead=90, chain_min=16)),
            ("D 2 rows, ovh 90, chain>=40", sim_affine(tr, rows=2, overhead=90, chain_min=40)),
            # round 4, for the next round: emit and fetch as ONE phase (a finished ray's slot is refilled in the step that
            # reports its hit), against the shipped form with the same emit threshold
            ("D shipped form, emit>=40", sim_affine(tr, rows=2, overhead=90, chain_min=24, fetch_min=40, emit_min=40)),
            ("D retire (emit+fetch), >=40", sim_affine(tr, rows=2, overhead=90, chain_min=24, fetch_min=40, emit_min=40, retire=True)),
            ("D retire (emit+fetch), >=24", sim_affine(tr, rows=2, overhead=90, chain_min=24, fetch_min=40, emit_min=24, retire=True)),
            ("D retire (emit+fetch), >=16", sim_affine(tr, rows=2, overhead=90, chain_min=24, fetch_min=40, emit_min=16, retire=True)),
            # more rows per lane at the shipped form's realistic scheduling price (LDS for them exists only at fewer waves per
            # SIMD: 3 rows x 3 waves = 9 rays per SIMD lane against the shipped 2 x 4 = 8)
            ("D shipped form, 3 rows", sim_affine(tr, rows=3, overhead=90, chain_min=24, fetch_min=40, emit_min=40)),
            ("D shipped form, 4 rows", sim_affine(tr, rows=4, overhead=90, chain_min=24, fetch_min=40, emit_min=40)),
        ):
            print(f"   {name:28s} wave-instr/ray {ins / len(tr):7.2f}   lane utilisation {use / (64.0 * ins):5.3f}")


if __name__ == "__main__":
    main()
