#!/usr/bin/env python3
"""Static check of the compiler-managed spill state of a gfx950 kernel (no GPU needed).

    python profiles/isa_spill_check.py <file.s> [kernel-name-substring ...]

For every kernel of the assembly whose (mangled) name contains one of the substrings (default: all), builds the
control-flow graph of its body and runs two forward MUST analyses to a fixpoint:

  * SGPR spills to VGPR lanes: a `v_readlane_b32 sX, vN, L` of a spill register vN (one that is the destination of
    some v_writelane_b32) must be dominated, on EVERY path from the kernel entry, by a `v_writelane_b32 vN, sY, L`
    of the same lane (or by a reload of vN from a scratch slot whose stored image had that lane written).
  * VGPR spills to scratch: a `scratch_load_dword vN, off, off offset:K ; ... Reload` must be dominated on every
    path by a `scratch_store_dword off, vM, off offset:K ; ... Spill`.

It also lists whole-register spills of the lane-spill VGPRs themselves and checks they are bracketed by
`s_or_saveexec_b64 ..., -1` (all lanes enabled): a partial-EXEC store of such a register would drop the SGPRs parked in
the inactive lanes.

What it cannot see: whether the EXEC mask at a VGPR reload is a subset of the mask at the matching spill (per-lane
liveness is the register allocator's contract, not visible in the text)."""
import re, sys, collections

def kernels(text):
    """name -> list of lines of the body (between the label and .Lfunc_end)."""
    out = {}
    cur = None
    for l in text.split("\n"):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", l)
        if m:
            cur = m.group(1); out[cur] = []; continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                cur = None; continue
            out[cur].append(l)
    return out

BR = re.compile(r"^\s+(s_branch|s_cbranch_\w+)\s+(\.LBB\w+)")
LBL = re.compile(r"^(\.LBB\w+):")

def analyse(name, lines):
    # basic blocks
    long_target = None
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []; order.append(cur)
    first_line = {}
    for ln, l in enumerate(lines):
        m = LBL.match(l)
        if m:
            nxt = m.group(1)
            if nxt not in blocks:
                blocks[nxt] = []; order.append(nxt); first_line[nxt] = ln + 2
            blocks[cur].append(("fall", nxt))
            cur = nxt
            continue
        s = l.strip()
        if not l.startswith("\t") or not s or s[0] in ".;":
            continue
        blocks[cur].append(("ins", s))
        op = s.split()[0]
        # a long branch (branch relaxation of a kernel this size): s_getpc_b64 / s_add_u32 sN, sN, (.LBBx-.Lpost_getpcK)&... /
        # s_addc_u32 / s_setpc_b64 — the target is named in the s_add_u32
        m = re.search(r"\((\.LBB\w+)-\.Lpost_getpc\d+\)&", s)
        if m: long_target = m.group(1)
        if op.startswith("s_setpc"):
            blocks[cur].append(("jump", long_target))
        if op in ("s_endpgm",) or op == "s_branch" or op.startswith("s_cbranch") or op.startswith("s_setpc"):
            # start a fresh anonymous block after a terminator
            nxt = "%s.after%d" % (cur, len(blocks))
            blocks[nxt] = []; order.append(nxt); first_line[nxt] = ln + 3
            if op.startswith("s_cbranch"):
                blocks[cur].append(("fall", nxt))
            cur = nxt
    succ = collections.defaultdict(list)
    for b in order:
        for kind, x in blocks[b]:
            if kind in ("fall", "jump"):
                succ[b].append(x)
            else:
                m = BR.match("\t" + x)
                if m: succ[b].append(m.group(2))
    pred = collections.defaultdict(list)
    for b, ss in succ.items():
        for s_ in ss: pred[s_].append(b)
    spill_regs = set()
    for b in order:
        for kind, x in blocks[b]:
            if kind == "ins" and x.startswith("v_writelane_b32"):
                spill_regs.add(x.split()[1].rstrip(","))
    WL = re.compile(r"v_writelane_b32 (v\d+), (\S+), (\d+)")
    RL = re.compile(r"v_readlane_b32 (\S+), (v\d+), (\d+)")
    ST = re.compile(r"scratch_store_dword(x\d)? off, (v\[?[\d:]+\]?), off(?: offset:(\d+))?")
    LD = re.compile(r"scratch_load_dword(x\d)? (v\[?[\d:]+\]?), off, off(?: offset:(\d+))?")
    def regs_of(tok, n):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m: return ["v%d" % k for k in range(int(m.group(1)), int(m.group(2)) + 1)]
        return [tok]
    # state: frozenset of facts. facts: ("L", vreg, lane) lane written; ("S", offset) slot written;
    # ("SL", offset, lane): slot holds an image of a lane-spill register with that lane written
    TOP = None
    def transfer(b, state, report):
        st = set(state)
        for kind, x in blocks[b]:
            if kind != "ins": continue
            m = WL.match(x)
            if m:
                st.add(("L", m.group(1), int(m.group(3)))); continue
            m = RL.match(x)
            if m and m.group(2) in spill_regs:
                if report is not None and ("L", m.group(2), int(m.group(3))) not in st:
                    report.append((b, x, "lane read before written on some path"))
                continue
            m = ST.match(x)
            if m and ("Spill" in x or True):
                n = {None: 1, "x2": 2, "x3": 3, "x4": 4}[m.group(1)]
                off = int(m.group(3) or 0)
                for k, r in enumerate(regs_of(m.group(2), n)):
                    o = off + 4 * k
                    st.add(("S", o))
                    for f in [f for f in st if f[0] == "SL" and f[1] == o]: st.discard(f)
                    if r in spill_regs:
                        for f in [f for f in st if f[0] == "L" and f[1] == r]: st.add(("SL", o, f[2]))
                continue
            m = LD.match(x)
            if m:
                n = {None: 1, "x2": 2, "x3": 3, "x4": 4}[m.group(1)]
                off = int(m.group(3) or 0)
                for k, r in enumerate(regs_of(m.group(2), n)):
                    o = off + 4 * k
                    if "Reload" in x and report is not None and ("S", o) not in st:
                        report.append((b, x, "scratch slot reloaded before spilled on some path"))
                    if r in spill_regs:
                        for f in [f for f in st if f[0] == "L" and f[1] == r]: st.discard(f)
                        for f in [f for f in st if f[0] == "SL" and f[1] == o]: st.add(("L", r, f[2]))
                continue
        return frozenset(st)
    IN = {b: TOP for b in order}
    IN["entry"] = frozenset()
    OUT = {b: TOP for b in order}
    work = collections.deque(order)
    inq = set(order)
    while work:
        b = work.popleft(); inq.discard(b)
        if b != "entry":
            ps = [OUT[p] for p in pred[b] if OUT[p] is not TOP]
            if not ps: continue
            s = ps[0]
            for p in ps[1:]: s = s & p
            IN[b] = s
        o = transfer(b, IN[b], None)
        if OUT[b] is TOP or o != OUT[b]:
            OUT[b] = o
            for s_ in succ[b]:
                if s_ not in inq: work.append(s_); inq.add(s_)
    report = []
    n_rl = n_ld = 0
    for b in order:
        if IN[b] is TOP: continue  # unreachable
        transfer(b, IN[b], report)
        for kind, x in blocks[b]:
            if kind == "ins":
                n_rl += bool(RL.match(x)) and RL.match(x).group(2) in spill_regs
                n_ld += ("Reload" in x)
    # one witness path per violation: entry -> ... -> the block, through blocks that never produce the missing fact
    def fact_of(x):
        m = RL.match(x)
        if m: return ("L", m.group(2), int(m.group(3)))
        m = LD.match(x)
        return ("S", int(m.group(3) or 0))
    def gens(b, f):
        return f in transfer(b, frozenset(), None)
    paths = []
    for b, x, why in report:
        f = fact_of(x)
        prev = {"entry": None}; q = collections.deque(["entry"])
        while q:
            c = q.popleft()
            if c == b: break
            if gens(c, f): continue
            for s_ in succ[c]:
                if s_ not in prev: prev[s_] = c; q.append(s_)
        p, c = [], b
        while c is not None and c in prev: p.append("%s@%d" % (c, first_line.get(c, 1))); c = prev[c]
        paths.append(list(reversed(p)))
    # whole-register spills of the lane-spill registers and their EXEC bracket
    wwm = []
    flat = [x for b in order for kind, x in blocks[b] if kind == "ins"]
    for i, x in enumerate(flat):
        m = ST.match(x) or LD.match(x)
        if m and any(r in spill_regs for r in regs_of(m.group(2), 1)):
            prev = " | ".join(flat[max(0, i - 3):i])
            wwm.append((x, "s_or_saveexec_b64" in prev and "-1" in prev))
    # classify: a lane read that feeds only `s_and / s_andn2 / s_or ..., exec` (result not EXEC) is a LANE MASK being merged —
    # the lowering of a divergent i1 phi whose incoming value is undefined on some edge (SILowerI1Copies): only the bits of
    # the lanes active at the merge are replaced, only the bits of active lanes are consumed, so the unwritten bits on the
    # flagged path belong to lanes that are not running. Everything else stays a violation.
    def consumer(b, x):
        m = RL.match(x)
        if not m: return None
        n = int(m.group(1)[1:]) if m.group(1).startswith("s") and m.group(1)[1:].isdigit() else None
        if n is None: return None
        ins = [y for kind, y in blocks[b] if kind == "ins"]
        i = ins.index(x)
        pair = ("s[%d:%d]" % (n, n + 1), "s[%d:%d]" % (n - 1, n))
        for y in ins[i + 1:]:
            toks = re.split(r"[ ,]+", y)
            if any(t in pair or t == "s%d" % n for t in toks[2:]):
                return y
            if toks[1:2] and (toks[1] in pair or toks[1] == "s%d" % n):
                return None  # overwritten before any use: dead read
        return None
    MASKOP = re.compile(r"s_(and|andn2|or)_b64 (s\[\d+:\d+\]), (s\[\d+:\d+\]), exec$")
    kinds = []
    for b, x, why in report:
        c = consumer(b, x)
        kinds.append("lane-mask merge" if c and MASKOP.match(c) else "unclassified")
    return dict(blocks=len(order), spill_regs=sorted(spill_regs), readlanes=n_rl, reloads=n_ld, violations=report, wwm=wwm, paths=paths, kinds=kinds)

def main():
    text = open(sys.argv[1]).read()
    pats = [a for a in sys.argv[2:] if not a.startswith("--")]
    total = 0
    for name, lines in kernels(text).items():
        if pats and not any(p in name for p in pats): continue
        r = analyse(name, lines)
        n_mask = sum(1 for k in r["kinds"] if k == "lane-mask merge")
        total += len(r["violations"]) - n_mask
        print("%s\n  blocks %d, lane-spill VGPRs %s, %d spill readlanes, %d scratch reloads checked: %d not dominated by their spill "
              "(%d of them lane-mask merges, %d unclassified); "
              "%d whole-register spills/reloads of lane-spill VGPRs (%d without an all-lanes EXEC bracket)" % (
                  name, r["blocks"], ",".join(r["spill_regs"]) or "-", r["readlanes"], r["reloads"], len(r["violations"]),
                  n_mask, len(r["violations"]) - n_mask, len(r["wwm"]), sum(1 for _, ok in r["wwm"] if not ok)))
        for ((b, x, why), pth), kind in list(zip(zip(r["violations"], r["paths"]), r["kinds"]))[:60]:
            if kind == "lane-mask merge" and "--all" not in sys.argv: continue
            print("    %s: %s  <- %s [%s]" % (b, x, why, kind))
            if "--paths" in sys.argv: print("        witness (block@line of the kernel body): " + " > ".join(pth[-12:]))
        for x, ok in r["wwm"]:
            if not ok: print("    no all-lanes bracket: %s" % x)
    print("total unclassified: %d" % total)
    return 1 if total else 0

if __name__ == "__main__":
    sys.exit(main())
