#!/usr/bin/env python3
"""Per-kernel resource usage and static instruction mix from the gfx950 ISA (no GPU needed: hipcc cross-compiles).
    python profiles/isa_resources.py > profiles/r02_isa_resources.txt
Compiles kernels/pathtrace.hip and kernels/traverse.hip to assembly with the Makefile's flags and reads the
`amdhsa.kernels` metadata (VGPRs, spilled VGPRs, SGPRs spilled to VGPR lanes, scratch and LDS bytes) and counts the
instruction classes of each kernel's body."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crust-render_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I../../include",
         "-fno-slp-vectorize", "-O2", "-mllvm", "-amdgpu-set-wave-priority", "--cuda-device-only", "-S"]
print("%-64s %5s %6s %6s %7s %6s | %6s %6s %6s %7s %5s %5s %7s" % ("kernel", "vgpr", "vspill", "sspill", "scratch", "lds", "instr", "valu", "v_mov", "cndmask", "vmem", "lds", "lane_rw"))
for src in ("kernels/pathtrace.hip", "kernels/traverse.hip"):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", tmp.name], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        s = open(tmp.name).read()
    meta = {}
    for e in s[s.index("amdhsa.kernels:"):].split("  - .agpr_count")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, e) or [None, "?"])[1]
        meta[g("name")] = [g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")]
    for f in re.split(r"\n(?=_Z[0-9A-Za-z_]+:)", s):
        name = f.split(":", 1)[0]
        if name not in meta:
            continue
        c = collections.Counter()
        for l in f.split("\n"):
            if not l.startswith("\t") or l.strip().startswith((".", ";")):
                continue
            op = l.split()[0]
            c["instr"] += 1
            if op.startswith("v_"):
                c["valu"] += 1
                c["v_mov"] += op.startswith("v_mov")
                c["cndmask"] += op.startswith("v_cndmask")
                c["lane_rw"] += op.startswith(("v_readlane", "v_writelane"))
            elif op.startswith(("global_", "flat_", "scratch_", "buffer_")):
                c["vmem"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem)
        dem = re.sub(r"\(.*", "", dem).replace("void crt::", "")
        m = meta[name]
        print("%-64s %5s %6s %6s %7s %6s | %6d %6d %6d %7d %5d %5d %7d" % (dem[:64], m[0], m[1], m[2], m[3], m[4], c["instr"], c["valu"], c["v_mov"], c["cndmask"], c["vmem"], c["lds"], c["lane_rw"]))
