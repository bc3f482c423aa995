#!/usr/bin/env python3
"""Regenerates scenes/stress.usda.xz: the reference's BVH stress scene (36 placements of one 9 800-triangle UV sphere,
a 30 000-triangle field of thin shards, floor, occluder slab, one RectLight; 640x360, 16 spp, depth 6, variance 0 —
docs/simd.md:221 publishes its render time), as written by the reference's own generator and packed with xz.

    python scenes/make_stress.py            (needs /root/reference; the generator is stdlib Python with a fixed seed)

The generator script itself is the reference's and does not travel; its OUTPUT is data — 15.6 MB of USDA text, 0.7 MB
packed — and is committed like the sample scenes. crust-render_amd/usda.py reads the .xz directly."""
import lzma
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = "/root/reference/scripts/gen_stress_scene.py"


def main():
    if not os.path.exists(GEN):
        raise SystemExit("%s is not here: scenes/stress.usda.xz can only be regenerated where the reference is" % GEN)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "stress.usda")
        print(subprocess.run([sys.executable, GEN, out], check=True, capture_output=True, text=True).stdout.strip())
        with open(out, "rb") as f:
            raw = f.read()
    packed = lzma.compress(raw, preset=9)
    with open(os.path.join(HERE, "stress.usda.xz"), "wb") as f:
        f.write(packed)
    print("scenes/stress.usda.xz: %d -> %d bytes" % (len(raw), len(packed)))


if __name__ == "__main__":
    main()
