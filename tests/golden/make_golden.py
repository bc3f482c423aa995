#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory from the oracle (run from the repo root:
`python tests/golden/make_golden.py`). Data only: seeded inputs are rebuilt by the tests from tests/fixtures.py
and tests/scenes.py; the files hold the expected OUTPUTS.

  traverse_<scene>.npz  256 rays of the scene's standard batch -> t,u,v,normal (float bits), geom/prim ids, front,
                        occluded flags, for [0.001, inf)
  render_<scene>.npz    small forward-mode render -> image float bits + the eight RayStats counters
  seam_calls.npz        192 Material calls per lobe class (scatter_importance, eval, emitted_directional) and 768 Light calls
                        (sample_li, pdf_at_point, escaped) -> the result records' bits (include/crt.h layouts)

The oracle is pinned by the reference's known-answer tests (tests/test_oracle_rt.py, tests/test_oracle_shade.py);
these files freeze its outputs so that a later change to either side shows up as a diff against fixed data."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import ora  # noqa: E402
import scenes  # noqa: E402
import golden_inputs as gi  # noqa: E402


def main():
    for name in scenes.ALL:
        make, extent = scenes.ALL[name]
        s = make(ora)
        rays = gi.traverse_rays(name, extent)
        hf, ids, front = s.intersect_n(rays, 0.001, float("inf"))
        occ = s.occluded_n(rays, 0.001, float("inf"))
        np.savez_compressed(os.path.join(HERE, f"traverse_{name}.npz"), hit_bits=hf.view(np.uint32), ids=ids,
                            front=front.astype(np.uint8), occluded=occ.astype(np.uint8))
    from __graft_entry__ import load_package
    import ora_world
    crt = load_package()
    for name, w, h, spp, depth in gi.RENDERS:
        desc = crt.usda.load(os.path.join(ROOT, "scenes", name + ".usda"), w, h)
        o = ora_world.OracleRenderer(desc, crt.usda, max_depth=depth)
        img, st = o.render(spp, forward=1)
        counters = np.array([getattr(st, f) for f, _ in ora.RayStats._fields_], dtype=np.uint64)
        np.savez_compressed(os.path.join(HERE, f"render_{name}.npz"), image_bits=img.view(np.uint32), counters=counters)
    import seam_cases as sc
    drv = sc.oracle_drivers()
    out = {}
    for cls in sc.CLASSES:
        mats, q = gi.seam_material_case(cls)
        out[cls + "_scatter"] = drv.scatter(mats, q).view(np.uint32)
        out[cls + "_eval"] = drv.eval(mats, q).view(np.uint32)
        out[cls + "_emitted"] = drv.emitted(mats, q).view(np.uint32)
    table, q = gi.seam_light_case()
    out["light_sample"] = drv.light_sample(table, q).view(np.uint32)
    out["light_pdf"] = drv.light_pdf(table, q).view(np.uint32)
    out["light_escaped"] = drv.light_escaped(table, q).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "seam_calls.npz"), **out)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
