#!/usr/bin/env python3
"""bench_published.py — the end-to-end renders the reference publishes (docs/simd.md:222-223), at the
scenes' own authored settings: 640x360, 128 spp with adaptive stopping (variance 0.05, min 32 spp), depth 32.

Published: wall clock, min of 3, all cores of a 4-core Xeon E5-2699 v4. Here: render-phase wall clock (scene already
built and resident, film read back at the end), min of 3, one MI355X. Sample values differ from the reference's
(sampler parity unpinned), so the per-pixel stopping decisions — and the total sample count — are this
build's own; the oracle takes exactly the same ones (tests/test_gpu_render.py::test_adaptive_stopping_identical_to_oracle)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PUBLISHED_S = {"cornellbox": 8.38, "openpbr_showcase": 3.63}


def main():
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    crt = load_package()
    out = {"device": torch.cuda.get_device_name(0), "renders": {}}
    for name, published in PUBLISHED_S.items():
        path = os.path.join(ROOT, "scenes", name + ".usda")
        desc = crt.usda.load(path)
        s = desc.settings
        row = {"settings": {k: s[k] for k in ("width", "height", "spp", "min_spp", "variance", "max_depth")},
               "published_cpu_seconds": published}
        for batch in (4, 16):
            best = None
            for _ in range(3):
                r, _d = crt.load_usda(path, variance=s["variance"], min_spp=s["min_spp"])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r.render_adaptive(s["spp"], batch=batch)
                img = r.image()
                dt = time.perf_counter() - t0
                st = r.stats()
                counts = r.sample_counts()
                if best is None or dt < best["seconds"]:
                    best = {"seconds": round(dt, 4), "rays": st.total_rays(), "mray_s": round(st.total_rays() / dt / 1e6, 1),
                            "mean_spp": round(float(counts.mean()), 2), "pixels_at_full_spp": int((counts >= s["spp"]).sum()),
                            "finite": bool(np.isfinite(img).all())}
            row["batch_%d" % batch] = best
        row["speedup_vs_published"] = round(published / row["batch_16"]["seconds"], 1)
        out["renders"][name] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
