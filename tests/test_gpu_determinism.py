"""Run-to-run determinism of the wavefront path tracer: two renderers of the same scene in one process must produce
the same bits and the same counters, whatever order the queues happened to be filled in. (Per-path arithmetic is
order-free and every output slot is private to its path; a difference here means a race or an uninitialised read —
a build of round 2 showed exactly that on `sun_sky` in the fused kernel, a few ulps in most pixels, and only this
kind of test sees it when the oracle comparison runs once.)"""
import os

import numpy as np
import pytest

import ora

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,w,h,spp,depth", [("sun_sky", 192, 108, 8, 8), ("veach_mis", 192, 108, 8, 8),
                                                ("openpbr_showcase", 192, 108, 8, 12), ("cornellbox_guided", 96, 96, 8, 8),
                                                ("PointInstancedMedCity", 192, 108, 4, 6)])
def test_two_renders_are_bit_identical(crt, name, w, h, spp, depth):
    import torch
    path = os.path.join(ROOT, "scenes", name + ".usda")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "scenes", name + ".usd")
    imgs, stats = [], []
    for _ in range(3):
        r, _desc = crt.load_usda(path, w, h, depth)
        r.render_samples(0, spp)
        torch.cuda.synchronize()
        imgs.append(r.image())
        stats.append(r.stats())
        del r
    for k in (1, 2):
        assert np.array_equal(imgs[0].view(np.uint32), imgs[k].view(np.uint32)), (name, k, int((imgs[0] != imgs[k]).any(axis=-1).sum()))
        for f, _t in ora.RayStats._fields_:
            assert getattr(stats[0], f) == getattr(stats[k], f), (name, f)
