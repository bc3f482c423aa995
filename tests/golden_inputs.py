"""Seeded inputs of the golden vectors (tests/golden/): shared by the generator and the tests."""
import numpy as np

import fixtures as fx

N_RAYS = 256
RENDERS = [("cornellbox", 48, 27, 4, 8), ("veach_mis", 48, 27, 4, 8), ("openpbr_showcase", 48, 27, 4, 12),
           ("cornellbox_guided", 32, 32, 4, 8), ("sun_sky", 48, 27, 4, 8),
           ("motionblur", 48, 27, 4, 4), ("rectlight", 48, 27, 4, 4), ("light_visibility", 48, 27, 4, 4),
           ("instancing", 48, 27, 4, 8)]


def traverse_rays(name, extent):
    rays = fx.ray_batch(N_RAYS, extent)
    if name == "mixed":
        k = np.arange(N_RAYS)
        rays[:, 7] = np.array([0xFFFFFFFF, 1, 2, 4], dtype=np.uint32)[k % 4].view(np.float32)
        rays[:, 6] = ((k % 5) / np.float32(5.0)).astype(np.float32)
    return rays


SEAM_CALLS = 192  # calls per lobe class and Material method; 4 x as many Light calls


def seam_material_case(cls):
    """-> (materials, queries) of one lobe class, as numpy records in the C ABI's layouts (tests/seam_cases.py)."""
    import seam_cases as sc
    rng = np.random.default_rng(7000 + sc.CLASSES.index(cls))
    mats = sc.materials(cls, 33, rng)
    return mats, sc.shade_queries(SEAM_CALLS, len(mats), rng)


def seam_light_case():
    import seam_cases as sc
    rng = np.random.default_rng(7100)
    table = sc.lights(rng)
    return table, sc.light_queries(4 * SEAM_CALLS, table, rng)
