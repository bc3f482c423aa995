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
