"""Pixel-tile sharding across ranks and the one collective of the multi-GPU path.

Every pixel is independent (per-pixel sampler keys, per-pixel film estimator: tracer.rs:543, :559-560, :599-634),
and the reference already renders 16x16 tiles independently (tracer.rs:424-459). So the frame shards by tile with
the scene replicated, ranks never exchange anything while tracing, and the only collective is a gather of the
finished tile buffers: one all_gather of equal-size padded buffers (RCCL over xGMI with backend "nccl"; gloo in
the CPU tests). Ownership is disjoint, so the assembled frame equals the single-rank frame bit for bit.
"""
import ctypes as C

import numpy as np


def shard_pixels(width, height, rank, world):
    """Linear buffer indices (j*width+i) of the pixels `rank` owns, in trace order (C ABI: crt_shard_pixels)."""
    from . import lib
    n = lib().crt_shard_pixels(width, height, rank, world, None)
    out = np.zeros(n, dtype=np.uint32)
    if n:
        lib().crt_shard_pixels(width, height, rank, world, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def padded_count(width, height, world):
    """Common per-rank buffer length for the gather: the largest shard, rounded up to a 256-pixel multiple."""
    tx, ty = (width + 15) // 16, (height + 15) // 16
    tiles = (tx * ty + world - 1) // world
    return ((tiles * 256 + 255) // 256) * 256


def gather_frame(film_local, width, height, rank, world, dist=None):
    """film_local: torch tensor [n_owned, 3] (any device) of this rank's pixel means in shard_pixels order.
    Returns the full frame [height*width, 3] on every rank (buffer order: row 0 = bottom). With world == 1 (or
    dist None) it is a local scatter; otherwise ONE all_gather_into_tensor of padded per-rank buffers."""
    import torch
    n_max = padded_count(width, height, world)
    send = torch.zeros(n_max * 3, dtype=torch.float32, device=film_local.device)
    send[: film_local.numel()] = film_local.reshape(-1)
    if world > 1 and dist is not None:
        recv = torch.empty(world * n_max * 3, dtype=torch.float32, device=film_local.device)
        dist.all_gather_into_tensor(recv, send)
    else:
        recv = send
    frame = torch.zeros(height * width, 3, dtype=torch.float32, device=film_local.device)
    recv = recv.reshape(world, n_max, 3)
    for r in range(world):
        idx = torch.from_numpy(shard_pixels(width, height, r, world).astype(np.int64)).to(film_local.device)
        frame[idx] = recv[r, : idx.numel()]
    return frame
