"""Pixel-tile sharding across ranks and the one collective of the multi-GPU path.

Every pixel is independent (per-pixel sampler keys, per-pixel film estimator: tracer.rs:543, :559-560, :599-634),
and the reference already renders 16x16 tiles independently (tracer.rs:424-459). So the frame shards by tile with
the scene replicated, ranks never exchange anything while tracing, and the only collective is a gather of the
finished tile buffers: equal-size padded buffers, gathered to rank 0 (bench.py) or to every rank (all_gather) —
RCCL over xGMI with backend "nccl"; gloo in the CPU tests. Ownership is disjoint, so the assembled frame equals the single-rank frame bit for bit.
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
    """Common per-rank buffer length for the gather: the largest shard, rounded up to a 256-pixel multiple
    (C ABI: crt_shard_padded_count)."""
    from . import lib
    return int(lib().crt_shard_padded_count(width, height, world))


class GatherPlan:
    """Everything of the tile gather that does not depend on the pixels: per-rank index tensors and the send /
    receive buffers, built once (outside any timed region)."""

    def __init__(self, width, height, world, device):
        import torch
        self.width, self.height, self.world = width, height, world
        self.n_max = padded_count(width, height, world)
        self.idx = [torch.from_numpy(shard_pixels(width, height, r, world).astype(np.int64)).to(device) for r in range(world)]
        self.send = torch.zeros(self.n_max * 3, dtype=torch.float32, device=device)
        self.recv = torch.empty(world * self.n_max * 3, dtype=torch.float32, device=device) if world > 1 else self.send
        self.frame = torch.zeros(height * width, 3, dtype=torch.float32, device=device)
        # buffers in HBM: the frame is assembled by the library's own kernel (crt_gather_plan_assemble) — what a host
        # without torch calls after its own collective; CPU tensors (the gloo tests) take the indexed stores below
        self._plan = None
        if torch.device(device).type == "cuda":
            from . import lib
            self._plan = lib().crt_gather_plan_new(width, height, world)
            if not self._plan:
                raise RuntimeError("crt_gather_plan_new failed: " + lib().crt_last_error().decode())
            assert lib().crt_gather_plan_padded_count(self._plan) == self.n_max

    def __del__(self):
        try:
            if self._plan:
                from . import lib
                lib().crt_gather_plan_free(self._plan)
        except Exception:
            pass

    def gather(self, film_local, dist=None, dst=None):
        """film_local: [n_owned, 3] of this rank (same device as the plan). ONE collective, then one indexed store
        per rank's shard. dst=None: all_gather_into_tensor, every rank gets the frame. dst=r: a gather to rank r
        only (1/N of the bytes; RCCL runs it as N-1 point-to-point transfers over xGMI) — the other ranks return
        None. Returns the full frame [height*width, 3] (buffer order: row 0 = bottom)."""
        self.send[: film_local.numel()] = film_local.reshape(-1)
        if self.world > 1 and dist is not None:
            if dst is None:
                dist.all_gather_into_tensor(self.recv, self.send)
            else:
                mine = dist.get_rank() == dst
                parts = list(self.recv.reshape(self.world, -1).unbind(0)) if mine else None
                dist.gather(self.send, parts, dst=dst)
                if not mine:
                    return None
        if self._plan:
            from . import lib, _check, _stream_ptr
            _check(lib().crt_gather_plan_assemble(self._plan, self.recv.data_ptr(), self.frame.data_ptr(), _stream_ptr(None)),
                   "crt_gather_plan_assemble")
            return self.frame
        recv = self.recv.reshape(self.world, self.n_max, 3)
        for r in range(self.world):
            self.frame[self.idx[r]] = recv[r, : self.idx[r].numel()]
        return self.frame


def gather_frame(film_local, width, height, rank, world, dist=None):
    """One-shot form of GatherPlan(...).gather(...): film_local [n_owned, 3] in shard_pixels order -> full frame on
    every rank. With world == 1 (or dist None) it is a local scatter."""
    return GatherPlan(width, height, world, film_local.device).gather(film_local, dist)


def import_once(path, width=None, height=None, dist=None, src=0):
    """The scene description of a multi-rank job, imported ONCE: rank `src` reads and composes the USD file
    (usda.load: seconds for PointInstancedMedCity's crate, 14 s for the packed stress scene's 15.6 MB of text), every
    other rank receives the result in one object broadcast instead of parsing the file again on its own cores.
    The commit that follows (usda.build_world -> crt_commit) is deterministic (bvh.rs:22-24), so all ranks hold the same
    tree. Without a process group (dist None or world 1) this is usda.load."""
    from . import usda
    if dist is None or dist.get_world_size() == 1:
        return usda.load(path, width, height)
    box = [usda.load(path, width, height) if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


RAY_STATS_FIELDS = 8


def reduce_ray_stats(stats, dist=None, device="cpu"):
    """Sum of the ranks' RayStats (stats.rs:128-147: eight 64-bit counters, declaration order) — one all_reduce.
    `stats`: a RayStats ctypes struct (or any object with its _fields_). Returns the eight totals as a list of int."""
    import torch
    vals = [int(getattr(stats, f)) for f, _t in stats._fields_][:RAY_STATS_FIELDS]
    t = torch.tensor(vals, dtype=torch.int64, device=device)
    if dist is not None and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]
