"""Multi-rank path on CPU: pixel-tile sharding + the tile gather, world_size 2 over gloo (no GPU needed)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_shards_partition_the_frame(crt):
    for (w, h) in [(80, 48), (1920, 1080), (33, 17), (16, 16)]:
        for world in (1, 2, 3, 8):
            seen = np.zeros(w * h, dtype=np.int32)
            for rank in range(world):
                idx = crt.shard.shard_pixels(w, h, rank, world)
                assert idx.size <= crt.shard.padded_count(w, h, world)
                seen[idx] += 1
            assert (seen == 1).all(), (w, h, world)
    # tiles are 16x16 and dealt round-robin (tracer.rs:424, :1671-1686): tile t -> rank t % world
    idx = crt.shard.shard_pixels(64, 32, 1, 2)
    assert idx[0] == 16 and idx[15] == 31 and idx[16] == 64 + 16


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    crt = load_package()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    w, h = 80, 48
    idx = crt.shard.shard_pixels(w, h, rank, world)
    # stand-in for the rank's film: a deterministic function of the pixel index
    film = torch.from_numpy(np.stack([idx * 0.5, idx * 2.0 + 1.0, -idx.astype(np.float64)], axis=1).astype(np.float32))
    frame = crt.shard.gather_frame(film, w, h, rank, world, dist)
    np.save(os.path.join(out_dir, f"frame{rank}.npy"), frame.numpy())
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    all_idx = np.arange(80 * 48)
    want = np.stack([all_idx * 0.5, all_idx * 2.0 + 1.0, -all_idx.astype(np.float64)], axis=1).astype(np.float32)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"frame{r}.npy"))
        assert np.array_equal(got, want)


def _worker_gather_to_root(rank, world, port, out_dir, w, h):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    crt = load_package()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    idx = crt.shard.shard_pixels(w, h, rank, world)
    film = torch.from_numpy(np.stack([idx * 0.25, idx + 0.5, idx * -2.0], axis=1).astype(np.float32))
    plan = crt.shard.GatherPlan(w, h, world, "cpu")
    frame = plan.gather(film, dist, dst=0)  # bench.py's collective: a gather to rank 0 only
    assert (frame is None) == (rank != 0)
    if rank == 0:
        np.save(os.path.join(out_dir, "frame_root.npy"), frame.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(4, 1920, 1080), (8, 200, 120)])
def test_gather_to_rank0_reassembles_uneven_shards_gloo(tmp_path, world, w, h):
    """The driver's multi-GPU run in miniature: `world` ranks own round-robin 16x16 tiles of a frame whose tile count
    does not divide evenly (1920x1080: 120 x 68 = 8160 tiles with a 1080 = 67.5 x 16 ragged last row), pad to the
    largest shard, ONE gather to rank 0, which scatters every shard back to its pixels."""
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 2000) + world
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_gather_to_root, args=(r, world, port, str(tmp_path), w, h)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    i = np.arange(w * h)
    want = np.stack([i * 0.25, i + 0.5, i * -2.0], axis=1).astype(np.float32)
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "frame_root.npy")), want)


def _worker_import_once(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import pickle
    import torch.distributed as dist
    from __graft_entry__ import load_package
    crt = load_package()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if rank != 0:  # only rank 0 may touch the file
        def refuse(*a, **k):
            raise AssertionError("rank %d parsed the USD file itself" % rank)
        crt.usda.load = refuse
    desc = crt.shard.import_once(crt.scene_path("cornellbox_guided"), 96, 54, dist)
    # the commit is each rank's own and deterministic: same tree everywhere (primitive counts and the image's statistics)
    scene, mats, _protos = crt.usda.build_world(desc, crt, crt.default_material)
    summary = dict(geoms=len(desc.geoms), lights=len(desc.lights), settings=desc.settings, mats=len(mats),
                   prims=scene.primitive_count(), image=scene.image_check(),
                   verts=[g["verts"].tobytes() for g in desc.geoms if g["kind"] == "mesh"])
    with open(os.path.join(out_dir, "desc%d.pkl" % rank), "wb") as f:
        pickle.dump(summary, f)
    # the job's RayStats: eight 64-bit counters summed over the ranks in one all_reduce (stats.rs:128-147)
    st = crt.CrtRayStats()
    for k, (name, _t) in enumerate(st._fields_):
        setattr(st, name, (rank + 1) * 10 ** k + (1 << 40) * (k == 1))
    total = crt.shard.reduce_ray_stats(st, dist)
    with open(os.path.join(out_dir, "stats%d.pkl" % rank), "wb") as f:
        pickle.dump(total, f)
    dist.destroy_process_group()


def test_scene_is_imported_once_and_ray_stats_are_reduced_gloo(tmp_path):
    """N > 1: rank 0 imports the USD file and broadcasts the description (the other rank's parser is disabled), both
    ranks commit the same tree; the eight RayStats counters add up over the ranks (64-bit: closest_hit alone passes 2^32
    within one bench step)."""
    import pickle
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_import_once, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    d = [pickle.load(open(os.path.join(str(tmp_path), "desc%d.pkl" % r), "rb")) for r in range(2)]
    assert d[0] == d[1] and d[0]["geoms"] > 0 and d[0]["lights"] > 0
    s = [pickle.load(open(os.path.join(str(tmp_path), "stats%d.pkl" % r), "rb")) for r in range(2)]
    assert s[0] == s[1] == [3 * 10 ** k + (2 << 40) * (k == 1) for k in range(8)]


def _worker_fit_batch(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    class Err(RuntimeError):
        pass

    class FakeCrt:
        CrtError = Err

    class FakeRenderer:  # allocates up to `limit` samples per batch: the ranks of a node need not have the same free HBM
        def __init__(self, limit):
            self.limit, self.calls, self.cleared = limit, [], 0

        def render_samples(self, begin, count, stream):
            self.calls.append(count)
            if count > self.limit:
                raise Err("out of memory")

        def clear(self, stream):
            self.cleared += 1

    r = FakeRenderer(8 if rank == 0 else 64)  # per job sample counts (spp_per_step x world): rank 0 fits 8, rank 1 fits 64
    got = bench.fit_batch(FakeCrt, r, 32, world, None, dist, "cpu")
    with open(os.path.join(out_dir, "fit%d.txt" % rank), "w") as f:
        f.write("%d %s %d" % (got, ",".join(map(str, r.calls)), r.cleared))
    dist.destroy_process_group()


def test_ranks_agree_on_the_smallest_batch_any_of_them_fits(tmp_path):
    """bench.py's halving fallback with N > 1: every rank halves on its own until ITS batch fits, then one
    all-reduce(MIN) makes all of them run the smallest shape — rank 0 (32 -> 16 -> 8 -> 4 spp per share at world 2) and
    rank 1 (32 at once) both come out at 4 — and the probe's samples are cleared on every rank."""
    import torch.multiprocessing as mp
    port = 29500 + ((os.getpid() + 911) % 2000)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_fit_batch, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    a = open(os.path.join(str(tmp_path), "fit0.txt")).read().split()
    b = open(os.path.join(str(tmp_path), "fit1.txt")).read().split()
    assert a[0] == b[0] == "4" and a[1] == "64,32,16,8" and b[1] == "64" and a[2] == b[2] == "1"
