"""Parity of the HIP traversal kernels (through the C ABI) with the oracle: bit-exact t/u/v/normal and ids.

Inputs are the reference's own probe scenes and LCG ray batches (ray_throughput.rs:50-116) plus a mixed scene
covering masks, smooth normals, nested/scaled/rotated instances and transform motion blur."""
import numpy as np
import pytest

import fixtures as fx
import ora
import scenes

pytestmark = pytest.mark.gpu

INF = float("inf")


def _batch(name, n, extent):
    rays = fx.ray_batch(n, extent)
    if name == "shards":
        # thin diagonal slivers: aim at points along the diagonal band so hits are common
        g = fx.Lcg(0xBEEF)
        for i in range(n):
            s = g.unit() * np.float32(30.0)
            tgt = np.array([s + g.centered() * 6, s + g.centered() * 6, s + g.centered() * 6], np.float32)
            o = np.array([g.centered() * 80, g.centered() * 80, g.centered() * 80], np.float32)
            d = (tgt - o).astype(np.float32)
            rays[i, 0:3] = o
            rays[i, 3:6] = d / np.sqrt((d * d).sum(dtype=np.float32))
    if name == "mixed":
        # vary the ray category and the shutter time deterministically
        k = np.arange(n)
        masks = np.array([0xFFFFFFFF, 1, 2, 4], dtype=np.uint32)[k % 4]
        rays[:, 7] = masks.view(np.float32)
        rays[:, 6] = ((k % 5) / np.float32(5.0)).astype(np.float32)
    return rays


def _compare(crt, name, t_min, t_max, n=4096):
    import torch
    make, extent = scenes.ALL[name]
    o_scene, p_scene = make(ora), make(crt)
    rays = _batch(name, n, extent)
    hf, ids, front = o_scene.intersect_n(rays, t_min, t_max)
    occ = o_scene.occluded_n(rays, t_min, t_max)
    d_rays = crt.rays_to_device(rays)
    hits = crt.hits_to_host(p_scene.intersect_n(d_rays, t_min, t_max))
    d_occ = p_scene.occluded_n(d_rays, t_min, t_max)
    torch.cuda.synchronize()
    got_occ = d_occ.cpu().numpy()
    assert np.array_equal(hits["geom_id"], ids[:, 0]), name
    assert np.array_equal(hits["prim_id"], ids[:, 1]), name
    hit = ids[:, 0] != 0xFFFFFFFF
    assert hit.sum() > n // 40, f"{name}: only {hit.sum()} hits — the batch is not exercising the kernel"
    for k, f in enumerate(("t",)):
        assert np.array_equal(hits[f][hit].view(np.uint32), hf[hit, 0].view(np.uint32)), name
    assert np.array_equal(hits["normal"][hit].view(np.uint32), hf[hit, 1:4].view(np.uint32)), name
    assert np.array_equal(hits["u"][hit].view(np.uint32), hf[hit, 4].view(np.uint32)), name
    assert np.array_equal(hits["v"][hit].view(np.uint32), hf[hit, 5].view(np.uint32)), name
    assert np.array_equal(hits["front_face"][hit], front[hit].astype(np.uint32)), name
    assert np.array_equal(got_occ.astype(np.uint8), occ), name
    return int(hit.sum())


@pytest.mark.parametrize("name", list(scenes.ALL))
def test_intersect_occluded_match_oracle_bitwise(crt, name):
    _compare(crt, name, 0.001, INF)


@pytest.mark.parametrize("name", ["tri_spheres", "mixed", "sphere_grid"])
@pytest.mark.parametrize("t_max", [0.5, 3.0, 9.0])
def test_bounded_ranges_match_oracle(crt, name, t_max):  # bvh.rs:1583-1607 t_max in {0.5, 3, inf}
    import torch
    make, extent = scenes.ALL[name]
    o_scene, p_scene = make(ora), make(crt)
    rays = _batch(name, 2048, extent)
    hf, ids, front = o_scene.intersect_n(rays, 0.001, t_max)
    occ = o_scene.occluded_n(rays, 0.001, t_max)
    d_rays = crt.rays_to_device(rays)
    hits = crt.hits_to_host(p_scene.intersect_n(d_rays, 0.001, t_max))
    got = p_scene.occluded_n(d_rays, 0.001, t_max).cpu().numpy()
    assert np.array_equal(hits["geom_id"], ids[:, 0]) and np.array_equal(hits["prim_id"], ids[:, 1])
    hit = ids[:, 0] != 0xFFFFFFFF
    assert np.array_equal(hits["t"][hit].view(np.uint32), hf[hit, 0].view(np.uint32))
    assert np.array_equal(got.astype(np.uint8), occ)
    # hit_any == hit.is_some() for every ray and range
    assert np.array_equal(got.astype(bool), hit)


def test_stats_counters_match_oracle(crt):
    """The device counters mirror bvh.rs:53-57; on closest-hit they must equal the oracle's, count for count."""
    import ctypes as C
    make, extent = scenes.ALL["instances"]
    o_scene, p_scene = make(ora), make(crt)
    rays = fx.ray_batch(2048, extent)
    st = ora.TravStats()
    ora.lib().ora_set_trav_stats(C.byref(st))
    o_scene.intersect_n(rays, 0.001, INF)
    ora.lib().ora_set_trav_stats(None)
    ds = crt.CrtTravStats()
    p_scene.intersect_n(crt.rays_to_device(rays), 0.001, INF, stats=ds)
    for f in ("queries", "nodes", "leaves", "packets", "prims"):
        assert list(getattr(ds, f)) == list(getattr(st, f)), f
    assert ds.instance_descents == st.instance_descents and ds.rays == 2048


def test_api_semantics_through_single_ray_entry_points(crt):
    """scene.rs:481-1016 through crt_intersect1 / crt_occluded1 (the drop-in single-ray seam)."""
    R = crt.Ray
    b = crt.SceneBuilder()
    ball = b.attach_sphere((0, 0, 0), 1.0)
    s = b.commit()
    h = s.intersect(R((0, 0, -5), (0, 0, 1)), 0.001, INF)  # lib.rs:9-23
    assert h.geom_id == ball and abs(h.t - 4.0) < 1e-4 and h.front_face == 1
    assert not s.occluded(R((0, 0, -5), (0, 0, 1)), 0.001, 3.9)
    inside = s.intersect(R((0, 0, 0), (0, 0, 1)), 0.001, 100.0)  # scene.rs:590-595
    assert inside.front_face == 0 and np.allclose(list(inside.normal), (0, 0, -1), atol=1e-4)
    # masks (scene.rs:840-855)
    b = crt.SceneBuilder()
    b.attach_sphere((0, 0, 0), 1.0, mask=crt.MASK_SHADOW)
    s = b.commit()
    assert s.intersect(R((0, 0, -5), (0, 0, 1), mask=crt.MASK_CAMERA), 0.001, 100.0) is None
    assert s.intersect(R((0, 0, -5), (0, 0, 1), mask=crt.MASK_SHADOW), 0.001, 100.0) is not None
    assert s.occluded(R((0, 0, -5), (0, 0, 1), mask=crt.MASK_SHADOW), 0.001, 100.0)
    # instance reports the instance's geom_id and the inner prim_id (scene.rs:984-1015)
    inner = crt.SceneBuilder()
    inner.attach_triangles([(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)], [(0, 1, 2), (0, 2, 3)])
    b = crt.SceneBuilder()
    b.attach_sphere((0, -100, 0), 1.0)
    inst = b.attach_instance(inner.commit(), crt.affine(t=(0, 0, 5)))
    h = b.commit().intersect(R((-0.5, 0.5, 0), (0, 0, 1)), 0.001, INF)
    assert h.geom_id == inst and h.prim_id == 1
    # motion blur (scene.rs:958-981)
    unit = crt.SceneBuilder()
    unit.attach_sphere((0, 0, 0), 1.0)
    b = crt.SceneBuilder()
    b.attach_instance(unit.commit(), crt.IDENTITY12, crt.affine(t=(4, 0, 0)))
    s = b.commit()
    assert s.intersect(R((0, 0, -5), (0, 0, 1), time=0.0), 0.001, INF) is not None
    assert s.intersect(R((0, 0, -5), (0, 0, 1), time=1.0), 0.001, INF) is None
    h = s.intersect(R((2, 0, -5), (0, 0, 1), time=0.5), 0.001, INF)
    assert h is not None and abs(h.t - 4.0) < 1e-4
    # empty scene misses (bvh.rs:1663-1668)
    e = crt.SceneBuilder().commit()
    assert e.intersect(R((0, 0, 0), (1, 0, 0)), 0.001, INF) is None and not e.occluded(R((0, 0, 0), (1, 0, 0)), 0.001, INF)


def test_large_batch_properties(crt):
    """Full-size, size-independent checks: occluded == (intersect hit) on 2M rays; hit points lie on the
    unit spheres of tri_spheres' tessellation to within the chord error."""
    import torch
    make, extent = scenes.ALL["tri_spheres"]
    p_scene = make(crt)
    base = fx.ray_batch(4096, extent)
    n = 1 << 21
    g = torch.Generator(device="cpu").manual_seed(7)
    o = (torch.rand(n, 3, generator=g) - 0.5) * (4.0 * extent)
    t = (torch.rand(n, 3, generator=g) - 0.5) * extent
    d = torch.nn.functional.normalize(t - o, dim=1)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o.numpy(); rays[:, 3:6] = d.numpy(); rays[:, 7] = base[0, 7]
    d_rays = crt.rays_to_device(rays)
    hits = crt.hits_to_host(p_scene.intersect_n(d_rays, 0.001, INF))
    occ = p_scene.occluded_n(d_rays, 0.001, INF).cpu().numpy().astype(bool)
    hit = hits["geom_id"] != 0xFFFFFFFF
    assert np.array_equal(occ, hit)
    assert 0.05 < hit.mean() < 0.95
    p = rays[hit, 0:3] + hits["t"][hit, None] * rays[hit, 3:6]
    centers = np.round(p / 2.5) * 2.5
    r = np.linalg.norm(p - centers, axis=1)
    assert np.all(r < 1.0 + 1e-4) and np.all(r > 0.98)
    assert np.all(hits["geom_id"][hit] < 27)


def _build_pile(api, n_spheres=300, n_inst=280):
    """Piles of more than 255 coincident primitives: 300 spheres with one centre and radius, 280 instances of one
    prototype with one placement. No split plane separates them: the builder falls back to median splits by input
    order (bvh.rs:1148-1152), every leaf overlaps every other, and a ray through the pile visits all of them."""
    b = api.SceneBuilder()
    for k in range(n_spheres):
        b.attach_sphere((0.0, 0.0, 0.0), 1.0 + 0.0 * k, mask=1 if k < 200 else 2)  # the first 200 invisible to mask 2
    pile = b.commit()
    pb = api.SceneBuilder()
    pb.attach_sphere((0.0, 0.0, 0.0), 0.5)
    proto = pb.commit()
    b = api.SceneBuilder()
    for k in range(n_inst):
        b.attach_instance(proto, api.affine(None, (4.0, 0.0, 0.0)), mask=4 if k < 270 else 2)
    b.attach_instance(pile, api.affine(None, (0.0, 0.0, 0.0)))
    return b.commit(), (pile, proto)


def test_piles_of_more_than_255_coincident_primitives(crt):
    """Rays whose category only sees the entries beyond the 255th of a pile must still find them, and the deep,
    fully overlapping subtrees must not overflow anything silently: hits, ids and occlusion identical to the oracle,
    and the scene's traversal error word stays clear. (With median splits a single leaf never grows past a few
    entries — its scalar counter has 24 bits regardless, see traverse_pool.hip.h.)"""
    import torch
    o_scene, _k1 = _build_pile(ora)
    p_scene, _k2 = _build_pile(crt)
    assert o_scene.primitive_count() == p_scene.primitive_count() == 281
    rays = fx.ray_batch(2048, 3.0)
    rays[:, 0] += np.float32(2.0)  # between the two piles
    k = np.arange(rays.shape[0])
    rays[:, 7] = np.array([0xFFFFFFFF, 1, 2, 4], dtype=np.uint32)[k % 4].view(np.float32)
    hf, ids, front = o_scene.intersect_n(rays, 0.001, INF)
    occ = o_scene.occluded_n(rays, 0.001, INF)
    d_rays = crt.rays_to_device(rays)
    hits = crt.hits_to_host(p_scene.intersect_n(d_rays, 0.001, INF))
    got_occ = p_scene.occluded_n(d_rays, 0.001, INF).cpu().numpy()
    torch.cuda.synchronize()
    p_scene.traversal_error()  # no stack overflow on this scene
    hit = ids[:, 0] != 0xFFFFFFFF
    m2 = (k % 4 == 2) & hit
    assert m2.sum() > 20, "rays of category 2 must reach the entries past the 255th"
    assert set(np.unique(ids[m2, 0])) <= {271, 272, 273, 274, 275, 276, 277, 278, 279, 280}
    assert np.array_equal(hits["geom_id"], ids[:, 0]) and np.array_equal(hits["prim_id"], ids[:, 1])
    assert np.array_equal(hits["t"][hit].view(np.uint32), hf[hit, 0].view(np.uint32))
    assert np.array_equal(hits["normal"][hit].view(np.uint32), hf[hit, 1:4].view(np.uint32))
    assert np.array_equal(got_occ.astype(np.uint8), occ)


def test_single_ray_queries_from_concurrent_threads(crt):
    """Scene::intersect / occluded are `&self` and thread-safe (scene.rs:344), called from every worker of the host
    integrator (tracer.rs:428): eight threads, two scenes, each thread's answers equal the oracle's."""
    import threading
    cases = []
    for name in ("mixed", "sphere_grid"):
        make, extent = scenes.ALL[name]
        rays = _batch(name, 96, extent)
        o_scene, p_scene = make(ora), make(crt)
        hf, ids, front = o_scene.intersect_n(rays, 0.001, INF)
        occ = o_scene.occluded_n(rays, 0.001, INF)
        cases.append((p_scene, rays, hf, ids, occ))
    errors = []

    def work(tid):
        try:
            p_scene, rays, hf, ids, occ = cases[tid % 2]
            for i in range(tid % 3, rays.shape[0], 3):
                r = crt.Ray(rays[i, 0:3], rays[i, 3:6], float(rays[i, 6]), int(rays[i, 7:8].view(np.uint32)[0]))
                h = p_scene.intersect(r, 0.001, INF)
                if ids[i, 0] == 0xFFFFFFFF:
                    assert h is None, (tid, i)
                else:
                    assert h is not None and (h.geom_id, h.prim_id) == (ids[i, 0], ids[i, 1]), (tid, i)
                    assert np.float32(h.t).view(np.uint32) == hf[i, 0].view(np.uint32), (tid, i)
                assert bool(p_scene.occluded(r, 0.001, INF)) == bool(occ[i]), (tid, i)
            crt.lib().crt_thread_release()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


@pytest.mark.parametrize("n", [1, 63, 255, 256, 257, 1000, 70001])
def test_ragged_batch_sizes_are_dealt_completely(crt, n):
    """The batched queries deal their rays to the workgroups in round-robin 256-ray blocks (traverse.hip, dealt_ray):
    every ray of a batch of any size — smaller than a block, a ragged last block, fewer blocks than workgroups — is
    traced exactly once and lands in its own output slot."""
    import torch
    make, extent = scenes.ALL["mixed"]
    o_scene, p_scene = make(ora), make(crt)
    rays = fx.ray_batch(n, extent)
    hf, ids, _front = o_scene.intersect_n(rays, 0.001, INF)
    occ = o_scene.occluded_n(rays, 0.001, INF)
    d_rays = crt.rays_to_device(rays)
    d_hits = torch.full(((n + 3) * 40,), 0xAB, dtype=torch.uint8, device="cuda")  # guard records behind the batch
    p_scene.intersect_n(d_rays, 0.001, INF, d_hits)
    d_occ = torch.full((n + 3,), 7, dtype=torch.int32, device="cuda")
    p_scene.occluded_n(d_rays, 0.001, INF, d_occ)
    torch.cuda.synchronize()
    hits = crt.hits_to_host(d_hits[: n * 40])
    hit = ids[:, 0] != 0xFFFFFFFF
    assert np.array_equal(hits["geom_id"], ids[:, 0]) and np.array_equal(hits["prim_id"], ids[:, 1])
    assert np.array_equal(hits["t"][hit].view(np.uint32), hf[hit, 0].view(np.uint32))
    assert np.array_equal(d_occ[:n].cpu().numpy().astype(np.uint8), occ)
    assert bool((d_hits[n * 40:] == 0xAB).all()) and bool((d_occ[n:] == 7).all()), "a write behind the batch"
