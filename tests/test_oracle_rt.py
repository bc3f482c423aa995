"""Pins the ORACLE's kernel layer against the reference's own known-answer tests.

Every test here is a port of a #[test] in /root/reference/crates/crust-rt/src/{triangle,bvh,scene}.rs
(cited per test); together they are the golden vectors SURVEY §8(c) lists for the kernel.
"""
import ctypes as C
import math

import numpy as np
import pytest

import fixtures as fx
import ora
from ora import INF, MASK_ALL, MASK_CAMERA, MASK_SHADOW, ray

f32 = np.float32


def tri_hit(r, v0, v1, v2, t_min=0.001, t_max=INF):
    out = np.zeros(3, dtype=np.float32)
    a, b, c = (np.asarray(v, dtype=np.float32) for v in (v0, v1, v2))
    ok = ora.lib().ora_triangle_intersect(C.byref(r), ora._fp(a), ora._fp(b), ora._fp(c), t_min, t_max, ora._fp(out))
    return tuple(out) if ok else None


def tri4(r, tris, masks, ray_mask=MASK_ALL, t_min=0.001, t_max=INF):
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    masks = np.ascontiguousarray(masks, dtype=np.uint32)
    hits, fb, act = C.c_uint32(), C.c_uint32(), C.c_uint32()
    t = np.zeros(4, np.float32); u = np.zeros(4, np.float32); v = np.zeros(4, np.float32)
    ora.lib().ora_tri4_intersect(C.byref(r), ora._fp(tris), ora._up(masks), tris.shape[0], ray_mask, t_min, t_max,
                                 C.byref(hits), C.byref(fb), C.byref(act), ora._fp(t), ora._fp(u), ora._fp(v))
    return hits.value, fb.value, act.value, t, u, v


# ---------------------------------------------------------------- triangle.rs
def test_interior_hit_has_expected_t_and_barycentrics():  # triangle.rs:440-453
    r = ray((2, 1, 0), (0, 0, 1))
    t, u, v = tri_hit(r, (0, 0, 5), (4, 0, 5), (0, 4, 5))
    assert abs(t - 5.0) < 1e-5 and abs(u - 0.5) < 1e-5 and abs(v - 0.25) < 1e-5


def test_respects_t_range():  # triangle.rs:456-469
    tri = ((-1, -1, 5), (1, -1, 5), (0, 1, 5))
    r = ray((0, 0, 0), (0, 0, 1))
    assert tri_hit(r, *tri, 0.001, 4.9) is None
    assert tri_hit(r, *tri, 5.1, 100.0) is None
    assert tri_hit(r, *tri, 0.001, INF) is not None
    assert tri_hit(ray((0, 0, 0), (0, 0, -1)), *tri) is None


def test_shared_edge_is_watertight():  # triangle.rs:476-497
    p00 = np.array([-1.3371, -0.7713, 3.7], np.float32)
    p10 = np.array([1.9241, -1.1157, 4.3], np.float32)
    p11 = np.array([1.6083, 1.4127, 3.9], np.float32)
    p01 = np.array([-0.9743, 0.8291, 4.1], np.float32)
    tris = [(p00, p10, p11), (p00, p11, p01)]
    origin = np.array([0.1731, -0.0913, 0.0], np.float32)
    for i in range(0, 10001):
        s = f32(i) / f32(10000.0)
        target = p00 + s * (p11 - p00)
        r = ray(origin, (target - origin).astype(np.float32))
        hits = sum(1 for t in tris if tri_hit(r, *t) is not None)
        assert hits >= 1, f"pinhole at s={s}"


def test_shared_vertex_is_covered():  # triangle.rs:501-518
    hub = (0.2137, 0.5391, 5.1)
    rim = [(1.3, 0.4, 5.0), (0.3, 1.7, 5.3), (-1.1, 0.6, 4.9), (-0.2, -1.2, 5.2)]
    r = ray((0, 0, 0), hub)
    assert sum(1 for k in range(4) if tri_hit(r, hub, rim[k], rim[(k + 1) % 4]) is not None) >= 1


def _norm3(d):
    ln = np.sqrt(f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
    return (d / ln).astype(np.float32)


def test_simd_matches_scalar_bitwise():  # triangle.rs:525-594
    g = fx.Lcg(0x12345678)
    nx = g.centered
    compared = hits = 0
    for _ in range(2000):
        tris = []
        for i in range(4):
            base = np.array([nx(), nx(), nx()], np.float32) * f32(4.0)
            tris.append([base + np.array([nx(), nx(), nx()], np.float32) for _ in range(3)])
        origin = np.array([nx(), nx(), nx()], np.float32) * f32(6.0)
        aimed = nx() > 0.0
        if aimed:
            pick = int(f32(nx() + f32(0.5)) * f32(4.0)) % 4
            v0, v1, v2 = tris[pick]
            a, b = f32(nx() + f32(0.5)), f32(nx() + f32(0.5))
            if a + b > 1.0:
                a, b = f32(1.0) - a, f32(1.0) - b
            d = ((v0 + (v1 - v0) * a + (v2 - v0) * b) - origin).astype(np.float32)
        else:
            d = np.array([nx(), nx(), nx()], np.float32)
        if f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) < 1e-8:
            continue
        r = ray(origin, _norm3(d))
        flat = np.array([np.concatenate(t) for t in tris], np.float32)
        h, fb, act, t4, u4, v4 = tri4(r, flat, [MASK_ALL] * 4)
        for lane in range(4):
            sc = tri_hit(r, *tris[lane])
            if fb & (1 << lane):
                continue
            compared += 1
            simd = bool(h & (1 << lane))
            assert simd == (sc is not None)
            if simd:
                hits += 1
                assert np.float32(t4[lane]).view(np.uint32) == np.float32(sc[0]).view(np.uint32)
                assert np.float32(u4[lane]).view(np.uint32) == np.float32(sc[1]).view(np.uint32)
                assert np.float32(v4[lane]).view(np.uint32) == np.float32(sc[2]).view(np.uint32)
    assert compared > 7000 and hits > 50


def test_simd_respects_t_range_like_scalar():  # triangle.rs:599-626
    a, b, c = (-1, -1, 5), (1, -1, 5), (0, 1, 5)
    cases = [(a, b, c), (a, c, b)]
    flat = np.array([np.concatenate([np.array(v, np.float32) for v in t]) for t in cases], np.float32)
    for (t_min, t_max) in [(0.001, 4.9), (5.1, 100.0), (0.001, INF), (4.9, 5.1)]:
        for d in [(0, 0, 1), (0, 0, -1)]:
            o = (0, 0, 0.0 if d[2] > 0 else 10.0)
            r = ray(o, d)
            h, *_ = tri4(r, flat, [MASK_ALL] * 2, MASK_ALL, t_min, t_max)
            for lane, t in enumerate(cases):
                assert bool(h & (1 << lane)) == (tri_hit(r, *t, t_min, t_max) is not None)


def test_simd_inactive_lanes_never_hit():  # triangle.rs:630-644
    flat = np.array([[-1, -1, 5, 1, -1, 5, 0, 1, 5]], np.float32)
    h, fb, act, *_ = tri4(ray((0, 0, 0), (0, 0, 1)), flat, [MASK_ALL])
    assert act == 0b0001 and h == 0b0001 and (fb & ~1) == 0


def test_simd_masks_gate_lanes():  # triangle.rs:648-681
    def tri(z):
        return [-1, -1, z, 1, -1, z, 0, 1, z]
    flat = np.array([tri(5.0), tri(6.0), tri(7.0)], np.float32)
    masks = [MASK_CAMERA, MASK_SHADOW, MASK_ALL]
    r = ray((0, 0, 0), (0, 0, 1))
    assert tri4(r, flat, masks, MASK_CAMERA)[0] == 0b101
    assert tri4(r, flat, masks, MASK_SHADOW)[0] == 0b110
    assert tri4(r, flat, masks, MASK_ALL)[0] == 0b111
    assert tri4(r, flat, masks, 1 << 20)[0] == 0b100


def test_axis_aligned_rays_hit():  # triangle.rs:686-698
    tri = ((-1, -1, 2), (1, -1, 2), (0, 1, 2))
    for d in [(0, 0, 1), (0, 0, -1)]:
        o = (0, 0, 0.0 if d[2] > 0 else 4.0)
        hit = tri_hit(ray(o, d), *tri)
        assert hit is not None and abs(hit[0] - 2.0) < 1e-5


# ---------------------------------------------------------------- bvh.rs
def sphere_grid_scene(n):
    b = ora.SceneBuilder()
    for c in fx.sphere_grid_centers(n):
        b.attach_sphere(c, 0.5)
    return b.commit()


def shards_scene(n):
    b = ora.SceneBuilder()
    v, i = fx.diagonal_shards(n)
    b.attach_triangles(v, i)
    return b.commit()


ORIGINS = [(-5, 4.5, 4.5), (20, 3, 3), (4.5, -5, 4.5), (0, 0, -10), (5, 5.2, -3)]


def _dirs():
    def n(v):
        return _norm3(np.array(v, np.float32))
    return [np.array([1, 0, 0], np.float32), n([-1, 0.05, 0.02]), np.array([0, 1, 0], np.float32), n([0.3, 0.3, 1.0]),
            np.array([0, 0, -1], np.float32), np.array([0.577, 0.577, 0.577], np.float32)]


def assert_matches_linear(scene):  # bvh.rs:1472-1515
    for o in ORIGINS:
        for d in _dirs():
            r = ray(o, d)
            a = scene.intersect(r)
            b = scene.linear_scan(r)
            assert (a is None) == (b is None)
            if a is not None:
                assert abs(a.t - b.t) < 1e-4 and a.geom_id == b.geom_id
            assert scene.occluded(r) == (b is not None)


def test_matches_linear_scan():  # bvh.rs:1518-1521
    assert_matches_linear(sphere_grid_scene(4))


def test_spatial_splits_match_linear_scan():  # bvh.rs:1524-1527
    assert_matches_linear(shards_scene(64))


def _leaf_ref_count(scene):  # bvh.rs:431-434
    nodes, leaves, packets, indices = scene.arrays()
    packed = sum(bin(int(p[40])).count("1") for p in packets)  # word 40 = active
    return packed + len(indices)


def test_spatial_splits_duplicate_references():  # bvh.rs:1533-1541
    s = shards_scene(64)
    assert _leaf_ref_count(s) > s.primitive_count()


def test_triangles_are_packed_into_simd_lanes():  # bvh.rs:1546-1566
    s = shards_scene(64)
    c = s.counts()
    assert c["packets"] > 0 and c["indices"] == 0
    g = sphere_grid_scene(4)
    c = g.counts()
    assert c["packets"] == 0 and c["indices"] == _leaf_ref_count(g)
    b = ora.SceneBuilder()
    v, i = fx.diagonal_shards(16)
    b.attach_triangles(v, i)
    for cc in fx.sphere_grid_centers(2):
        b.attach_sphere(cc, 0.5)
    m = b.commit()
    c = m.counts()
    assert c["packets"] > 0 and c["indices"] > 0 and _leaf_ref_count(m) >= m.primitive_count()


def test_packets_are_well_filled():  # bvh.rs:1571-1579 (reference measures ~2.9 of 4, docs/simd.md:176-178)
    s = shards_scene(256)
    _, _, packets, _ = s.arrays()
    lanes = sum(bin(int(p[40])).count("1") for p in packets)
    avg = lanes / len(packets)
    assert avg >= 2.5
    assert abs(avg - 2.89) < 0.15, avg


def test_hit_any_matches_hit():  # bvh.rs:1583-1607
    s = sphere_grid_scene(4)
    dirs = [np.array([1, 0, 0], np.float32), _norm3(np.array([-1, 0.05, 0.02], np.float32)),
            _norm3(np.array([0.3, 0.3, 1.0], np.float32))]
    for o in [(-5, 4.5, 4.5), (20, 3, 3), (4.5, 4.5, 4.5)]:
        for d in dirs:
            for t_max in (0.5, 3.0, INF):
                r = ray(o, d)
                assert s.occluded(r, 0.001, t_max) == (s.intersect(r, 0.001, t_max) is not None)


def test_wide_node_is_two_cache_lines():  # bvh.rs:1613-1616
    assert C.sizeof(ora.WideNode) == 128 and C.sizeof(ora.Tri4) == 192 and C.sizeof(ora.Leaf) == 16


def test_build_is_deterministic():  # bvh.rs:1621-1638
    a, b = sphere_grid_scene(6).arrays(), sphere_grid_scene(6).arrays()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_collapse_widens_the_tree():  # bvh.rs:1643-1660
    s = sphere_grid_scene(6)
    nodes, leaves, _, _ = s.arrays()
    n_leaf_slots = sum(bin((int(n[28]) >> 4) & 0xF).count("1") for n in nodes)  # word 28 = flags
    assert n_leaf_slots > 0 and n_leaf_slots == len(leaves)
    assert len(nodes) * 2 < max(n_leaf_slots, 2) * 2 - 1


def test_empty_bvh_misses():  # bvh.rs:1663-1668
    s = ora.SceneBuilder().commit()
    assert s.intersect(ray((0, 0, 0), (1, 0, 0))) is None and s.bounds() is None


def test_bounds_cover_all_prims():  # bvh.rs:1671-1676
    b = sphere_grid_scene(3).bounds()
    assert (b[:3] <= -0.5).all() and (b[3:] >= 6.5).all()


def test_eight_wide_packets_would_not_reduce_vector_rounds():  # bvh.rs:1738-1765
    b = ora.SceneBuilder()
    v, i = fx.uv_sphere((0, 0, 0), 1.0, 80, 40)
    b.attach_triangles(v, i)
    s = b.commit()
    _, leaves, packets, _ = s.arrays()
    per_leaf = [sum(bin(int(packets[k][40])).count("1") for k in range(l[0], l[0] + l[1])) for l in leaves]
    assert 0 < max(per_leaf) <= 4
    assert sum(-(-n // 4) for n in per_leaf) == sum(-(-n // 8) for n in per_leaf)


# ---------------------------------------------------------------- scene.rs + lib.rs doc-test
def unit_sphere_scene():
    b = ora.SceneBuilder()
    b.attach_sphere((0, 0, 0), 1.0)
    return b.commit()


def test_doc_example():  # lib.rs:9-23
    b = ora.SceneBuilder()
    ball = b.attach_sphere((0, 0, 0), 1.0)
    s = b.commit()
    h = s.intersect(ray((0, 0, -5), (0, 0, 1)))
    assert h.geom_id == ball and abs(h.t - 4.0) < 1e-4
    assert not s.occluded(ray((0, 0, -5), (0, 0, 1)), 0.001, 3.9)


def test_reserved_slots_keep_ids_dense_and_stay_invisible():  # scene.rs:499-537
    b = ora.SceneBuilder()
    a = b.attach_sphere((-5, 0, 0), 1.0)
    ph = b.attach_empty()
    c = b.attach_sphere((5, 0, 0), 1.0)
    assert (a, ph, c) == (0, 1, 2)
    s = b.commit()
    assert s.geometry_count() == 3 and s.primitive_count() == 2
    b = ora.SceneBuilder()
    b.attach_sphere((-5, 0, 0), 1.0)
    slot = b.attach_empty()
    b.set_sphere(slot, (0, 0, 0), 1.0)
    s = b.commit()
    assert s.primitive_count() == 2
    h = s.intersect(ray((0, 0, -8), (0, 0, 1)), 1e-4, 3.4028235e38)
    assert h.geom_id == slot


def test_ids_map_back_to_geometries():  # scene.rs:540-579
    b = ora.SceneBuilder()
    ball = b.attach_sphere((-3, 0, 0), 1.0)
    quad = b.attach_triangles([(2, -1, -1), (2, -1, 1), (2, 1, 1), (2, 1, -1)], [(0, 1, 2), (0, 2, 3)])
    s = b.commit()
    assert s.geometry_count() == 2 and s.primitive_count() == 3
    assert s.intersect(ray((-3, 0, -5), (0, 0, 1))).geom_id == ball
    h = s.intersect(ray((0, 0.5, -0.5), (1, 0, 0)))
    assert h.geom_id == quad and h.prim_id == 1


def test_front_face_semantics_match_ray_side():  # scene.rs:582-596
    s = unit_sphere_scene()
    o = s.intersect(ray((0, 0, -5), (0, 0, 1)), 0.001, 100.0)
    assert o.front_face and np.allclose(o.normal.np(), (0, 0, -1), atol=1e-4)
    i = s.intersect(ray((0, 0, 0), (0, 0, 1)), 0.001, 100.0)
    assert not i.front_face and np.allclose(i.normal.np(), (0, 0, -1), atol=1e-4)


def tr(x, y, z):
    return ora.affine(t=(x, y, z))


def test_motion_flags():  # scene.rs:603-652
    assert not unit_sphere_scene().has_motion()
    b = ora.SceneBuilder()
    b.attach_instance(unit_sphere_scene(), tr(3, 0, 0))
    assert not b.commit().has_motion()
    mid = ora.SceneBuilder()
    mid.attach_instance(unit_sphere_scene(), ora.IDENTITY12, tr(4, 0, 0))
    mid = mid.commit()
    assert mid.has_motion()
    root = ora.SceneBuilder()
    root.attach_instance(mid, tr(0, 7, 0))
    assert root.commit().has_motion()


def nested_scene():
    leaf = unit_sphere_scene()
    mid = ora.SceneBuilder()
    for x in (-2.0, 2.0):
        mid.attach_instance(leaf, tr(x, 0, 0))
    mid = mid.commit()
    root = ora.SceneBuilder()
    for y in (-5.0, 5.0):
        root.attach_instance(mid, tr(0, y, 0))
    return root.commit()


def test_unique_breakdown_counts_shared_prototypes_once():  # scene.rs:716-748
    s = nested_scene()
    top = s.primitive_breakdown()
    assert top["instances"] == 2 and top["spheres"] == 0
    unique = s.unique_primitive_breakdown()
    assert unique["spheres"] == 1, "shared prototype counted more than once"
    assert unique["instances"] == 4


def test_instances_nest():  # scene.rs:655-710
    s = nested_scene()
    assert s.primitive_count() == 2
    for (x, y) in [(-2, -5), (2, -5), (-2, 5), (2, 5)]:
        r = ray((x, y, -8), (0, 0, 1))
        h = s.intersect(r, 0.001, 100.0)
        assert h is not None and abs(h.t - 7.0) < 1e-3
        assert np.allclose(h.normal.np(), (0, 0, -1), atol=1e-4)
        assert s.occluded(r, 0.001, 100.0)
    assert s.intersect(ray((0, 0, -8), (0, 0, 1)), 0.001, 100.0) is None


def test_nested_instances_compose_transforms_and_normals():  # scene.rs:755-803
    mid = ora.SceneBuilder()
    mid.attach_instance(unit_sphere_scene(), ora.affine(np.diag([2.0, 1.0, 1.0])))
    mid = mid.commit()
    a = math.pi / 2
    # glam from_rotation_z(angle): x_axis = (cos, sin, 0), y_axis = (-sin, cos, 0)
    s_, c_ = f32(math.sin(f32(a))), f32(math.cos(f32(a)))
    rz = np.array([[c_, -s_, 0], [s_, c_, 0], [0, 0, 1]], np.float32)
    root = ora.SceneBuilder()
    root.attach_instance(mid, ora.affine(rz))
    s = root.commit()
    ay = s.intersect(ray((0, -8, 0), (0, 1, 0)), 0.001, 100.0)
    assert ay is not None and abs(ay.t - 6.0) < 1e-3
    ax = s.intersect(ray((-8, 0, 0), (1, 0, 0)), 0.001, 100.0)
    assert ax is not None and abs(ax.t - 7.0) < 1e-3
    assert np.allclose(ay.normal.np(), (0, -1, 0), atol=1e-4)


def test_nested_instances_respect_masks_at_each_level():  # scene.rs:808-837
    mid = ora.SceneBuilder()
    mid.attach_instance(unit_sphere_scene(), ora.IDENTITY12, None, MASK_SHADOW)
    mid = mid.commit()
    root = ora.SceneBuilder()
    root.attach_instance(mid, ora.IDENTITY12, None, MASK_SHADOW | MASK_CAMERA)
    s = root.commit()
    assert s.intersect(ray((0, 0, -5), (0, 0, 1), mask=MASK_CAMERA), 0.001, 100.0) is None
    assert s.intersect(ray((0, 0, -5), (0, 0, 1), mask=MASK_SHADOW), 0.001, 100.0) is not None


def test_masks_filter_by_ray_category():  # scene.rs:840-855
    b = ora.SceneBuilder()
    b.attach_sphere((0, 0, 0), 1.0, MASK_SHADOW)
    s = b.commit()
    rc, rs = ray((0, 0, -5), (0, 0, 1), mask=MASK_CAMERA), ray((0, 0, -5), (0, 0, 1), mask=MASK_SHADOW)
    assert s.intersect(rc, 0.001, 100.0) is None and s.intersect(rs, 0.001, 100.0) is not None
    assert not s.occluded(rc, 0.001, 100.0) and s.occluded(rs, 0.001, 100.0)


def test_smooth_normals_interpolate():  # scene.rs:858-882
    def n(v):
        return _norm3(np.array(v, np.float32))
    b = ora.SceneBuilder()
    b.attach_triangles([(-1, -1, 2), (1, -1, 2), (0, 1, 2)], [(0, 1, 2)],
                       [n((-0.5, 0, -1)), n((0.5, 0, -1)), n((0, 0.5, -1))])
    h = b.commit().intersect(ray((0.6, -0.7, 0), (0, 0, 1)), 0.001, 100.0)
    assert h.normal.x > 0.1 and h.normal.z < 0.0


def test_translated_instance_matches_baked():  # scene.rs:885-899
    b = ora.SceneBuilder()
    b.attach_instance(unit_sphere_scene(), tr(3, 0, 0))
    s = b.commit()
    r = ray((3, 0, -5), (0, 0, 1))
    h = s.intersect(r)
    assert abs(h.t - 4.0) < 1e-4 and np.allclose(h.normal.np(), (0, 0, -1), atol=1e-4)
    assert s.occluded(r) and not s.occluded(r, 0.001, 3.9)


def test_nonuniform_scale_transforms_normals_correctly():  # scene.rs:902-925
    b = ora.SceneBuilder()
    b.attach_instance(unit_sphere_scene(), ora.affine(np.diag([2.0, 1.0, 1.0])))
    h = b.commit().intersect(ray((1, 5, 0), (0, -1, 0)))
    e = np.array([0.5, math.sqrt(3.0), 0.0])
    e = e / np.linalg.norm(e)
    assert np.allclose(h.normal.np(), e, atol=1e-3)


def test_rotated_instance_hits_where_baked_triangle_would():  # scene.rs:928-955
    inner = ora.SceneBuilder()
    inner.attach_triangles([(-1, -1, 0), (1, -1, 0), (0, 1, 0)], [(0, 1, 2)])
    a = f32(math.pi / 2)
    s_, c_ = f32(math.sin(a)), f32(math.cos(a))
    ry = np.array([[c_, 0, s_], [0, 1, 0], [-s_, 0, c_]], np.float32)  # glam from_rotation_y
    m = ora.affine(ry, ry @ np.array([0, 0, 2], np.float32))
    b = ora.SceneBuilder()
    b.attach_instance(inner.commit(), m)
    h = b.commit().intersect(ray((5, 0, 0), (-1, 0, 0)))
    assert abs(h.t - 3.0) < 1e-4 and np.allclose(h.normal.np(), (1, 0, 0), atol=1e-4)


def test_motion_blur_interpolates_position():  # scene.rs:958-981
    b = ora.SceneBuilder()
    b.attach_instance(unit_sphere_scene(), ora.IDENTITY12, tr(4, 0, 0))
    s = b.commit()
    assert s.intersect(ray((0, 0, -5), (0, 0, 1), time=0.0)) is not None
    assert s.intersect(ray((4, 0, -5), (0, 0, 1), time=1.0)) is not None
    assert s.intersect(ray((0, 0, -5), (0, 0, 1), time=1.0)) is None
    h = s.intersect(ray((2, 0, -5), (0, 0, 1), time=0.5))
    assert h is not None and abs(h.t - 4.0) < 1e-4
    bb = s.bounds()
    assert bb[0] <= -1.0 and bb[3] >= 5.0


def test_instance_hits_report_instance_geom_id_and_inner_prim_id():  # scene.rs:984-1015
    inner = ora.SceneBuilder()
    inner.attach_triangles([(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)], [(0, 1, 2), (0, 2, 3)])
    b = ora.SceneBuilder()
    b.attach_sphere((0, -100, 0), 1.0)
    inst = b.attach_instance(inner.commit(), tr(0, 0, 5))
    h = b.commit().intersect(ray((-0.5, 0.5, 0), (0, 0, 1)))
    assert h.geom_id == inst and h.prim_id == 1


# ---- scene.rs:602-655: has_motion ----
def _unit_sphere_scene(api):
    b = api.SceneBuilder()
    b.attach_sphere((0.0, 0.0, 0.0), 1.0)
    return b.commit()


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_motion_flag_reports_a_static_scene_as_static(which, crt):  # scene.rs:602-619
    api = ora if which == "oracle" else crt
    assert not _unit_sphere_scene(api).has_motion()
    b = api.SceneBuilder()
    b.attach_instance(_unit_sphere_scene(api), api.affine(None, (3.0, 0.0, 0.0)))
    assert not b.commit().has_motion()  # a static placement of static geometry is still static


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_motion_flag_propagates_through_nesting(which, crt):  # scene.rs:621-655
    api = ora if which == "oracle" else crt
    mid = api.SceneBuilder()
    mid.attach_instance(_unit_sphere_scene(api), api.affine(), api.affine(None, (4.0, 0.0, 0.0)))
    mid = mid.commit()
    assert mid.has_motion(), "the level that authored the motion"
    root = api.SceneBuilder()
    root.attach_instance(mid, api.affine(None, (0.0, 7.0, 0.0)))
    assert root.commit().has_motion(), "a static placement of a moving scene is still moving"
