"""Scene recipes shared by the parity tests: each builds the SAME geometry through any SceneBuilder-shaped
API (the oracle's tests/ora.py or the product's crust_render_amd), so both sides see identical inputs."""
import math

import numpy as np

import fixtures as fx

f32 = np.float32


def tri_spheres(api):  # ray_throughput.rs:50-65
    b = api.SceneBuilder()
    for v, i in fx.tri_spheres():
        b.attach_triangles(v, i)
    return b.commit()


def sphere_grid(api):  # ray_throughput.rs:67-80
    b = api.SceneBuilder()
    for c in fx.sphere_grid_probe():
        b.attach_sphere(c, 0.6)
    return b.commit()


def instances(api):  # ray_throughput.rs:82-101
    inner = api.SceneBuilder()
    v, i = fx.uv_sphere((0, 0, 0), 1.0, 24, 12)
    inner.attach_triangles(v, i)
    inner = inner.commit()
    b = api.SceneBuilder()
    for t in fx.instance_translations():
        b.attach_instance(inner, api.affine(t=t))
    return b.commit()


def shards(api, n=64):  # bvh.rs:1443-1458
    b = api.SceneBuilder()
    v, i = fx.diagonal_shards(n)
    b.attach_triangles(v, i)
    return b.commit()


def small_sphere_grid(api, n=4):  # bvh.rs:1423-1438
    b = api.SceneBuilder()
    for c in fx.sphere_grid_centers(n):
        b.attach_sphere(c, 0.5)
    return b.commit()


def mixed(api):
    """Triangles + spheres + masked geometry + smooth normals + a nested, scaled, rotated instance tree and a
    motion-blurred instance: every primitive kind and every traversal branch in one scene."""
    leaf = api.SceneBuilder()
    v, i = fx.uv_sphere((0, 0, 0), 1.0, 16, 8)
    nrm = v / np.linalg.norm(v, axis=1, keepdims=True)
    leaf.attach_triangles(v, i, nrm.astype(np.float32))
    leaf.attach_sphere((0.0, 1.6, 0.0), 0.5)
    leaf = leaf.commit()
    mid = api.SceneBuilder()
    mid.attach_instance(leaf, api.affine(np.diag([1.5, 1.0, 0.75]).astype(np.float32), (-2.0, 0.0, 0.0)))
    a = f32(0.6)
    rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], np.float32)
    mid.attach_instance(leaf, api.affine(rz, (2.0, 0.5, 0.0)))
    mid = mid.commit()
    b = api.SceneBuilder()
    v, i = fx.diagonal_shards(48)
    b.attach_triangles((v * f32(0.4) - f32(4.0)).astype(np.float32), i)
    for c in fx.sphere_grid_centers(3):
        b.attach_sphere((c * f32(0.8) + np.array([3.0, -4.0, -3.0], np.float32)).astype(np.float32), 0.45)
    b.attach_sphere((0.0, 0.0, 6.0), 1.0, mask=2)  # shadow rays only
    b.attach_instance(mid, api.affine(t=(0.0, 3.0, 0.0)))
    b.attach_instance(mid, api.affine(np.diag([0.5, 0.5, 0.5]).astype(np.float32), (0.0, -3.0, 2.0)), mask=1 | 4)
    b.attach_instance(leaf, api.affine(t=(5.0, 0.0, 0.0)), api.affine(t=(7.0, 1.0, 0.0)))  # motion blur
    floor = np.array([(-8, -6, -8), (8, -6, -8), (8, -6, 8), (-8, -6, 8)], np.float32)
    b.attach_triangles(floor, [(0, 1, 2), (0, 2, 3)])
    return b.commit()


ALL = {"tri_spheres": (tri_spheres, 6.0), "sphere_grid": (sphere_grid, 14.0), "instances": (instances, 7.0),
       "shards": (shards, 12.0), "small_sphere_grid": (small_sphere_grid, 6.0), "mixed": (mixed, 8.0)}
