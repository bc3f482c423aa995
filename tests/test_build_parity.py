"""The product's host builder (C++, crust-render_amd/csrc/bvh_build.cpp) must emit the same tree as the
oracle's independent restatement (plain C, oracle/ora_rt.c): node bounds, child links, flags, leaf ranges,
packet lanes and index order, word for word. Runs on CPU (the builder is host code)."""
import numpy as np
import pytest

import ora
import scenes


@pytest.mark.parametrize("name", list(scenes.ALL))
def test_tree_matches_oracle(crt, name):
    make, _ = scenes.ALL[name]
    a = make(ora)
    b = make(crt)
    on, ol, op, oi = a.arrays()
    pn, pl, pp, pi, counts = b.tree()
    assert counts == a.counts()
    assert np.array_equal(on[:, :29], pn[:, :29])  # bounds, children, flags (pad words excluded)
    assert np.array_equal(ol, pl)
    assert np.array_equal(op[:, :47], pp[:, :47])  # word 47 is the device-only normal_ok nibble
    assert np.array_equal(oi, pi)
    ob, pb = a.bounds(), b.bounds()
    assert (ob is None and pb is None) or np.array_equal(ob.view(np.uint32), pb.view(np.uint32))
    assert a.primitive_count() == b.primitive_count() and a.geometry_count() == b.geometry_count()
    assert a.has_motion() == b.has_motion()


def _same_tree(a, b):
    on, ol, op, oi = a.arrays()
    pn, pl, pp, pi, counts = b.tree()
    assert counts == a.counts()
    assert np.array_equal(on[:, :29], pn[:, :29]) and np.array_equal(ol, pl)
    assert np.array_equal(op[:, :47], pp[:, :47]) and np.array_equal(oi, pi)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_random_scenes_build_the_same_trees(crt, seed):
    """Random triangle soups (degenerate, coincident, lattice-aligned), spheres and nested / moving instances: the
    top-level tree and every prototype tree are identical word for word."""
    import fuzz_scenes
    recipe = fuzz_scenes.recipe(seed)
    a, a_protos = fuzz_scenes.build(ora, recipe)
    b, b_protos = fuzz_scenes.build(crt, recipe)
    _same_tree(a, b)
    for x, y in zip(a_protos, b_protos):
        _same_tree(x, y)
    ob, pb = a.bounds(), b.bounds()
    assert np.array_equal(ob.view(np.uint32), pb.view(np.uint32)) and a.has_motion() == b.has_motion()


def test_empty_and_reserved_slots(crt):
    b = crt.SceneBuilder()
    s = b.commit()
    assert s.bounds() is None and s.primitive_count() == 0
    b = crt.SceneBuilder()
    a = b.attach_sphere((-5, 0, 0), 1.0)
    ph = b.attach_empty()
    c = b.attach_sphere((5, 0, 0), 1.0)
    assert (a, ph, c) == (0, 1, 2) and b.count() == 3
    s = b.commit()
    assert s.geometry_count() == 3 and s.primitive_count() == 2  # scene.rs:499-516
    b = crt.SceneBuilder()
    with pytest.raises(crt.CrtError) as e:
        b.set_sphere(7, (0, 0, 0), 1.0)  # scene.rs:197-201: the reference panics
    assert e.value.code == -2


def test_unique_breakdown_counts_shared_prototypes_once(crt):  # scene.rs:716-748, through the C ABI
    leaf = crt.SceneBuilder()
    leaf.attach_sphere((0, 0, 0), 1.0)
    leaf = leaf.commit()
    mid = crt.SceneBuilder()
    for x in (-2.0, 2.0):
        mid.attach_instance(leaf, [1, 0, 0, 0, 1, 0, 0, 0, 1, x, 0, 0])
    mid = mid.commit()
    root = crt.SceneBuilder()
    for y in (-5.0, 5.0):
        root.attach_instance(mid, [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, y, 0])
    s = root.commit()
    top, unique = s.primitive_breakdown(), s.unique_primitive_breakdown()
    assert top["instances"] == 2 and top["spheres"] == 0
    assert unique["spheres"] == 1 and unique["instances"] == 4


def test_out_of_range_indices_are_skipped(crt):
    b = crt.SceneBuilder()
    b.attach_triangles([(0, 0, 0), (1, 0, 0), (0, 1, 0)], [(0, 1, 2), (0, 1, 9)])  # scene.rs:251-253
    assert b.commit().primitive_count() == 1


@pytest.mark.parametrize("name", list(scenes.ALL))
def test_primitive_extents_match_oracle(crt, name):
    """Scene::primitive_extents (scene.rs:446-455): count, scene diagonal, mean and max primitive diagonal, bit for bit."""
    make, _ = scenes.ALL[name]
    a, b = make(ora).primitive_extents(), make(crt).primitive_extents()
    assert a[0] == b[0]
    assert np.array_equal(np.array(a[1:], np.float32).view(np.uint32), np.array(b[1:], np.float32).view(np.uint32))
    if a[0]:
        assert 0.0 < b[2] <= b[3] * 1.0001 and b[3] <= b[1] * 1.0001  # mean <= max <= scene diagonal, up to f32 rounding


def test_primitive_extents_of_an_empty_scene(crt):
    assert crt.SceneBuilder().commit().primitive_extents() == (0, 0.0, 0.0, 0.0)


def test_commit_refuses_instance_nesting_beyond_the_kernel_frames(crt):
    """The device kernels carry 8 instance frames (the importer's own limit, usd_import.rs:60): a scene 8 levels deep
    commits, a 9th level is refused at commit with the reason in crt_last_error — never skipped at trace time."""
    b = crt.SceneBuilder()
    b.attach_sphere((0, 0, 0), 1.0)
    s = b.commit()
    for depth in range(2, 9):  # levels 2..8
        b = crt.SceneBuilder()
        b.attach_instance(s)
        s = b.commit()
        assert s.primitive_count() == 1
    b = crt.SceneBuilder()
    b.attach_instance(s)
    with pytest.raises(RuntimeError, match="nesting"):
        b.commit()


def test_large_builds_use_a_bounded_number_of_helper_threads(crt):
    """A build large enough to fork at many levels (the reference forks with rayon::join on a bounded pool,
    bvh.rs:1156-1162): the tree equals the oracle's, and the process never holds more threads than cores + a few."""
    import threading
    import time
    rng = np.random.default_rng(3)
    n = 120_000
    c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
    verts = (c + rng.uniform(-0.2, 0.2, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    peak = [0]
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            with open("/proc/self/status") as f:
                for line in f:
                    if line.startswith("Threads:"):
                        peak[0] = max(peak[0], int(line.split()[1]))
            time.sleep(0.002)
    base = threading.active_count()
    with open("/proc/self/status") as f:
        before = next(int(l.split()[1]) for l in f if l.startswith("Threads:"))
    t = threading.Thread(target=watch)
    t.start()
    b = crt.SceneBuilder()
    b.attach_triangles(verts, idx)
    s = b.commit()
    stop.set()
    t.join()
    import os
    assert peak[0] - before <= (os.cpu_count() or 1) + 2, (peak[0], before, base)
    a = ora.SceneBuilder()
    a.attach_triangles(verts, idx)
    _same_tree(a.commit(), s)
