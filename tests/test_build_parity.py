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
