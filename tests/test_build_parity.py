"""The product's host builder (C++, crust-render_amd/csrc/bvh_build.cpp) must emit the same tree as the
oracle's independent restatement (plain C, oracle/ora_rt.c): node bounds, child links, flags, leaf ranges,
packet lanes and index order, word for word. Runs on CPU (the builder is host code)."""
import os
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


# ---- the device image, checked on the host (crt_scene_image_check: no GPU, nothing uploaded) ----

@pytest.mark.parametrize("name", list(scenes.ALL))
def test_device_image_child_words_decode_to_their_leaves(crt, name):
    """Every child word of the flattened image — plain, direct (bit 30: the scalar list itself) and direct-instance
    (bit 27: consecutive instance slots) — decodes to exactly the node, leaf, list or slots it stands for; the node
    numbering is a permutation with the queried root at 0 and the instanced roots right behind it (the kernels' LDS
    window); every index is in range (crust-render_amd/csrc/scene.cpp, flatten_image)."""
    make, _ = scenes.ALL[name]
    st = make(crt).image_check()
    assert st["nodes"] > 0
    words = st["leaf_words_plain"] + st["leaf_words_direct_index"] + st["leaf_words_direct_instance"]
    assert words > 0
    if name == "instances":  # 125 placements of one prototype: instance-heavy -> direct leaves, every leaf a run of slots
        assert st["direct_leaves"] == 1 and st["instances"] == 125 and st["staged_roots"] == 1
        assert st["leaf_words_direct_instance"] > 0 and st["leaf_words_direct_index"] == 0
    if name in ("sphere_grid",):  # no triangles at all: direct leaves, lists of primitive indices
        assert st["direct_leaves"] == 1 and st["leaf_words_direct_index"] > 0 and st["leaf_words_direct_instance"] == 0
    if name in ("tri_spheres", "shards"):  # triangle packets only: plain leaf words
        assert st["direct_leaves"] == 0 and st["leaf_words_plain"] == words


@pytest.mark.parametrize("seed", [3, 7, 13, 16, 18, 21])
def test_device_image_of_random_scenes_is_consistent(crt, seed, monkeypatch):
    """The same self-check over the fuzz recipes (nested and moving instances, spheres, masks), with the direct forms
    forced on and off and both LDS splits: the encoder must be right for every combination the A/B knobs can reach."""
    import fuzz_scenes
    recipe = fuzz_scenes.recipe(seed)
    for env in ({}, {"CRT_DIRECT_LEAVES": "1"}, {"CRT_DIRECT_LEAVES": "1", "CRT_DIRECT_INST": "0"}, {"CRT_DIRECT_LEAVES": "0"},
                {"CRT_POOL_STACK_RT": "10", "CRT_DIRECT_LEAVES": "1"}, {"CRT_INST_ORDER": "0", "CRT_DIRECT_LEAVES": "1"}):
        for k in ("CRT_DIRECT_LEAVES", "CRT_DIRECT_INST", "CRT_POOL_STACK_RT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            if k != "CRT_INST_ORDER":  # read once per process (static): left at its default here
                monkeypatch.setenv(k, v)
        scene, _protos = fuzz_scenes.build(crt, recipe)
        st = scene.image_check()
        assert st["direct_leaves"] == (0 if env.get("CRT_DIRECT_LEAVES") == "0" else st["direct_leaves"])
        if env.get("CRT_DIRECT_INST") == "0":
            assert st["leaf_words_direct_instance"] == 0
        if env.get("CRT_DIRECT_LEAVES") == "0":
            assert st["leaf_words_direct_index"] == 0 and st["leaf_words_direct_instance"] == 0


def test_device_image_places_a_moving_instance_behind_the_normals(crt):
    """A moving instance's two placements live in the normals array, addressed through its flags word (16-byte aligned,
    behind the shading normals of smooth meshes); a static instance carries no offset."""
    import fixtures as fx
    inner = crt.SceneBuilder()
    v, i = fx.uv_sphere((0, 0, 0), 1.0, 8, 4)
    inner.attach_triangles(v, i, normals=v / np.linalg.norm(v, axis=1, keepdims=True))
    inner = inner.commit()
    b = crt.SceneBuilder()
    b.attach_instance(inner, crt.affine(t=(0, 0, 0)))
    b.attach_instance(inner, crt.affine(t=(3, 0, 0)), crt.affine(t=(3, 1, 0)))
    b.attach_instance(inner, crt.affine(t=(6, 0, 0)), crt.affine(t=(6, 0, 2)))
    st = b.commit().image_check()
    assert st["instances"] == 3 and st["moving_instances"] == 2 and st["staged_roots"] == 1


@pytest.mark.parametrize("knobs", [{"CRT_DIRECT_LEAVES": "0"}, {"CRT_DIRECT_LEAVES": "0", "CRT_DIRECT_INST": "0"},
                                   {"CRT_DIRECT_LEAVES": "1", "CRT_DIRECT_INST": "0"}])
def test_instance_heavy_images_carry_no_direct_word_the_engine_was_told_not_to_read(crt, knobs, monkeypatch):
    """The round-2 A/B abort (profiles/README.md, "The r02f abort"): an engine variant built WITHOUT the direct leaf form
    met direct words the host had still written — on PointInstancedMedCity, the first instance-heavy scene of the run —
    read a leaf index with bit 30 set, faulted, and took the box's GPU with it. The host now derives what it writes from
    the same switches the engine is built with (scene.cpp: `if (!CRT_DIRECT_LEAVES) direct = false`, and the run-time
    knobs of the same names); this checks the run-time half on the two instance-heavy images of the test set: no direct
    word of a form that is switched off, and every word still decodes to its leaf."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    make, _ = scenes.ALL["instances"]  # 125 placements of one prototype
    st = make(crt).image_check()
    desc = crt.usda.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "PointInstancedMedCity.usd"), 64, 36)
    scene, _mats, _protos = crt.usda.build_world(desc, crt, crt.default_material)
    for s in (st, scene.image_check()):
        assert s["instances"] >= 125
        if knobs["CRT_DIRECT_LEAVES"] == "0":
            assert s["direct_leaves"] == 0 and s["leaf_words_direct_index"] == 0 and s["leaf_words_direct_instance"] == 0
            assert s["leaf_words_plain"] > 0
        else:
            assert s["direct_leaves"] == 1 and s["leaf_words_direct_instance"] == 0 and s["leaf_words_direct_index"] > 0


_ENGINE_KNOBS = ("CRT_DIRECT_LEAVES", "CRT_DIRECT_INST", "CRT_WIDE", "CRT_POOL_STACK_RT", "CRT_COLD")


def test_every_knob_combination_selects_an_engine_that_can_decode_the_image(crt, monkeypatch):
    """The engine-vs-image class, closed in one place (crt_internal.h, select_engine): rounds 2 and 3 each lost a GPU box
    to a traversal-engine instance meeting an image form it was not built for (profiles/README.md, "The r02f abort",
    "The r03w fault": direct child words read by the four-wave kernels, which carry no direct-leaf engine). For every
    combination of the A/B knobs that shape the image or the choice, on the six scene recipes, what a launch in this
    process would select (want_wide = -2) decodes every child word of the image and keeps every cold field it needs —
    crt_scene_engine_select checks the selection against a census of the flattened image, on the host."""
    import itertools
    built = {name: make(crt) for name, (make, _) in scenes.ALL.items()}
    values = {"CRT_DIRECT_LEAVES": (None, "0", "1"), "CRT_DIRECT_INST": (None, "0"), "CRT_WIDE": (None, "0", "1"),
              "CRT_POOL_STACK_RT": (None, "3", "6", "10"), "CRT_COLD": (None, "7")}
    n_wide = n_direct = 0
    for combo in itertools.product(*(values[k] for k in _ENGINE_KNOBS)):
        for k, v in zip(_ENGINE_KNOBS, combo):
            monkeypatch.delenv(k, raising=False) if v is None else monkeypatch.setenv(k, v)
        env = dict(zip(_ENGINE_KNOBS, combo))
        for name, scene in built.items():
            img = scene.image_check()
            rsel = scene.engine_select(-3)  # the renderer's launches: large flat trees and direct-leaf images differ from the queries'
            if img["direct_leaves"] and env["CRT_WIDE"] != "0":
                assert rsel["wide"] == 2 and rsel["direct"] == 1, (name, env)  # four-wave kernels, direct-engine instances
            elif env["CRT_WIDE"] == "0":
                assert rsel["wide"] == 0, (name, env)
            else:
                assert rsel["wide"] in (0, 1) and rsel["direct_words"] == 0, (name, env)
            sel = scene.engine_select(-2)  # raises on a selection the image census contradicts
            words = img["leaf_words_direct_index"] + img["leaf_words_direct_instance"]
            assert sel["direct_words"] == words, (name, env)
            if words or img["direct_leaves"]:
                assert sel["wide"] == 0 and sel["direct"] == 1, (name, env)
                with pytest.raises(crt.CrtError) as refused:  # asked for outright: refused, never launched
                    scene.engine_select(1)
                assert refused.value.code == -5
            if sel["wide"]:
                assert words == 0 and (sel["lds_stack"], sel["window"]) in ((4, 26), (5, 12)), (name, env)
            else:
                assert sel["lds_stack"] in (6, 10) and sel["window"] == (16 if sel["lds_stack"] == 10 else 72), (name, env)
            assert sel["cold"] & ~sel["ext_cold"] == 0 and sel["cold"] & ~sel["path_cold"] == 0, (name, env)
            if env["CRT_WIDE"] == "1" and not img["direct_leaves"]:  # (the flag, not the census: select_engine is conservative)
                assert sel["wide"] == 1, (name, env)
            if env["CRT_WIDE"] == "0":
                assert sel["wide"] == 0, (name, env)
            n_wide += sel["wide"]
            n_direct += sel["direct"]
    assert n_wide > 0 and n_direct > 0  # both kinds of instance were exercised


def test_a_small_flat_scene_with_direct_leaves_forced_never_selects_the_four_wave_kernels(crt, monkeypatch):
    """The third way in the round-3 review found: CRT_DIRECT_LEAVES=1 on a small flat scene (cornellbox: packet-free
    leaves of its two instance placements) writes direct words while the old wide_split — pool_stack, packets and node
    count only — still chose the four-wave kernels. Same for CRT_POOL_STACK_RT=6 on an instance-heavy image."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "cornellbox.usda")
    scene, _mats, _protos = crt.usda.build_world(crt.usda.load(path, 64, 36), crt, crt.default_material)
    assert scene.engine_select(-2)["wide"] == 1 and scene.image_check()["direct_leaves"] == 0  # the shipped default
    monkeypatch.setenv("CRT_DIRECT_LEAVES", "1")
    sel, img = scene.engine_select(-2), scene.image_check()
    assert img["direct_leaves"] == 1 and img["leaf_words_direct_instance"] + img["leaf_words_direct_index"] > 0
    assert sel["wide"] == 0 and sel["direct"] == 1
    monkeypatch.delenv("CRT_DIRECT_LEAVES")
    monkeypatch.setenv("CRT_POOL_STACK_RT", "6")
    make, _ = scenes.ALL["instances"]  # 125 placements: direct leaves by default, 6 + 72 split forced
    s = make(crt)
    sel = s.engine_select(-2)
    assert s.image_check()["direct_leaves"] == 1 and sel["wide"] == 0 and sel["direct"] == 1 and sel["lds_stack"] == 6
