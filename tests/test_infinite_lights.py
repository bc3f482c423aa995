"""Lights at infinity (DistantLight, uniform DomeLight — light.rs:214-392) through the oracle's integrator and the
scene importer. The GPU side of the same scenes is in test_gpu_render.py / test_gpu_fuzz.py."""
import importlib
import os

import numpy as np

import ora_world

usda = importlib.import_module("crust-render_amd.usda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _world(strategy, lights, width=24, height=16, geoms=True):
    d = usda.SceneDesc()
    if geoms:
        g = np.array([(-8, 0, -8), (8, 0, -8), (8, 0, 8), (-8, 0, 8)], dtype=np.float32)
        d.geoms.append(dict(kind="mesh", verts=g, idx=np.array([(0, 2, 1), (0, 3, 2)], np.uint32), mask=0xFFFFFFFF,
                            material={"base_color": (0.6, 0.6, 0.6), "specular_roughness": 0.5}, name="ground"))
        d.geoms.append(dict(kind="sphere", center=np.array([0, 1, 0], np.float32), radius=np.float32(1.0),
                            mask=0xFFFFFFFF, material={"base_color": (0.8, 0.3, 0.2), "specular_roughness": 0.3},
                            name="ball"))
    else:  # the builder wants something to build over: a speck far behind the camera
        d.geoms.append(dict(kind="sphere", center=np.array([0, 0, 500], np.float32), radius=np.float32(0.01),
                            mask=0xFFFFFFFF, material={"base_color": (0.5, 0.5, 0.5)}, name="speck"))
    d.lights.extend(lights)
    d.camera = dict(lookfrom=np.array([0, 2.5, 7], np.float32), lookat=np.array([0, 1, 0], np.float32),
                    vup=np.array([0, 1, 0], np.float32), vfov_deg=np.float32(45), aspect=np.float32(width / height),
                    aperture=np.float32(0), focus_dist=np.float32(7))
    d.settings = dict(usda.DEFAULTS, strategy=strategy, filter="box", filter_radius=0.5, width=width, height=height,
                      max_depth=5, frame=0)
    return d


def test_all_strategies_estimate_the_same_image():
    """escaped_emission's MIS weights (tracer.rs:966-1009) against NEE of the same lights: light-only, bsdf-only and
    both heuristics must converge on one image. A wrong weight on either side shows up as a brightness shift."""
    means = {}
    for s in ("power", "balance", "light", "bsdf"):
        lights = [usda.distant_light((0.4, -1, -0.3), (2, 2, 2), 30.0), usda.dome_light((0.2, 0.3, 0.5))]
        img, _ = ora_world.OracleRenderer(_world(s, lights), usda).render(256, forward=1)
        assert np.isfinite(img).all()
        means[s] = img.mean(axis=(0, 1))
    for s in ("balance", "light", "bsdf"):
        assert np.allclose(means[s], means["power"], rtol=0.02), (s, means[s], means["power"])


def test_dome_replaces_the_sky_gradient():
    """tracer.rs:1000-1008: the built-in gradient is added only where no light at infinity covers the direction."""
    tint = np.array([0.25, 0.5, 0.75], np.float32)
    img, st = ora_world.OracleRenderer(_world("power", [usda.dome_light(tint)], 8, 8, geoms=False), usda).render(4, forward=1)
    assert st.ended_escaped == st.camera_rays == 8 * 8 * 4
    assert np.allclose(img, tint[None, None, :], rtol=1e-6)
    sky, _ = ora_world.OracleRenderer(_world("power", [], 8, 8, geoms=False), usda).render(4, forward=1)
    assert not np.allclose(sky, tint[None, None, :], rtol=1e-2)  # without the dome: the gradient


def test_sun_outside_its_cone_leaves_the_gradient():
    """A distant light covers only its cone (light.rs:268-282): elsewhere the sky gradient still shows."""
    sun = usda.distant_light((0, -1, 0), (5, 5, 5), 2.0)  # straight down: its disc is at the zenith, out of view
    a, _ = ora_world.OracleRenderer(_world("power", [sun], 8, 8, geoms=False), usda).render(4, forward=1)
    b, _ = ora_world.OracleRenderer(_world("power", [], 8, 8, geoms=False), usda).render(4, forward=1)
    assert np.array_equal(a, b)


def test_importer_reads_distant_and_dome_lights():  # usd_import.rs:2360-2377, :2389-2460
    d = usda.load(os.path.join(ROOT, "scenes", "sun_sky.usda"), 96, 54)
    kinds = sorted(l["kind"] for l in d.lights)
    assert kinds == ["distant", "dome", "sphere"]
    sun = next(l for l in d.lights if l["kind"] == "distant")
    # rotateY(35) . rotateX(-48) applied to -Z
    rx, ry = np.radians(-48.0), np.radians(35.0)
    v = np.array([0.0, 0.0, -1.0])
    v = np.array([v[0], np.cos(rx) * v[1] - np.sin(rx) * v[2], np.sin(rx) * v[1] + np.cos(rx) * v[2]])
    v = np.array([np.cos(ry) * v[0] + np.sin(ry) * v[2], v[1], -np.sin(ry) * v[0] + np.cos(ry) * v[2]])
    assert np.allclose(sun["direction"], v, atol=1e-6)
    assert sun["geom_id"] == 0xFFFFFFFF
    assert abs(sun["cos_half_angle"] - np.cos(np.radians(2.0))) < 1e-6
    assert np.allclose(sun["radiance"], np.float32(3.0) * np.array([1.0, 0.93, 0.82], np.float32))
    dome = next(l for l in d.lights if l["kind"] == "dome")
    assert np.allclose(dome["radiance"], np.float32(0.6) * np.array([0.45, 0.62, 0.95], np.float32))
    # lights at infinity add no geometry: the only light geometry is the lamp's sphere
    assert sum(1 for g in d.geoms if g.get("material", {}).get("_preset") == "emissive") == 1
