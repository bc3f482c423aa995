"""Oracle-side counterpart of crust_render_amd.load_usda: same SceneDesc, built through the oracle."""
import ctypes as C
import os

import numpy as np

import ora

STRATEGY = {"power": 0, "mis": 0, "balance": 1, "light": 2, "bsdf": 3}
FILTER = {"box": 0, "triangle": 1}


def make_lights(light_dicts):
    arr = (ora.Light * max(len(light_dicts), 1))()
    for k, d in enumerate(light_dicts):
        l = arr[k]
        l.kind = {"sphere": ora.LIGHT_SPHERE, "rect": ora.LIGHT_RECT, "distant": ora.LIGHT_DISTANT,
                  "dome": ora.LIGHT_DOME}[d["kind"]]
        l.geom_id = int(d["geom_id"])
        l.radiance[:] = [float(x) for x in d["radiance"]]
        if d["kind"] == "distant":  # derived form, see include/crt.h
            l.normal[:] = [float(x) for x in d["direction"]]
            l.radius = float(d["cos_half_angle"])
            l.center[0] = float(d["solid_angle"])
        elif d["kind"] == "dome":
            pass
        elif d["kind"] == "sphere":
            l.center[:] = [float(x) for x in d["center"]]
            l.radius = float(d["radius"])
        else:
            for f in ("origin", "edge_u", "edge_v", "normal"):
                getattr(l, f)[:] = [float(x) for x in d[f]]
    return arr


class OracleRenderer:
    def __init__(self, desc, usda_mod, width=None, height=None, max_depth=None, forward=1, variance=0.0, min_spp=None):
        self.desc = desc
        self.scene, mats, self._protos = usda_mod.build_world(desc, ora, ora.default_material)
        self._mats = (ora.Material * max(len(mats), 1))(*mats)
        self._lights = make_lights(desc.lights)
        s = desc.settings
        cam = ora.Camera()
        c = desc.camera
        ora.lib().ora_camera_new(C.byref(cam), ora.v3(c["lookfrom"]), ora.v3(c["lookat"]), ora.v3(c["vup"]),
                                 float(c["vfov_deg"]), float(c["aspect"]), float(c["aperture"]), float(c["focus_dist"]))
        self.job = ora.RenderJob()
        j = self.job
        j.scene = self.scene.h
        j.materials = self._mats
        j.n_materials = len(mats)
        j.lights = self._lights
        j.n_lights = len(desc.lights)
        j.camera = cam
        j.width, j.height = s["width"], s["height"]
        j.spp = s["spp"]
        j.max_depth = s["max_depth"] if max_depth is None else max_depth
        j.min_spp = s["min_spp"] if min_spp is None else min_spp
        j.variance_threshold = float(variance)
        j.frame = s["frame"]
        j.strategy = STRATEGY[s["strategy"]]
        j.filter_kind = FILTER[s["filter"]]
        j.filter_radius = s["filter_radius"]
        j.forward = forward

    def render(self, spp, threads=None, forward=None):
        if forward is not None:
            self.job.forward = forward
        self.job.spp = spp
        img = np.zeros((self.job.height, self.job.width, 3), dtype=np.float32)
        st = ora.RayStats()
        ora.lib().ora_render(C.byref(self.job), ora._fp(img), C.byref(st), threads or (os.cpu_count() or 1))
        return img, st

    def render_serial_trav(self, spp, forward=1):
        """Whole frame on the calling thread -> (image, RayStats, TravStats of the closest-hit queries, of the any-hit
        queries): the oracle's traversal counters (bvh.rs:39-57), the two query kinds apart as on the device."""
        self.job.forward = forward
        self.job.spp = spp
        img = np.zeros((self.job.height, self.job.width, 3), dtype=np.float32)
        st, closest, anyhit = ora.RayStats(), ora.TravStats(), ora.TravStats()
        ora.lib().ora_render_serial_trav(C.byref(self.job), ora._fp(img), C.byref(st), C.byref(closest), C.byref(anyhit))
        return img, st, closest, anyhit

    def render_pixels(self, idx, spp, threads=None, forward=None):
        """The pixels idx (linear buffer indices j*width+i) only -> ([n, 3] means, RayStats over those pixels)."""
        if forward is not None:
            self.job.forward = forward
        self.job.spp = spp
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        out = np.zeros((idx.size, 3), dtype=np.float32)
        st = ora.RayStats()
        ora.lib().ora_render_pixels(C.byref(self.job), idx.ctypes.data_as(C.POINTER(C.c_uint32)), idx.size, ora._fp(out),
                                    C.byref(st), threads or (os.cpu_count() or 1))
        return out, st
