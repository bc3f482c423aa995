"""A host that keeps the reference's per-pixel integrator and calls somebody else's implementation of its two seams.

SURVEY §8(b): the kernel seam (World::intersect / occluded, rt_world.rs:207-237) and the Material / Light traits
(material.rs:26-116, light.rs:120-151) are what a host with its own trace_path (tracer.rs:1086-1558) calls. The oracle's
scalar trace_path plays that host: with OraSeamHooks set (oracle/ora_pt.h) every call it makes across those seams is handed,
one query at a time, to a KERNEL backend (intersect / occluded) and a SHADING backend (the six trait methods, in the C
ABI's record layouts, include/crt.h). Backends:
  DeviceKernel / DeviceShade   libcrt_amd.so through its C ABI (tests/test_gpu_seam_integrator.py)
  OracleKernel / DriversShade  the oracle's own traversal; the device's shading SOURCE compiled as host C++, or the oracle's
                               batched drivers (tests/test_seam_integrator_host.py, no GPU)
Test infrastructure only."""
import ctypes as C

import numpy as np

import ora
import seam_cases as sc


class DeviceKernel:
    """crt_intersect1 / crt_occluded1 (scene.rs:366-479 behind the C ABI)."""

    def __init__(self, crt, scene):
        self.crt, self.scene = crt, scene

    def _ray(self, r):
        c = self.crt.CrtRay()
        c.origin[:] = (r.origin.x, r.origin.y, r.origin.z)
        c.dir[:] = (r.dir.x, r.dir.y, r.dir.z)
        c.time, c.mask = r.time, r.mask
        return c

    def intersect(self, ray, t_min, t_max, out):
        h = self.crt.CrtRayHit()
        rc = self.crt.lib().crt_intersect1(self.scene.h, C.byref(self._ray(ray)), t_min, t_max, C.byref(h))
        assert rc in (0, 1), rc
        if rc == 1:
            out.t, out.front_face, out.u, out.v, out.geom_id, out.prim_id = h.t, int(h.front_face), h.u, h.v, h.geom_id, h.prim_id
            out.normal.x, out.normal.y, out.normal.z = h.normal[0], h.normal[1], h.normal[2]
        return rc

    def occluded(self, ray, t_min, t_max):
        rc = self.crt.lib().crt_occluded1(self.scene.h, C.byref(self._ray(ray)), t_min, t_max)
        assert rc in (0, 1), rc
        return rc


class OracleKernel:
    def __init__(self, scene):
        self.scene = scene

    def intersect(self, ray, t_min, t_max, out):
        return ora.lib().ora_intersect(self.scene.h, C.byref(ray), t_min, t_max, C.byref(out))

    def occluded(self, ray, t_min, t_max):
        return ora.lib().ora_occluded(self.scene.h, C.byref(ray), t_min, t_max)


class DeviceShade:
    """crt_material_*_n / crt_light_*_n on tables resident in HBM, one query per launch."""

    def __init__(self, crt, mats, lights):
        sh = crt.shading
        self.sh, self.mats, self.lights = sh, sh.DeviceMaterials(mats), sh.DeviceLights(lights)

    def _run(self, fn, q, dtype):
        return fn(self.sh.to_device(q)).cpu().numpy().view(dtype)

    def scatter(self, q): return self._run(self.mats.scatter_importance, q, sc.SCATTER_SAMPLE)
    def eval(self, q): return self._run(self.mats.eval, q, sc.BSDF_EVAL)
    def emitted(self, q): return self._run(self.mats.emitted_directional, q, np.float32).reshape(-1, 3)
    def light_sample(self, q): return self._run(self.lights.sample_li, q, sc.LIGHT_SAMPLE)
    def light_pdf(self, q): return self._run(self.lights.pdf_at_point, q, np.float32)
    def light_escaped(self, q): return self._run(self.lights.escaped, q, sc.LIGHT_SAMPLE)


class DriversShade:
    """seam_cases.Drivers (the oracle's batched drivers, or the device source compiled as host C++) on the oracle
    renderer's own tables."""

    def __init__(self, drivers, oracle_renderer):
        o = oracle_renderer
        self.d = drivers
        self.mats = np.frombuffer(o._mats, dtype=sc.MATERIAL)[:max(o.job.n_materials, 1)].copy()
        self.lights = np.frombuffer(o._lights, dtype=sc.LIGHT)[:max(o.job.n_lights, 1)].copy()

    def scatter(self, q): return self.d.scatter(self.mats, q)
    def eval(self, q): return self.d.eval(self.mats, q)
    def emitted(self, q): return self.d.emitted(self.mats, q)
    def light_sample(self, q): return self.d.light_sample(self.lights, q)
    def light_pdf(self, q): return self.d.light_pdf(self.lights, q)
    def light_escaped(self, q): return self.d.light_escaped(self.lights, q)


class SeamHost:
    """OraSeamHooks whose members marshal one call into the C ABI's records and hand it to the backends."""

    def __init__(self, kernel, shade):
        self.kernel, self.shade = kernel, shade
        self.calls = dict.fromkeys(("intersect", "occluded", "scatter", "eval", "emitted", "sample_li", "pdf_at_point",
                                    "escaped"), 0)
        H = ora.SeamHooks
        self.hooks = H(None, H.INTERSECT(self.intersect), H.OCCLUDED(self.occluded), H.MAT_SCATTER(self.scatter),
                       H.MAT_EVAL(self.eval), H.MAT_EMITTED(self.emitted), H.LIGHT_SAMPLE(self.sample_li),
                       H.LIGHT_PDF(self.pdf_at_point), H.LIGHT_ESCAPED(self.escaped))

    def render(self, oracle_renderer, spp, forward):
        """The oracle's integrator on this host's seams, one thread -> (image, RayStats)."""
        self.calls = dict.fromkeys(self.calls, 0)
        ora.set_seam_hooks(self.hooks)
        try:
            return oracle_renderer.render(spp, threads=1, forward=forward)
        finally:
            ora.set_seam_hooks(None)

    # -- kernel seam --
    def intersect(self, _ctx, ray, t_min, t_max, out):
        self.calls["intersect"] += 1
        return self.kernel.intersect(ray.contents, t_min, t_max, out.contents)

    def occluded(self, _ctx, ray, t_min, t_max):
        self.calls["occluded"] += 1
        return self.kernel.occluded(ray.contents, t_min, t_max)

    # -- Material (material.rs:26-116) --
    @staticmethod
    def _shade_query(material, ray_dir, rec):
        q = np.zeros(1, dtype=sc.SHADE_QUERY)
        rec = rec.contents
        q["ray_dir"][0] = (ray_dir[0], ray_dir[1], ray_dir[2])
        q["material"] = material
        q["p"][0] = (rec.p.x, rec.p.y, rec.p.z)
        q["t"] = rec.t
        q["normal"][0] = (rec.normal.x, rec.normal.y, rec.normal.z)
        q["front_face"] = 1 if rec.front_face else 0
        return q

    def scatter(self, _ctx, material, ray_dir, rec, dom_pattern, dom_index, out):
        self.calls["scatter"] += 1
        q = self._shade_query(material, ray_dir, rec)
        q["sampler_pattern"], q["sampler_index"] = dom_pattern, dom_index
        s = self.shade.scatter(q)[0]
        if not s["some"]:
            return 0
        o = out.contents
        o.origin.x, o.origin.y, o.origin.z = (float(x) for x in s["origin"])
        o.dir.x, o.dir.y, o.dir.z = (float(x) for x in s["dir"])
        o.value.x, o.value.y, o.value.z = (float(x) for x in s["value"])
        o.pdf = float(s["pdf"])
        o.delta = 1 if int(s["flags"]) & 1 else 0
        o.medium = 1 if int(s["flags"]) & 2 else 0
        return 1

    def eval(self, _ctx, material, ray_dir, rec, wi, value, pdf):
        self.calls["eval"] += 1
        q = self._shade_query(material, ray_dir, rec)
        q["wi"][0] = (wi[0], wi[1], wi[2])
        e = self.shade.eval(q)[0]
        if not e["some"]:
            return 0
        for k in range(3):
            value[k] = float(e["value"][k])
        pdf[0] = float(e["pdf"])
        return 1

    def emitted(self, _ctx, material, cos_theta_o, rgb):
        self.calls["emitted"] += 1
        q = np.zeros(1, dtype=sc.SHADE_QUERY)
        q["material"], q["cos_theta_o"] = material, cos_theta_o
        e = self.shade.emitted(q)[0]
        for k in range(3):
            rgb[k] = float(e[k])

    # -- Light (light.rs:120-151) --
    @staticmethod
    def _light_query(light, frm):
        q = np.zeros(1, dtype=sc.LIGHT_QUERY)
        q["light"] = light
        q["from"][0] = (frm[0], frm[1], frm[2])
        return q

    def sample_li(self, _ctx, light, frm, u, v, out):
        self.calls["sample_li"] += 1
        q = self._light_query(light, frm)
        q["u"], q["v"] = u, v
        s = self.shade.light_sample(q)[0]
        if not s["some"]:
            return 0
        o = out.contents
        o.direction.x, o.direction.y, o.direction.z = (float(x) for x in s["direction"])
        o.radiance.x, o.radiance.y, o.radiance.z = (float(x) for x in s["radiance"])
        o.distance, o.pdf = float(s["distance"]), float(s["pdf"])
        return 1

    def pdf_at_point(self, _ctx, light, frm, point):
        self.calls["pdf_at_point"] += 1
        q = self._light_query(light, frm)
        q["point"][0] = (point[0], point[1], point[2])
        return float(self.shade.light_pdf(q)[0])

    def escaped(self, _ctx, light, direction, radiance, pdf):
        self.calls["escaped"] += 1
        q = self._light_query(light, (0.0, 0.0, 0.0))
        q["point"][0] = (direction[0], direction[1], direction[2])
        s = self.shade.light_escaped(q)[0]
        if not s["some"]:
            return 0
        for k in range(3):
            radiance[k] = float(s["radiance"][k])
        pdf[0] = float(s["pdf"])
        return 1


# scene, width, height, spp, depth, the seam functions the scene must have exercised
CASES = [("cornellbox", 32, 32, 4, 8, ("intersect", "scatter", "emitted")),
         ("veach_mis", 40, 24, 4, 8, ("intersect", "occluded", "scatter", "eval", "emitted", "sample_li", "pdf_at_point")),
         ("openpbr_showcase", 40, 24, 4, 12, ("intersect", "occluded", "scatter", "eval", "emitted", "sample_li")),
         ("sun_sky", 40, 24, 4, 8, ("intersect", "occluded", "scatter", "eval", "sample_li", "escaped")),
         ("domelight", 40, 24, 2, 8, ("intersect", "occluded", "scatter", "eval", "sample_li", "escaped")),
         ("rectlight", 40, 24, 4, 4, ("intersect", "occluded", "scatter", "eval", "sample_li", "pdf_at_point")),
         ("light_visibility", 40, 24, 2, 4, ("intersect", "occluded", "scatter", "eval", "sample_li")),
         ("motionblur", 40, 24, 4, 4, ("intersect", "scatter")),
         ("instancing", 40, 24, 2, 8, ("intersect", "scatter")),
         ("nested_instancing", 40, 24, 2, 8, ("intersect", "scatter"))]


def check_against_own(name, host, o, spp, forward, must_call):
    """Render with the oracle on its own functions and on the host's seams: counters, call counts and image bits agree.
    -> (image, RayStats) of the hooked run."""
    own_img, own_st = o.render(spp, threads=1, forward=forward)
    img, st = host.render(o, spp, forward)
    for k in must_call:
        assert host.calls[k] > 0, (name, k, host.calls)
    for f, _t in ora.RayStats._fields_:
        assert getattr(st, f) == getattr(own_st, f), (name, f, getattr(st, f), getattr(own_st, f))
    assert host.calls["intersect"] == st.closest_hit and host.calls["occluded"] == st.shadow_rays, (name, host.calls)
    assert np.isfinite(img).all()
    bad = np.argwhere(img.view(np.uint32) != own_img.view(np.uint32))
    assert bad.shape[0] == 0, f"{name}: integrator on the hooked seams vs on the oracle's own: {bad.shape[0]} differ, {bad[:3]}"
    return img, st
