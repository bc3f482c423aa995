"""crust-render_amd — MI355X-native backend for crust-render's kernel seam.

Host-side mirror (Python; the reference's own toolchain, Rust, is absent from this image) of the
reference's operator interface over the C ABI in include/crt.h:

    crust_rt::Geometry / SceneBuilder / Scene / Ray / RayHit   crates/crust-rt/src/scene.rs:87-479
    crust_core::WorldBuilder / World                            crates/crust-core/src/rt_world.rs:89-294
    crust_core::Renderer / RenderSettings                       crates/crust-core/src/tracer.rs:137-735

Everything that computes runs in libcrt_amd.so (hand-written HIP for gfx950). There is no CPU fallback:
if the library is missing, or no HIP device is usable, the calls raise.

The directory name contains a hyphen, so the package is imported through `load()` in
__graft_entry__.py / tests/conftest.py under the module name `crust_render_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRT_AMD_LIB") or os.path.join(_HERE, "libcrt_amd.so")  # env: A/B builds only
CSRC = os.path.join(_HERE, "csrc")

MASK_CAMERA, MASK_SHADOW, MASK_INDIRECT, MASK_ALL = 1, 2, 4, 0xFFFFFFFF
INVALID_ID = 0xFFFFFFFF
INF = float("inf")

CRT_OK = 0
_ERRORS = {-1: "CRT_ERR_BAD_ARG", -2: "CRT_ERR_BAD_ID", -3: "CRT_ERR_NO_DEVICE", -4: "CRT_ERR_STACK",
           -5: "CRT_ERR_UNSUPPORTED", -6: "CRT_ERR_NO_MEMORY"}


class CrtError(RuntimeError):
    def __init__(self, code, what=""):
        detail = ""
        try:
            detail = _lib.crt_last_error().decode() if _lib is not None else ""
        except Exception:
            pass
        super().__init__(f"{what}: {_ERRORS.get(code, code)}" + (f" [{detail}]" if detail else ""))
        self.code = code


def build_native(force=False):
    """Compile libcrt_amd.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
    return LIB_PATH


# ------------------------------------------------------------------ C structs (include/crt.h)
class CrtRay(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("_pad0", C.c_float), ("dir", C.c_float * 3), ("_pad1", C.c_float),
                ("time", C.c_float), ("mask", C.c_uint32), ("_pad2", C.c_uint32 * 2)]


class CrtRayHit(C.Structure):
    _fields_ = [("t", C.c_float), ("normal", C.c_float * 3), ("front_face", C.c_uint32), ("u", C.c_float),
                ("v", C.c_float), ("geom_id", C.c_uint32), ("prim_id", C.c_uint32), ("_pad", C.c_uint32)]


class CrtTravStats(C.Structure):
    _fields_ = [("queries", C.c_uint64 * 2), ("nodes", C.c_uint64 * 2), ("leaves", C.c_uint64 * 2),
                ("packets", C.c_uint64 * 2), ("prims", C.c_uint64 * 2), ("accepted_hits", C.c_uint64),
                ("instance_descents", C.c_uint64), ("rays", C.c_uint64),
                ("phase_waves", C.c_uint64 * 8), ("phase_lanes", C.c_uint64 * 8),
                ("phase_cycles", C.c_uint64 * 8)]

    PHASES = ("loop", "fetch+setup", "node", "packet", "scalar prim", "instance exit", "emit", "f64 fallback")

    def utilisation(self):
        """Lane utilisation per traversal phase: {phase: (wave executions, live lanes / 64 per execution)}."""
        return {n: (int(self.phase_waves[k]), (self.phase_lanes[k] / (64.0 * self.phase_waves[k])) if self.phase_waves[k] else 0.0)
                for k, n in enumerate(self.PHASES)}

    def as_dict(self):
        d = {k: [int(getattr(self, k)[0]), int(getattr(self, k)[1])] for k in ("queries", "nodes", "leaves", "packets",
                                                                                "prims")}
        d.update(accepted_hits=int(self.accepted_hits), instance_descents=int(self.instance_descents),
                 rays=int(self.rays))
        return d

    def algorithmic_bytes(self):
        """SURVEY §8(d), verbatim: bytes(ray) = 48 (ray in) + 32 (hit out) + 128*nodes + 16*leaves + 192*packets
        + 128*accepted_hits + 112*instance_descents, summed over the rays of the launches counted."""
        n = sum
        return (80 * int(self.rays) + 128 * n(self.nodes) + 16 * n(self.leaves) + 192 * n(self.packets) +
                128 * int(self.accepted_hits) + 112 * int(self.instance_descents))


_MAT_FIELDS = [
    ("kind", C.c_uint32), ("thin_walled", C.c_uint32),
    ("base_weight", C.c_float), ("base_color", C.c_float * 3), ("base_diffuse_roughness", C.c_float),
    ("base_metalness", C.c_float),
    ("specular_weight", C.c_float), ("specular_color", C.c_float * 3), ("specular_roughness", C.c_float),
    ("specular_ior", C.c_float), ("specular_roughness_anisotropy", C.c_float),
    ("transmission_weight", C.c_float), ("transmission_color", C.c_float * 3), ("transmission_depth", C.c_float),
    ("transmission_scatter", C.c_float * 3), ("transmission_scatter_anisotropy", C.c_float),
    ("transmission_dispersion_scale", C.c_float), ("transmission_dispersion_abbe_number", C.c_float),
    ("subsurface_weight", C.c_float), ("subsurface_color", C.c_float * 3), ("subsurface_radius", C.c_float),
    ("subsurface_radius_scale", C.c_float * 3), ("subsurface_scatter_anisotropy", C.c_float),
    ("fuzz_weight", C.c_float), ("fuzz_color", C.c_float * 3), ("fuzz_roughness", C.c_float),
    ("coat_weight", C.c_float), ("coat_color", C.c_float * 3), ("coat_roughness", C.c_float),
    ("coat_roughness_anisotropy", C.c_float), ("coat_ior", C.c_float), ("coat_darkening", C.c_float),
    ("thin_film_weight", C.c_float), ("thin_film_thickness", C.c_float), ("thin_film_ior", C.c_float),
    ("emission_luminance", C.c_float), ("emission_color", C.c_float * 3),
    ("geometry_opacity", C.c_float),
]


class CrtMaterial(C.Structure):
    _fields_ = _MAT_FIELDS


MAT_OPENPBR, MAT_EMISSIVE = 0, 1
LIGHT_SPHERE, LIGHT_RECT, LIGHT_DISTANT, LIGHT_DOME = 0, 1, 2, 3
STRATEGY = {"power": 0, "mis": 0, "balance": 1, "light": 2, "bsdf": 3}
FILTER = {"box": 0, "triangle": 1}


class CrtLight(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("geom_id", C.c_uint32), ("radiance", C.c_float * 3),
                ("center", C.c_float * 3), ("radius", C.c_float), ("origin", C.c_float * 3),
                ("edge_u", C.c_float * 3), ("edge_v", C.c_float * 3), ("normal", C.c_float * 3)]


class CrtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("lens_radius", C.c_float)]


class CrtRenderSettings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32), ("frame", C.c_int32),
                ("strategy", C.c_int32), ("filter_kind", C.c_int32), ("filter_radius", C.c_float),
                ("variance_threshold", C.c_float), ("min_spp", C.c_uint32)]


class CrtRayStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("camera_rays", "closest_hit", "shadow_rays", "vertices", "rr_tested",
                                          "rr_killed", "ended_escaped", "ended_depth")]

    def total_rays(self):  # stats.rs:150-152
        return int(self.closest_hit) + int(self.shadow_rays)

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# Every symbol include/crt.h declares (checked by tests/test_abi.py against the header text).
ABI_SYMBOLS = [
    "crt_builder_new", "crt_builder_free", "crt_reserve", "crt_count", "crt_attach_triangles", "crt_attach_sphere",
    "crt_attach_instance", "crt_attach_empty", "crt_set_triangles", "crt_set_sphere", "crt_set_instance", "crt_commit",
    "crt_scene_retain", "crt_scene_release", "crt_scene_bounds", "crt_scene_geometry_count", "crt_scene_has_motion",
    "crt_scene_primitive_count", "crt_scene_primitive_breakdown", "crt_scene_unique_primitive_breakdown", "crt_scene_memory_footprint", "crt_scene_tree",
    "crt_shard_pixels", "crt_intersect1", "crt_occluded1", "crt_intersect_n", "crt_occluded_n", "crt_intersect_n_stats",
    "crt_occluded_n_stats", "crt_material_default", "crt_camera_new", "crt_renderer_new", "crt_renderer_free",
    "crt_renderer_pixel_count", "crt_renderer_pixel_indices", "crt_render_samples", "crt_film_resolve",
    "crt_film_read", "crt_film_clear", "crt_renderer_active_pixels", "crt_renderer_sample_counts", "crt_render_stats", "crt_renderer_profile", "crt_renderer_profile_read",
    "crt_render_samples_stats", "crt_version", "crt_last_error", "crt_device_info",
    "crt_scene_primitive_extents", "crt_scene_traversal_error", "crt_thread_release", "crt_renderer_shade_class_stats", "crt_renderer_pipeline", "crt_renderer_lanes", "crt_scene_image_check",
    "crt_scene_engine_select",
    "crt_material_scatter_n", "crt_material_eval_n", "crt_material_emitted_n", "crt_light_sample_n", "crt_light_pdf_n",
    "crt_light_escaped_n",
    "crt_shard_padded_count", "crt_gather_plan_new", "crt_gather_plan_free", "crt_gather_plan_padded_count",
    "crt_gather_plan_assemble", "crt_renderer_set_lanes",
]

_lib = None


def lib():
    """Loads libcrt_amd.so. Raises if the HIP extension has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950). "
                           "crust-render_amd has no CPU fallback.")
    # torch owns device memory and streams for this package; importing it first makes the process use ONE HIP
    # runtime (libcrt_amd.so's libamdhip64.so.7 dependency then binds to the copy torch already loaded).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    fp, up, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
    L.crt_version.restype = C.c_char_p
    L.crt_last_error.restype = C.c_char_p
    L.crt_device_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.crt_builder_new.restype = vp
    L.crt_builder_free.argtypes = [vp]
    L.crt_reserve.argtypes = [vp, C.c_size_t]
    L.crt_count.restype = C.c_size_t
    L.crt_count.argtypes = [vp]
    L.crt_attach_triangles.argtypes = [vp, fp, C.c_size_t, up, C.c_size_t, fp, C.c_size_t, C.c_uint32, up]
    L.crt_attach_sphere.argtypes = [vp, fp, C.c_float, C.c_uint32, up]
    L.crt_attach_instance.argtypes = [vp, vp, fp, fp, C.c_uint32, up]
    L.crt_attach_empty.argtypes = [vp, C.c_uint32, up]
    L.crt_set_triangles.argtypes = [vp, C.c_uint32, fp, C.c_size_t, up, C.c_size_t, fp, C.c_size_t]
    L.crt_set_sphere.argtypes = [vp, C.c_uint32, fp, C.c_float]
    L.crt_set_instance.argtypes = [vp, C.c_uint32, vp, fp, fp]
    L.crt_commit.restype = vp
    L.crt_commit.argtypes = [vp]
    L.crt_scene_retain.argtypes = [vp]
    L.crt_scene_release.argtypes = [vp]
    L.crt_scene_bounds.argtypes = [vp, fp]
    L.crt_scene_geometry_count.restype = C.c_uint32
    L.crt_scene_geometry_count.argtypes = [vp]
    L.crt_scene_has_motion.argtypes = [vp]
    L.crt_scene_primitive_count.restype = C.c_size_t
    L.crt_scene_primitive_count.argtypes = [vp]
    L.crt_scene_primitive_breakdown.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.crt_scene_primitive_extents.argtypes = [vp, C.POINTER(C.c_size_t), fp, fp, fp]
    L.crt_scene_traversal_error.argtypes = [vp, vp]
    L.crt_thread_release.restype = None
    L.crt_scene_unique_primitive_breakdown.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.crt_scene_memory_footprint.argtypes = [vp, C.POINTER(C.c_size_t)]
    if hasattr(L, "crt_scene_image_check"):  # absent from older A/B variant libraries
        L.crt_scene_image_check.argtypes = [vp, C.POINTER(C.c_uint64)]
    if hasattr(L, "crt_scene_engine_select"):  # absent from older A/B variant libraries
        L.crt_scene_engine_select.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32)]
    L.crt_scene_tree.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                 C.POINTER(up)]
    L.crt_intersect1.argtypes = [vp, C.POINTER(CrtRay), C.c_float, C.c_float, C.POINTER(CrtRayHit)]
    L.crt_occluded1.argtypes = [vp, C.POINTER(CrtRay), C.c_float, C.c_float]
    L.crt_intersect_n.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp]
    L.crt_occluded_n.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp]
    L.crt_intersect_n_stats.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp, C.POINTER(CrtTravStats)]
    L.crt_occluded_n_stats.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp, C.POINTER(CrtTravStats)]
    L.crt_material_default.argtypes = [C.POINTER(CrtMaterial)]
    L.crt_shard_pixels.restype = C.c_size_t
    L.crt_shard_pixels.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, up]
    if hasattr(L, "crt_renderer_new"):
        L.crt_camera_new.argtypes = [C.POINTER(CrtCamera), fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float]
        L.crt_renderer_new.restype = vp
        L.crt_renderer_new.argtypes = [vp, C.POINTER(CrtMaterial), C.c_size_t, C.POINTER(CrtLight), C.c_size_t,
                                       C.POINTER(CrtCamera), C.POINTER(CrtRenderSettings), C.c_uint32, C.c_uint32]
        L.crt_renderer_free.argtypes = [vp]
        L.crt_renderer_pixel_count.restype = C.c_size_t
        L.crt_renderer_pixel_count.argtypes = [vp]
        L.crt_renderer_pixel_indices.argtypes = [vp, up]
        L.crt_render_samples.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        L.crt_film_resolve.argtypes = [vp, vp, vp]
        L.crt_film_read.argtypes = [vp, fp]
        L.crt_film_clear.argtypes = [vp, vp]
        L.crt_render_stats.argtypes = [vp, C.POINTER(CrtRayStats)]
        L.crt_renderer_active_pixels.restype = C.c_size_t
        L.crt_renderer_active_pixels.argtypes = [vp]
        L.crt_renderer_sample_counts.argtypes = [vp, up]
        L.crt_renderer_profile.argtypes = [vp, C.c_int]
        if hasattr(L, "crt_renderer_lanes"):  # absent from older A/B variant libraries
            L.crt_renderer_lanes.argtypes = [vp]
            L.crt_renderer_lanes.restype = C.c_int
        if hasattr(L, "crt_renderer_set_lanes"):  # absent from older A/B variant libraries
            L.crt_renderer_set_lanes.argtypes = [vp, C.c_int]
        if hasattr(L, "crt_renderer_pipeline"):  # absent from older A/B variant libraries
            L.crt_renderer_pipeline.argtypes = [vp, C.POINTER(C.c_uint32)]
        if hasattr(L, "crt_renderer_shade_class_stats"):  # absent from older A/B variant libraries
            L.crt_renderer_shade_class_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.crt_renderer_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.crt_render_samples_stats.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.POINTER(CrtTravStats)]
    if hasattr(L, "crt_gather_plan_new"):
        L.crt_shard_padded_count.restype = C.c_size_t
        L.crt_shard_padded_count.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.crt_gather_plan_new.restype = vp
        L.crt_gather_plan_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.crt_gather_plan_free.argtypes = [vp]
        L.crt_gather_plan_padded_count.restype = C.c_size_t
        L.crt_gather_plan_padded_count.argtypes = [vp]
        L.crt_gather_plan_assemble.argtypes = [vp, vp, vp, vp]
    for name in ("crt_material_scatter_n", "crt_material_eval_n", "crt_material_emitted_n", "crt_light_sample_n",
                 "crt_light_pdf_n", "crt_light_escaped_n"):  # the shading seam (shading.py)
        if hasattr(L, name):
            getattr(L, name).argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp]
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        raise CrtError(rc, what)
    return rc


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


IDENTITY12 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float32)


def affine(m3=None, t=(0, 0, 0)):
    """glam Affine3A as 12 floats: matrix3 columns x, y, z then translation."""
    m = np.eye(3, dtype=np.float32) if m3 is None else np.asarray(m3, dtype=np.float32)
    return np.concatenate([m[:, 0], m[:, 1], m[:, 2], np.asarray(t, dtype=np.float32)]).astype(np.float32)


# ------------------------------------------------------------------ crust_rt mirror
class Ray:
    """crust_rt::Ray (ray.rs:18-55)."""

    def __init__(self, origin, dir, time=0.0, mask=MASK_ALL):
        self.origin, self.dir, self.time, self.mask = origin, dir, time, mask

    def c(self):
        r = CrtRay()
        r.origin[:] = [float(x) for x in self.origin]
        r.dir[:] = [float(x) for x in self.dir]
        r.time = self.time
        r.mask = self.mask
        return r


RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("_p0", np.float32), ("dir", np.float32, 3), ("_p1", np.float32),
                      ("time", np.float32), ("mask", np.uint32), ("_p2", np.uint32, 2)])
HIT_DTYPE = np.dtype([("t", np.float32), ("normal", np.float32, 3), ("front_face", np.uint32), ("u", np.float32),
                      ("v", np.float32), ("geom_id", np.uint32), ("prim_id", np.uint32), ("_pad", np.uint32)])
assert RAY_DTYPE.itemsize == 48 and HIT_DTYPE.itemsize == 40


def pack_rays(rays8):
    """[n, 8] float32 (o, d, time, mask bits) -> CrtRay records."""
    rays8 = np.ascontiguousarray(rays8, dtype=np.float32).reshape(-1, 8)
    out = np.zeros(rays8.shape[0], dtype=RAY_DTYPE)
    out["origin"] = rays8[:, 0:3]
    out["dir"] = rays8[:, 3:6]
    out["time"] = rays8[:, 6]
    out["mask"] = rays8[:, 7].view(np.uint32)
    return out


class Scene:
    """crust_rt::Scene (scene.rs:345-479): immutable, ref-counted."""

    def __init__(self, handle, keep=()):
        if not handle:
            raise RuntimeError("crt_commit failed: " + lib().crt_last_error().decode())
        self.h = handle
        self._keep = list(keep)

    def __del__(self):
        try:
            if self.h:
                lib().crt_scene_release(self.h)
        except Exception:
            pass

    def intersect(self, ray, t_min, t_max):
        h = CrtRayHit()
        rc = _check(lib().crt_intersect1(self.h, C.byref(ray.c()), t_min, t_max, C.byref(h)), "crt_intersect1")
        return h if rc == 1 else None

    def occluded(self, ray, t_min, t_max):
        return _check(lib().crt_occluded1(self.h, C.byref(ray.c()), t_min, t_max), "crt_occluded1") == 1

    def bounds(self):
        out = np.zeros(6, dtype=np.float32)
        return out if _check(lib().crt_scene_bounds(self.h, _fp(out)), "crt_scene_bounds") == 1 else None

    def geometry_count(self):
        return lib().crt_scene_geometry_count(self.h)

    def has_motion(self):
        return bool(lib().crt_scene_has_motion(self.h))

    def primitive_count(self):
        return lib().crt_scene_primitive_count(self.h)

    def primitive_breakdown(self):
        out = (C.c_size_t * 5)()
        _check(lib().crt_scene_primitive_breakdown(self.h, out), "crt_scene_primitive_breakdown")
        return dict(zip(("triangles", "spheres", "curve_segments", "cubic_curve_spans", "instances"), map(int, out)))

    def primitive_extents(self):
        """Scene::primitive_extents (scene.rs:446-455) -> (count, scene diagonal, mean, max primitive diagonal)."""
        n, d, mean, mx = C.c_size_t(), C.c_float(), C.c_float(), C.c_float()
        _check(lib().crt_scene_primitive_extents(self.h, C.byref(n), C.byref(d), C.byref(mean), C.byref(mx)),
               "crt_scene_primitive_extents")
        return n.value, d.value, mean.value, mx.value

    def traversal_error(self, stream=None):
        """Drains `stream`, reads and clears the scene's traversal error word: raises CrtError(CRT_ERR_STACK) if a
        batched launch since the last call overflowed a traversal stack."""
        _check(lib().crt_scene_traversal_error(self.h, _stream_ptr(stream)), "crt_scene_traversal_error")

    def unique_primitive_breakdown(self):
        """Scene::unique_primitive_breakdown (scene.rs:422-427): what is resident, shared prototypes once."""
        out = (C.c_size_t * 5)()
        _check(lib().crt_scene_unique_primitive_breakdown(self.h, out), "crt_scene_unique_primitive_breakdown")
        return dict(zip(("triangles", "spheres", "curve_segments", "cubic_curve_spans", "instances"), map(int, out)))

    def image_check(self):
        """Host-only self-check of the device image (crt.h, crt_scene_image_check): counts, or CrtError naming the
        broken invariant. Needs no GPU."""
        out = (C.c_uint64 * 8)()
        _check(lib().crt_scene_image_check(self.h, out), "crt_scene_image_check")
        return dict(zip(("nodes", "leaf_words_plain", "leaf_words_direct_index", "leaf_words_direct_instance", "instances",
                         "moving_instances", "staged_roots", "direct_leaves"), map(int, out)))

    def engine_select(self, want_wide=-1):
        """Which traversal-engine instance the library selects for this scene's image (crt.h, crt_scene_engine_select;
        host-only). want_wide: -1 the scene's preference, 0 / 1 asked for (CrtError CRT_ERR_UNSUPPORTED when the image
        cannot be decoded by it), -2 what a launch in this process would pick (CRT_WIDE included)."""
        out = (C.c_uint32 * 8)()
        _check(lib().crt_scene_engine_select(self.h, int(want_wide), out), "crt_scene_engine_select")
        return dict(zip(("wide", "direct", "lds_stack", "window", "ext_cold", "path_cold", "direct_words", "cold"),
                        map(int, out)))

    def memory_footprint(self):
        out = (C.c_size_t * 6)()
        _check(lib().crt_scene_memory_footprint(self.h, out), "crt_scene_memory_footprint")
        return dict(zip(("prim_nodes", "boxed_prims", "bvh_nodes", "leaves", "packets", "indices"), map(int, out)))

    def tree(self):
        """Host copies of the committed tree as uint32 words: nodes[n,32], leaves[n,4], packets[n,48], indices[n]."""
        counts = (C.c_size_t * 5)()
        pn, pl, pp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        pi = C.POINTER(C.c_uint32)()
        _check(lib().crt_scene_tree(self.h, counts, C.byref(pn), C.byref(pl), C.byref(pp), C.byref(pi)),
               "crt_scene_tree")

        def grab(ptr, n, words):
            if n == 0:
                return np.zeros((0, words), dtype=np.uint32)
            buf = C.cast(ptr, C.POINTER(C.c_uint32 * (n * words))).contents
            return np.frombuffer(buf, dtype=np.uint32).reshape(n, words).copy()

        return (grab(pn, counts[0], 32), grab(pl, counts[1], 4), grab(pp, counts[2], 48),
                grab(pi, counts[3], 1).reshape(-1), dict(nodes=counts[0], leaves=counts[1], packets=counts[2],
                                                         indices=counts[3], prims=counts[4]))

    # ---- batched device queries (torch tensors hold the HBM buffers) ----
    def intersect_n(self, d_rays, t_min, t_max, d_hits=None, stream=None, stats=None):
        """d_rays: torch uint8/any tensor on cuda holding n CrtRay records (48 B each). Returns d_hits (n*40 B)."""
        import torch
        n = d_rays.numel() * d_rays.element_size() // 48
        if d_hits is None:
            d_hits = torch.empty(n * 40, dtype=torch.uint8, device=d_rays.device)
        sp = _stream_ptr(stream)
        if stats is None:
            _check(lib().crt_intersect_n(self.h, d_rays.data_ptr(), n, t_min, t_max, d_hits.data_ptr(), sp),
                   "crt_intersect_n")
        else:
            _check(lib().crt_intersect_n_stats(self.h, d_rays.data_ptr(), n, t_min, t_max, d_hits.data_ptr(), sp,
                                               C.byref(stats)), "crt_intersect_n_stats")
        return d_hits

    def occluded_n(self, d_rays, t_min, t_max, d_out=None, stream=None, stats=None):
        import torch
        n = d_rays.numel() * d_rays.element_size() // 48
        if d_out is None:
            d_out = torch.empty(n, dtype=torch.int32, device=d_rays.device)
        sp = _stream_ptr(stream)
        if stats is None:
            _check(lib().crt_occluded_n(self.h, d_rays.data_ptr(), n, t_min, t_max, d_out.data_ptr(), sp),
                   "crt_occluded_n")
        else:
            _check(lib().crt_occluded_n_stats(self.h, d_rays.data_ptr(), n, t_min, t_max, d_out.data_ptr(), sp,
                                              C.byref(stats)), "crt_occluded_n_stats")
        return d_out


def _stream_ptr(stream):
    """HIP stream handle for the C ABI. Defaults to torch's current stream so launches order with torch ops."""
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


def rays_to_device(rays8, device="cuda:0"):
    import torch
    rec = pack_rays(rays8)
    return torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(device)


def hits_to_host(d_hits):
    return d_hits.cpu().numpy().view(HIT_DTYPE)


class SceneBuilder:
    """crust_rt::SceneBuilder (scene.rs:147-342). Geometry variants are the attach_* methods
    (Geometry::TriangleMesh / Sphere / Instance, scene.rs:87-122)."""

    def __init__(self):
        self.h = lib().crt_builder_new()
        self._keep = []

    def __del__(self):
        try:
            if self.h:
                lib().crt_builder_free(self.h)
        except Exception:
            pass

    def reserve(self, additional):
        _check(lib().crt_reserve(self.h, additional), "crt_reserve")

    def count(self):
        return lib().crt_count(self.h)

    @staticmethod
    def _mesh(verts, idx, normals):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        return verts, idx, nrm

    def attach_triangles(self, verts, idx, normals=None, mask=MASK_ALL):
        v, i, n = self._mesh(verts, idx, normals)
        gid = C.c_uint32()
        _check(lib().crt_attach_triangles(self.h, _fp(v), v.shape[0], _up(i), i.shape[0], _fp(n),
                                          0 if n is None else n.shape[0], mask, C.byref(gid)), "crt_attach_triangles")
        return gid.value

    def attach_sphere(self, center, radius, mask=MASK_ALL):
        c = np.asarray(center, dtype=np.float32)
        gid = C.c_uint32()
        _check(lib().crt_attach_sphere(self.h, _fp(c), radius, mask, C.byref(gid)), "crt_attach_sphere")
        return gid.value

    def attach_instance(self, scene, l2w=IDENTITY12, l2w_end=None, mask=MASK_ALL):
        a = np.ascontiguousarray(l2w, dtype=np.float32)
        e = None if l2w_end is None else np.ascontiguousarray(l2w_end, dtype=np.float32)
        gid = C.c_uint32()
        _check(lib().crt_attach_instance(self.h, scene.h, _fp(a), _fp(e), mask, C.byref(gid)), "crt_attach_instance")
        self._keep.append(scene)
        return gid.value

    def attach_empty(self, mask=MASK_ALL):
        gid = C.c_uint32()
        _check(lib().crt_attach_empty(self.h, mask, C.byref(gid)), "crt_attach_empty")
        return gid.value

    def set_triangles(self, gid, verts, idx, normals=None):
        v, i, n = self._mesh(verts, idx, normals)
        _check(lib().crt_set_triangles(self.h, gid, _fp(v), v.shape[0], _up(i), i.shape[0], _fp(n),
                                       0 if n is None else n.shape[0]), "crt_set_triangles")

    def set_sphere(self, gid, center, radius):
        c = np.asarray(center, dtype=np.float32)
        _check(lib().crt_set_sphere(self.h, gid, _fp(c), radius), "crt_set_sphere")

    def set_instance(self, gid, scene, l2w=IDENTITY12, l2w_end=None):
        a = np.ascontiguousarray(l2w, dtype=np.float32)
        e = None if l2w_end is None else np.ascontiguousarray(l2w_end, dtype=np.float32)
        _check(lib().crt_set_instance(self.h, gid, scene.h, _fp(a), _fp(e)), "crt_set_instance")
        self._keep.append(scene)

    def commit(self):
        h, self.h = self.h, None
        return Scene(lib().crt_commit(h), self._keep)


def default_material():
    m = CrtMaterial()
    lib().crt_material_default(C.byref(m))
    return m


def diffuse_material(rgb):  # OpenPBR::diffuse, openpbr.rs:177-183
    m = default_material()
    m.base_color[:] = [float(x) for x in rgb]
    m.specular_weight = 0.0
    return m


def emissive_material(rgb):  # Emissive::new, emissive.rs:16-19
    m = default_material()
    m.kind = MAT_EMISSIVE
    m.emission_color[:] = [float(x) for x in rgb]
    m.emission_luminance = 1.0
    return m


# ------------------------------------------------------------------ crust_core mirror: Renderer
class RenderSettings:
    """crust_core::RenderSettings (tracer.rs:640-735) for the wavefront path."""

    def __init__(self, width, height, max_depth=32, frame=0, strategy="power", pixel_filter="triangle",
                 filter_radius=None, variance_threshold=0.0, min_spp=32):
        self.width, self.height, self.max_depth, self.frame = int(width), int(height), int(max_depth), int(frame)
        self.strategy, self.pixel_filter = strategy, pixel_filter
        self.filter_radius = filter_radius if filter_radius is not None else (0.5 if pixel_filter == "box" else 1.0)
        self.variance_threshold = variance_threshold  # > 0: render_pixel's adaptive early stop (tracer.rs:609-617)
        self.min_spp = int(min_spp)

    def c(self):
        return CrtRenderSettings(self.width, self.height, self.max_depth, self.frame, STRATEGY[self.strategy],
                                 FILTER[self.pixel_filter], float(self.filter_radius), float(self.variance_threshold),
                                 self.min_spp)


def make_camera(lookfrom, lookat, vup, vfov_deg, aspect, aperture, focus_dist):
    """Camera::new (camera.rs:27-63)."""
    c = CrtCamera()
    a = [np.asarray(x, dtype=np.float32) for x in (lookfrom, lookat, vup)]
    lib().crt_camera_new(C.byref(c), _fp(a[0]), _fp(a[1]), _fp(a[2]), float(vfov_deg), float(aspect), float(aperture),
                         float(focus_dist))
    return c


def make_lights(light_dicts):
    arr = (CrtLight * max(len(light_dicts), 1))()
    for k, d in enumerate(light_dicts):
        l = arr[k]
        l.kind = {"sphere": LIGHT_SPHERE, "rect": LIGHT_RECT, "distant": LIGHT_DISTANT,
                  "dome": LIGHT_DOME}[d["kind"]]
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


class Renderer:
    """crust_core::Renderer (tracer.rs:137-213) over the wavefront kernels. `rank`/`world` select this
    process's share of the 16x16 pixel tiles (round-robin), for one-process-per-GPU sharding."""

    def __init__(self, scene, materials, lights, camera, settings, rank=0, world=1):
        self.scene = scene
        n = len(materials)
        self._mats = (CrtMaterial * max(n, 1))(*materials)
        self._lights = lights if not isinstance(lights, (list, tuple)) else make_lights(lights)
        self.n_lights = len(lights)
        self.settings = settings
        cs = settings.c()
        self.h = lib().crt_renderer_new(scene.h, self._mats, n, self._lights, self.n_lights, C.byref(camera),
                                        C.byref(cs), rank, world)
        if not self.h:
            raise CrtError(-1, "crt_renderer_new")
        self.n_pix = lib().crt_renderer_pixel_count(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().crt_renderer_free(self.h)
        except Exception:
            pass

    def pixel_indices(self):
        out = np.zeros(self.n_pix, dtype=np.uint32)
        _check(lib().crt_renderer_pixel_indices(self.h, _up(out)), "crt_renderer_pixel_indices")
        return out

    def render_samples(self, sample_begin, sample_count, stream=None):
        _check(lib().crt_render_samples(self.h, sample_begin, sample_count, _stream_ptr(stream)), "crt_render_samples")

    def render_samples_stats(self, sample_begin, sample_count, stream=None):
        st = (CrtTravStats * 2)()
        _check(lib().crt_render_samples_stats(self.h, sample_begin, sample_count, _stream_ptr(stream), st),
               "crt_render_samples_stats")
        return st[0], st[1]  # extend (closest-hit) launches, shadow (any-hit) launches

    def film_to(self, d_rgb, stream=None):
        _check(lib().crt_film_resolve(self.h, d_rgb.data_ptr(), _stream_ptr(stream)), "crt_film_resolve")

    def film(self):
        """Owned pixels' means, [n_pix, 3] float32, in pixel_indices() order."""
        out = np.zeros((self.n_pix, 3), dtype=np.float32)
        _check(lib().crt_film_read(self.h, _fp(out)), "crt_film_read")
        return out

    def image(self):
        """Full frame [height, width, 3] in BUFFER order (row 0 = bottom, buffer.rs:45-49); unowned pixels 0."""
        img = np.zeros((self.settings.height * self.settings.width, 3), dtype=np.float32)
        img[self.pixel_indices()] = self.film()
        return img.reshape(self.settings.height, self.settings.width, 3)

    def clear(self, stream=None):
        _check(lib().crt_film_clear(self.h, _stream_ptr(stream)), "crt_film_clear")

    def stats(self):
        s = CrtRayStats()
        _check(lib().crt_render_stats(self.h, C.byref(s)), "crt_render_stats")
        return s

    def active_pixels(self):
        """Adaptive stopping: owned pixels still sampling."""
        return int(lib().crt_renderer_active_pixels(self.h))

    def sample_counts(self):
        """Adaptive stopping: samples taken per owned pixel (pixel_indices order)."""
        out = np.zeros(self.n_pix, dtype=np.uint32)
        _check(lib().crt_renderer_sample_counts(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))), "crt_renderer_sample_counts")
        return out

    def render_adaptive(self, spp, first=None, batch=4, stream=None):
        """Renderer::render_pass with adaptive stopping (tracer.rs:515-636): `first` samples (default min_spp rounded
        up to a multiple of 4) for every pixel, then batches of `batch` for the pixels still sampling, up to `spp`.
        With batch == 4 the ray counters equal the reference's: the rule can only fire on multiples of 4."""
        first = first or -(-max(self.settings.min_spp, 2) // 4) * 4
        s = 0
        while s < spp and self.active_pixels() > 0:
            n = min(first if s == 0 else batch, spp - s)
            self.render_samples(s, n, stream)
            s += n
        return s

    SHADE_CLASSES = ("emissive / escaped", "base", "coat / fuzz / thin film", "transmissive / subsurface")

    def shade_class_stats(self, enable=None):
        """Material-class lane utilisation of the vertex step (crt_renderer_shade_class_stats): enable=True/False
        switches the counting; returns {class: (wave executions containing it, lanes holding it / 64 per execution)}
        since the last call and clears the counts."""
        w, l = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
        _check(lib().crt_renderer_shade_class_stats(self.h, -1 if enable is None else int(bool(enable)), w, l),
               "crt_renderer_shade_class_stats")
        return {n: (int(w[k]), (l[k] / (64.0 * w[k])) if w[k] else 0.0) for k, n in enumerate(self.SHADE_CLASSES)}

    def pipeline(self):
        """{'fused', 'wide', 'grid'}: the launch pipeline chosen for this scene (crt.h, crt_renderer_pipeline)."""
        out = (C.c_uint32 * 3)()
        _check(lib().crt_renderer_pipeline(self.h, out), "crt_renderer_pipeline")
        return dict(fused=bool(out[0]), wide=bool(out[1]), grid=int(out[2]))

    def lanes(self):
        """Sub-batches (own buffers, own HIP stream) the last batch ran as (crt.h, crt_renderer_lanes)."""
        return int(lib().crt_renderer_lanes(self.h)) if hasattr(lib(), "crt_renderer_lanes") else 1

    def set_lanes(self, lanes):
        """Lanes later batches may run as (1..4; crt.h, crt_renderer_set_lanes). Returns the count set."""
        return _check(lib().crt_renderer_set_lanes(self.h, int(lanes)), "crt_renderer_set_lanes")

    def profile(self, enable=True):
        _check(lib().crt_renderer_profile(self.h, 1 if enable else 0), "crt_renderer_profile")

    def profile_read(self):
        ms = (C.c_double * 4)()
        n = (C.c_uint64 * 4)()
        _check(lib().crt_renderer_profile_read(self.h, ms, n), "crt_renderer_profile_read")
        names = ("extend", "shade", "shadow", "other")
        return {names[k]: dict(ms=float(ms[k]), launches=int(n[k])) for k in range(4)}


def scene_path(name):
    """scenes/<name> with the extension the file has: .usda (text), .usd (binary crate), .usda.xz (packed text)."""
    import os
    if str(name).startswith("synthetic:") or os.path.exists(name):
        return name
    base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", name)
    for ext in (".usda", ".usd", ".usda.xz"):
        if os.path.exists(base + ext):
            return base + ext
    raise FileNotFoundError("no scene %r under scenes/" % name)


def load_usda(path, width=None, height=None, max_depth=None, rank=0, world=1, variance=0.0, min_spp=None, dist=None,
              timings=None):
    """Scene::from_usd (scene.rs / usd_import.rs:287-424) for the text sample scenes -> (Renderer, desc).
    `path` may also name a synthetic scene: "synthetic:city" or "synthetic:city:<side>" (synthetic.py).
    variance > 0 enables adaptive stopping (the scene files' own default is 0.05; 0 = every sample, the rule for
    comparable runs, scripts/check_images.sh:5-11). dist: the job's torch.distributed module — the file is then imported
    by rank 0 only and broadcast (shard.import_once). timings: a dict that receives `import_s` (this rank's share of the
    import: parsing on rank 0, waiting for the broadcast elsewhere) and `commit_s` (build_world: every rank's own commit)."""
    from . import usda, shard
    import sys
    import time
    t0 = time.perf_counter()
    if str(path).startswith("synthetic:"):
        from . import synthetic
        parts = str(path).split(":")
        kw = dict(side=int(parts[2])) if len(parts) > 2 else {}
        desc = getattr(synthetic, parts[1])(width or 640, height or 360, **kw)
    else:
        desc = shard.import_once(path, width, height, dist)
    me = sys.modules[__name__]
    t1 = time.perf_counter()
    scene, materials, protos = usda.build_world(desc, me, default_material)
    t2 = time.perf_counter()
    if timings is not None:
        timings.update(import_s=t1 - t0, commit_s=t2 - t1)
    s = desc.settings
    settings = RenderSettings(s["width"], s["height"], s["max_depth"] if max_depth is None else max_depth, s["frame"],
                              s["strategy"], s["filter"], s["filter_radius"], float(variance),
                              s["min_spp"] if min_spp is None else min_spp)
    cam = make_camera(**desc.camera)
    r = Renderer(scene, materials, desc.lights, cam, settings, rank, world)
    r._protos = protos
    return r, desc

def render_with_report(path, spp=None, out_exr=None, width=None, height=None, max_depth=None, variance=None,
                       batch=16):
    """The reference binary's run in one call (main.rs:491-648): load, commit, render, optionally write the EXR, and
    return (image, RenderStats) — the report carries the phases the reference prints ("Parse USD stage",
    "Commit acceleration structure", "Render", "Write image") and its scene / ray statistics blocks.
    spp / variance default to the scene's own RenderSettings (adaptive stopping on, as the binary renders)."""
    import sys
    from . import stats as _stats
    st = _stats.RenderStats()
    me = sys.modules[__name__]
    with st.phase("Parse USD stage"):
        desc = usda.load(path, width, height)
    s = desc.settings
    with st.phase("Commit acceleration structure"):
        scene, materials, protos = usda.build_world(desc, me, default_material)
        scene.memory_footprint()  # uploads the device image: part of the commit, not of the render
    var = s["variance"] if variance is None else variance
    settings = RenderSettings(s["width"], s["height"], s["max_depth"] if max_depth is None else max_depth, s["frame"],
                              s["strategy"], s["filter"], s["filter_radius"], float(var), s["min_spp"])
    r = Renderer(scene, materials, desc.lights, make_camera(**desc.camera), settings)
    r._protos = protos
    n = int(spp if spp is not None else s["spp"])
    with st.phase("Render"):
        if var > 0:
            r.render_adaptive(n, batch=batch)
        else:
            for b in range(0, n, 32):
                r.render_samples(b, min(32, n - b))
        img = r.image()  # synchronises: the film read-back ends the phase
    if out_exr:
        with st.phase("Write image"):
            exr.write_exr(out_exr, img[::-1])  # film buffer rows run bottom to top (buffer.rs:45-49)
    _stats.for_render(scene, r, n, st)
    return img, st


load_scene = load_usda  # the same loader under the name that fits all it reads (.usda, .usd/.usdc crates, synthetic:*)

from . import usda  # noqa: E402,F401  (the minimal USDA reader)
from . import usdc  # noqa: E402,F401  (the USDC crate reader)
from . import shard  # noqa: E402,F401  (pixel-tile sharding + the tile gather)
from . import exr  # noqa: E402,F401  (EXR writer / reader + the reference's exr_diff metrics)
from . import stats  # noqa: E402,F401  (RenderStats + the reference's report layout)
from . import shading  # noqa: E402,F401  (Material / Light as batched device functions: the shading seam)
from . import synthetic  # noqa: E402,F401  (scenes built in code: the labelled stand-in for BASELINE config 5)
