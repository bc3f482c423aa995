"""ctypes bindings for the ORACLE (oracle/_build/liboracle.so) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

MASK_CAMERA, MASK_SHADOW, MASK_INDIRECT, MASK_ALL = 1, 2, 4, 0xFFFFFFFF
INVALID_ID = 0xFFFFFFFF
INF = float("inf")


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)

    def np(self):
        return np.array([self.x, self.y, self.z], dtype=np.float32)


def v3(a):
    return V3(float(a[0]), float(a[1]), float(a[2]))


class Ray(C.Structure):
    _fields_ = [("origin", V3), ("dir", V3), ("time", C.c_float), ("mask", C.c_uint32)]


def ray(o, d, time=0.0, mask=MASK_ALL):
    return Ray(v3(o), v3(d), time, mask)


class RayHit(C.Structure):
    _fields_ = [("t", C.c_float), ("normal", V3), ("front_face", C.c_int), ("u", C.c_float), ("v", C.c_float),
                ("geom_id", C.c_uint32), ("prim_id", C.c_uint32)]


class TravStats(C.Structure):
    _fields_ = [("queries", C.c_uint64 * 2), ("nodes", C.c_uint64 * 2), ("leaves", C.c_uint64 * 2),
                ("packets", C.c_uint64 * 2), ("prims", C.c_uint64 * 2), ("accepted_hits", C.c_uint64),
                ("instance_descents", C.c_uint64), ("stack_high_water", C.c_uint64)]


class WideNode(C.Structure):
    _fields_ = [("bmin", (C.c_float * 4) * 3), ("bmax", (C.c_float * 4) * 3), ("child", C.c_uint32 * 4),
                ("flags", C.c_uint32), ("pad", C.c_uint32 * 3)]


class Leaf(C.Structure):
    _fields_ = [("pkt_first", C.c_uint32), ("pkt_count", C.c_uint32), ("idx_first", C.c_uint32),
                ("idx_count", C.c_uint32)]


class Tri4(C.Structure):
    _fields_ = [("v", ((C.c_float * 4) * 3) * 3), ("prim", C.c_uint32 * 4), ("active", C.c_uint32),
                ("mask_and", C.c_uint32), ("mask_or", C.c_uint32), ("masks", C.c_uint32 * 4), ("pad", C.c_uint32)]


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


class Material(C.Structure):
    _fields_ = _MAT_FIELDS


MAT_OPENPBR, MAT_EMISSIVE = 0, 1


class HitRecord(C.Structure):
    _fields_ = [("p", V3), ("normal", V3), ("t", C.c_float), ("front_face", C.c_int)]


class Sampler(C.Structure):
    _fields_ = [("pattern", C.c_uint32), ("index", C.c_uint32)]


class Scatter(C.Structure):
    _fields_ = [("origin", V3), ("dir", V3), ("value", V3), ("pdf", C.c_float), ("delta", C.c_int),
                ("medium", C.c_int)]


class Medium(C.Structure):
    _fields_ = [("sigma_a", V3), ("sigma_s", V3), ("g", C.c_float)]


LIGHT_SPHERE, LIGHT_RECT, LIGHT_DISTANT, LIGHT_DOME = 0, 1, 2, 3


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("geom_id", C.c_uint32), ("radiance", C.c_float * 3),
                ("center", C.c_float * 3), ("radius", C.c_float), ("origin", C.c_float * 3),
                ("edge_u", C.c_float * 3), ("edge_v", C.c_float * 3), ("normal", C.c_float * 3)]


class LightSample(C.Structure):
    _fields_ = [("direction", V3), ("distance", C.c_float), ("radiance", V3), ("pdf", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("origin", V3), ("lower_left", V3), ("horizontal", V3), ("vertical", V3), ("u", V3), ("v", V3),
                ("lens_radius", C.c_float)]


class RayStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("camera_rays", "closest_hit", "shadow_rays", "vertices", "rr_tested",
                                          "rr_killed", "ended_escaped", "ended_depth")]

    def total_rays(self):
        return self.closest_hit + self.shadow_rays


class RenderJob(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_uint32), ("camera", Camera),
                ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("min_spp", C.c_uint32), ("variance_threshold", C.c_float), ("frame", C.c_int32),
                ("strategy", C.c_int32), ("filter_kind", C.c_int32), ("filter_radius", C.c_float),
                ("forward", C.c_int32)]


_fpp, _u32 = C.POINTER(C.c_float), C.c_uint32


class SeamHooks(C.Structure):
    """OraSeamHooks (ora_pt.h): the integrator's calls across the kernel seam and the Material / Light traits."""
    INTERSECT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Ray), C.c_float, C.c_float, C.POINTER(RayHit))
    OCCLUDED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Ray), C.c_float, C.c_float)
    MAT_SCATTER = C.CFUNCTYPE(C.c_int, C.c_void_p, _u32, _fpp, C.POINTER(HitRecord), _u32, _u32, C.POINTER(Scatter))
    MAT_EVAL = C.CFUNCTYPE(C.c_int, C.c_void_p, _u32, _fpp, C.POINTER(HitRecord), _fpp, _fpp, _fpp)
    MAT_EMITTED = C.CFUNCTYPE(None, C.c_void_p, _u32, C.c_float, _fpp)
    LIGHT_SAMPLE = C.CFUNCTYPE(C.c_int, C.c_void_p, _u32, _fpp, C.c_float, C.c_float, C.POINTER(LightSample))
    LIGHT_PDF = C.CFUNCTYPE(C.c_float, C.c_void_p, _u32, _fpp, _fpp)
    LIGHT_ESCAPED = C.CFUNCTYPE(C.c_int, C.c_void_p, _u32, _fpp, _fpp, _fpp)
    _fields_ = [("ctx", C.c_void_p), ("intersect", INTERSECT), ("occluded", OCCLUDED), ("mat_scatter", MAT_SCATTER),
                ("mat_eval", MAT_EVAL), ("mat_emitted", MAT_EMITTED), ("light_sample", LIGHT_SAMPLE),
                ("light_pdf", LIGHT_PDF), ("light_escaped", LIGHT_ESCAPED)]


def set_seam_hooks(hooks):
    """hooks: a SeamHooks (keep it alive while set) or None."""
    lib().ora_set_seam_hooks(C.byref(hooks) if hooks is not None else None)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    up = C.POINTER(C.c_uint32)
    L.ora_builder_new.restype = C.c_void_p
    L.ora_attach_triangles.restype = C.c_uint32
    L.ora_attach_triangles.argtypes = [C.c_void_p, fp, C.c_size_t, up, C.c_size_t, fp, C.c_size_t, C.c_uint32]
    L.ora_attach_sphere.restype = C.c_uint32
    L.ora_attach_sphere.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32]
    L.ora_attach_instance.restype = C.c_uint32
    L.ora_attach_instance.argtypes = [C.c_void_p, C.c_void_p, fp, fp, C.c_uint32]
    L.ora_attach_empty.restype = C.c_uint32
    L.ora_attach_empty.argtypes = [C.c_void_p, C.c_uint32]
    L.ora_set_triangles.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_size_t, up, C.c_size_t, fp, C.c_size_t]
    L.ora_set_sphere.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float]
    L.ora_set_instance.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, fp, fp]
    L.ora_builder_count.restype = C.c_size_t
    L.ora_builder_count.argtypes = [C.c_void_p]
    L.ora_commit.restype = C.c_void_p
    L.ora_commit.argtypes = [C.c_void_p]
    L.ora_scene_free.argtypes = [C.c_void_p]
    L.ora_intersect.argtypes = [C.c_void_p, C.POINTER(Ray), C.c_float, C.c_float, C.POINTER(RayHit)]
    L.ora_occluded.argtypes = [C.c_void_p, C.POINTER(Ray), C.c_float, C.c_float]
    L.ora_linear_scan.argtypes = [C.c_void_p, C.POINTER(Ray), C.c_float, C.c_float, C.POINTER(RayHit)]
    L.ora_scene_bounds.argtypes = [C.c_void_p, fp]
    L.ora_geometry_count.restype = C.c_uint32
    L.ora_geometry_count.argtypes = [C.c_void_p]
    L.ora_has_motion.argtypes = [C.c_void_p]
    L.ora_primitive_count.restype = C.c_size_t
    L.ora_primitive_count.argtypes = [C.c_void_p]
    L.ora_primitive_breakdown.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.ora_primitive_extents.restype = C.c_size_t
    L.ora_primitive_extents.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.ora_unique_primitive_breakdown.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.ora_intersect_n.argtypes = [C.c_void_p, fp, C.c_size_t, C.c_float, C.c_float, fp, up, C.POINTER(C.c_uint8)]
    L.ora_occluded_n.argtypes = [C.c_void_p, fp, C.c_size_t, C.c_float, C.c_float, C.POINTER(C.c_uint8)]
    L.ora_set_trav_stats.argtypes = [C.POINTER(TravStats)]
    L.ora_bvh_counts.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.ora_bvh_nodes.restype = C.POINTER(WideNode)
    L.ora_bvh_nodes.argtypes = [C.c_void_p]
    L.ora_bvh_leaves.restype = C.POINTER(Leaf)
    L.ora_bvh_leaves.argtypes = [C.c_void_p]
    L.ora_bvh_packets.restype = C.POINTER(Tri4)
    L.ora_bvh_packets.argtypes = [C.c_void_p]
    L.ora_bvh_indices.restype = up
    L.ora_bvh_indices.argtypes = [C.c_void_p]
    L.ora_triangle_intersect.argtypes = [C.POINTER(Ray), fp, fp, fp, C.c_float, C.c_float, fp]
    L.ora_tri4_intersect.argtypes = [C.POINTER(Ray), fp, up, C.c_int, C.c_uint32, C.c_float, C.c_float, up, up, up,
                                     fp, fp, fp]
    L.ora_clip_triangle_aabb.argtypes = [fp, fp, fp, C.c_int, C.c_float, C.c_float, fp]
    L.ora_affine_inverse.argtypes = [fp, fp]
    # shading
    MP = C.POINTER(Material)
    L.ora_material_default.argtypes = [MP]
    L.ora_material_diffuse.argtypes = [MP, C.c_float, C.c_float, C.c_float]
    L.ora_material_emissive.argtypes = [MP, C.c_float, C.c_float, C.c_float]
    L.ora_mat_scatter.argtypes = [MP, V3, C.POINTER(HitRecord), Sampler, C.POINTER(Scatter)]
    L.ora_mat_eval.argtypes = [MP, V3, C.POINTER(HitRecord), V3, C.POINTER(V3), fp]
    L.ora_mat_emitted.restype = V3
    L.ora_mat_emitted.argtypes = [MP]
    L.ora_mat_emitted_directional.restype = V3
    L.ora_mat_emitted_directional.argtypes = [MP, C.c_float]
    L.ora_eval_all.restype = V3
    L.ora_eval_all.argtypes = [MP, V3, V3, C.c_int]
    L.ora_pdf_all.restype = C.c_float
    L.ora_pdf_all.argtypes = [MP, V3, V3, C.c_int]
    L.ora_lobe_pmf.argtypes = [MP, fp]
    L.ora_f0_from_ior.restype = C.c_float
    L.ora_f0_from_ior.argtypes = [C.c_float]
    L.ora_roughness_to_alpha_aniso.argtypes = [C.c_float, C.c_float, fp, fp]
    L.ora_fresnel_f82_tint.restype = V3
    L.ora_fresnel_f82_tint.argtypes = [C.c_float, V3, V3]
    L.ora_fresnel_dielectric.restype = C.c_float
    L.ora_fresnel_dielectric.argtypes = [C.c_float, C.c_float, C.c_float]
    L.ora_eon_albedo_exact.restype = C.c_float
    L.ora_eon_albedo_exact.argtypes = [C.c_float, C.c_float]
    L.ora_eon_albedo_approx.restype = C.c_float
    L.ora_eon_albedo_approx.argtypes = [C.c_float, C.c_float]
    L.ora_eon_diffuse.restype = V3
    L.ora_eon_diffuse.argtypes = [V3, C.c_float, V3, V3]
    L.ora_cauchy_ior.restype = C.c_float
    L.ora_cauchy_ior.argtypes = [C.c_float, C.c_float, C.c_float]
    L.ora_dispersive_ior.restype = V3
    L.ora_dispersive_ior.argtypes = [C.c_float, C.c_float, C.c_float]
    L.ora_thin_film_fresnel.restype = V3
    L.ora_thin_film_fresnel.argtypes = [C.c_float] * 5
    L.ora_sample_vndf.restype = V3
    L.ora_sample_vndf.argtypes = [V3, C.c_float, C.c_float, C.c_float, C.c_float]
    L.ora_coat_passage.restype = V3
    L.ora_coat_passage.argtypes = [MP, C.c_float]
    L.ora_tangent_frame.argtypes = [V3, C.POINTER(V3), C.POINTER(V3)]
    L.ora_balance_heuristic.restype = C.c_float
    L.ora_balance_heuristic.argtypes = [C.c_float, C.c_float]
    L.ora_power_heuristic.restype = C.c_float
    L.ora_power_heuristic.argtypes = [C.c_float, C.c_float]
    L.ora_cosine_hemisphere.restype = V3
    L.ora_cosine_hemisphere.argtypes = [C.c_float, C.c_float]
    L.ora_concentric_disk.restype = V3
    L.ora_concentric_disk.argtypes = [C.c_float, C.c_float]
    for n, res, args in [
        ("ora_medium_from_transmission", Medium, [V3, C.c_float, V3, C.c_float]),
        ("ora_medium_from_subsurface", Medium, [V3, C.c_float, V3, C.c_float]),
        ("ora_medium_transmittance", V3, [C.POINTER(Medium), C.c_float]),
        ("ora_medium_is_scattering", C.c_int, [C.POINTER(Medium)]),
        ("ora_medium_sigma_t_max", C.c_float, [C.POINTER(Medium)]),
        ("ora_medium_albedo", V3, [C.POINTER(Medium)]),
        ("ora_interior_medium", C.c_int, [MP, C.POINTER(Medium)]),
        ("ora_hg_phase", C.c_float, [C.c_float, C.c_float]),
        ("ora_sample_henyey_greenstein", V3, [V3, C.c_float, C.c_float, C.c_float]),
        ("ora_t_sheen_charlie", C.c_float, [C.c_float] * 4),
        ("ora_t_coat_darkening_factor", V3, [V3, C.c_float, C.c_float]),
        ("ora_t_coat_attenuation", V3, [MP, C.c_float, C.c_float]),
        ("ora_t_eval_coat", V3, [MP, V3, V3, V3, C.c_float, C.c_float]),
        ("ora_t_thin_film_fresnel_metal", V3, [C.c_float, C.c_float, C.c_float, V3, C.c_float]),
        ("ora_t_transmission_iors", V3, [MP]),
        ("ora_t_sample_transmission_thin", V3, [MP, V3, C.POINTER(HitRecord)]),
        ("ora_t_light_sample_point", V3, [C.POINTER(Light), C.c_float, C.c_float]),
        ("ora_t_light_normal_at", V3, [C.POINTER(Light), V3]),
        ("ora_t_light_area", C.c_float, [C.POINTER(Light)]),
        ("ora_t_sampler_new", C.c_uint32, [C.c_int] * 4),
        ("ora_t_new_domain", C.c_uint32, [C.c_uint32, C.c_int]),
        ("ora_t_draw4", None, [C.c_uint32, C.c_uint32, fp]),
        ("ora_t_rnd1", C.c_float, [C.c_uint32, C.c_uint32]),
    ]:
        getattr(L, n).restype = res
        getattr(L, n).argtypes = args
    L.ora_light_sample_li.argtypes = [C.POINTER(Light), V3, C.c_float, C.c_float, C.POINTER(LightSample)]
    L.ora_light_escaped.argtypes = [C.POINTER(Light), V3, C.POINTER(V3), fp]
    L.ora_light_distant.argtypes = [C.POINTER(Light), V3, V3, C.c_float]
    L.ora_align_to_normal.restype = V3
    L.ora_align_to_normal.argtypes = [V3, V3]
    L.ora_light_pdf_at_point.restype = C.c_float
    L.ora_light_pdf_at_point.argtypes = [C.POINTER(Light), V3, V3]
    L.ora_camera_new.argtypes = [C.POINTER(Camera), V3, V3, V3, C.c_float, C.c_float, C.c_float, C.c_float]
    L.ora_camera_get_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.POINTER(Ray)]
    L.ora_filter_sample.argtypes = [C.c_int, C.c_float, C.c_float, fp, fp]
    # integrator
    L.ora_render.argtypes = [C.POINTER(RenderJob), fp, C.POINTER(RayStats), C.c_int]
    L.ora_render_pixels.argtypes = [C.POINTER(RenderJob), C.POINTER(C.c_uint32), C.c_size_t, fp, C.POINTER(RayStats), C.c_int]
    L.ora_render_serial_trav.argtypes = [C.POINTER(RenderJob), fp, C.POINTER(RayStats), C.POINTER(TravStats), C.POINTER(TravStats)]
    L.ora_set_seam_hooks.argtypes = [C.c_void_p]
    L.ora_set_seam_hooks.restype = None
    L.ora_render_pixel.restype = C.c_uint32
    L.ora_render_pixel.argtypes = [C.POINTER(RenderJob), C.c_uint32, C.c_uint32, fp, C.POINTER(RayStats)]
    L.ora_render_sample.argtypes = [C.POINTER(RenderJob), C.c_uint32, C.c_uint32, C.c_uint32, fp,
                                    C.POINTER(RayStats)]
    L.ora_light_weight.restype = C.c_float
    L.ora_light_weight.argtypes = [C.c_int, C.c_float, C.c_float]
    L.ora_bounce_weight.restype = C.c_float
    L.ora_bounce_weight.argtypes = [C.c_int, C.c_float, C.c_float]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


IDENTITY12 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float32)


def affine(m3=None, t=(0, 0, 0)):
    """12 floats: columns x, y, z of the 3x3 then translation (glam Affine3A)."""
    m = np.eye(3, dtype=np.float32) if m3 is None else np.asarray(m3, dtype=np.float32)
    return np.concatenate([m[:, 0], m[:, 1], m[:, 2], np.asarray(t, dtype=np.float32)]).astype(np.float32)


class Scene:
    def __init__(self, handle, keep):
        self.h = handle
        self._keep = keep  # inner scenes must outlive this one

    def intersect(self, r, t_min=0.001, t_max=INF):
        h = RayHit()
        if lib().ora_intersect(self.h, C.byref(r), t_min, t_max, C.byref(h)):
            return h
        return None

    def occluded(self, r, t_min=0.001, t_max=INF):
        return bool(lib().ora_occluded(self.h, C.byref(r), t_min, t_max))

    def linear_scan(self, r, t_min=0.001, t_max=INF):
        h = RayHit()
        if lib().ora_linear_scan(self.h, C.byref(r), t_min, t_max, C.byref(h)):
            return h
        return None

    def bounds(self):
        out = np.zeros(6, dtype=np.float32)
        if lib().ora_scene_bounds(self.h, _fp(out)):
            return out
        return None

    def geometry_count(self):
        return lib().ora_geometry_count(self.h)

    def has_motion(self):
        return bool(lib().ora_has_motion(self.h))

    def primitive_count(self):
        return lib().ora_primitive_count(self.h)

    def _breakdown(self, fn):
        out = (C.c_size_t * 5)()
        fn(self.h, out)
        return dict(zip(("triangles", "spheres", "curve_segments", "cubic_curve_spans", "instances"), map(int, out)))

    def primitive_extents(self):
        out = (C.c_float * 3)()
        n = lib().ora_primitive_extents(self.h, out)
        return n, out[0], out[1], out[2]

    def primitive_breakdown(self):
        return self._breakdown(lib().ora_primitive_breakdown)

    def unique_primitive_breakdown(self):
        return self._breakdown(lib().ora_unique_primitive_breakdown)

    def intersect_n(self, rays, t_min=0.001, t_max=INF):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hf = np.zeros((n, 6), dtype=np.float32)
        ids = np.zeros((n, 2), dtype=np.uint32)
        front = np.zeros(n, dtype=np.uint8)
        lib().ora_intersect_n(self.h, _fp(rays), n, t_min, t_max, _fp(hf), _up(ids),
                              front.ctypes.data_as(C.POINTER(C.c_uint8)))
        return hf, ids, front

    def occluded_n(self, rays, t_min=0.001, t_max=INF):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        out = np.zeros(n, dtype=np.uint8)
        lib().ora_occluded_n(self.h, _fp(rays), n, t_min, t_max, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def counts(self):
        out = (C.c_size_t * 5)()
        lib().ora_bvh_counts(self.h, out)
        return dict(nodes=out[0], leaves=out[1], packets=out[2], indices=out[3], prims=out[4])

    def arrays(self):
        """(nodes[n,32] u32 view, leaves[n,4], packets[n,48] u32 view, indices[n]) as numpy copies."""
        c = self.counts()
        L = lib()

        def grab(ptr, n, words):
            if n == 0:
                return np.zeros((0, words), dtype=np.uint32)
            buf = C.cast(ptr, C.POINTER(C.c_uint32 * (n * words))).contents
            return np.frombuffer(buf, dtype=np.uint32).reshape(n, words).copy()

        return (grab(L.ora_bvh_nodes(self.h), c["nodes"], 32), grab(L.ora_bvh_leaves(self.h), c["leaves"], 4),
                grab(L.ora_bvh_packets(self.h), c["packets"], 48),
                grab(L.ora_bvh_indices(self.h), c["indices"], 1).reshape(-1))


class SceneBuilder:
    """Mirror of crust_rt::SceneBuilder (scene.rs:147-342) over the oracle."""

    def __init__(self):
        self.h = lib().ora_builder_new()
        self._keep = []

    def attach_triangles(self, verts, idx, normals=None, mask=MASK_ALL):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        return lib().ora_attach_triangles(self.h, _fp(verts), verts.shape[0], _up(idx), idx.shape[0], _fp(nrm),
                                          0 if nrm is None else nrm.shape[0], mask)

    def attach_sphere(self, c, r, mask=MASK_ALL):
        return lib().ora_attach_sphere(self.h, c[0], c[1], c[2], r, mask)

    def attach_instance(self, scene, l2w=IDENTITY12, l2w_end=None, mask=MASK_ALL):
        self._keep.append(scene)
        a = np.ascontiguousarray(l2w, dtype=np.float32)
        e = None if l2w_end is None else np.ascontiguousarray(l2w_end, dtype=np.float32)
        return lib().ora_attach_instance(self.h, scene.h, _fp(a), _fp(e), mask)

    def attach_empty(self, mask=MASK_ALL):
        return lib().ora_attach_empty(self.h, mask)

    def set_sphere(self, gid, c, r):
        return lib().ora_set_sphere(self.h, gid, c[0], c[1], c[2], r)

    def set_triangles(self, gid, verts, idx, normals=None):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        return lib().ora_set_triangles(self.h, gid, _fp(verts), verts.shape[0], _up(idx), idx.shape[0], _fp(nrm),
                                       0 if nrm is None else nrm.shape[0])

    def set_instance(self, gid, scene, l2w=IDENTITY12, l2w_end=None):
        self._keep.append(scene)
        a = np.ascontiguousarray(l2w, dtype=np.float32)
        e = None if l2w_end is None else np.ascontiguousarray(l2w_end, dtype=np.float32)
        return lib().ora_set_instance(self.h, gid, scene.h, _fp(a), _fp(e))

    def count(self):
        return lib().ora_builder_count(self.h)

    def commit(self):
        s = Scene(lib().ora_commit(self.h), self._keep)
        self.h = None
        return s


def default_material():
    m = Material()
    lib().ora_material_default(C.byref(m))
    return m


def diffuse_material(rgb):
    m = Material()
    lib().ora_material_diffuse(C.byref(m), rgb[0], rgb[1], rgb[2])
    return m


def emissive_material(rgb):
    m = Material()
    lib().ora_material_emissive(C.byref(m), rgb[0], rgb[1], rgb[2])
    return m
