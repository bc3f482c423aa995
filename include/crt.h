/* crt.h — C ABI of the MI355X-native crust-rt backend (libcrt_amd.so).
 *
 * Drop-in boundary for crust-render's kernel seam: every entry point below is what a Rust FFI
 * shim behind `crust_rt::{Geometry, SceneBuilder, Scene}` would bind. The
 * reference interface each one replaces is cited as file:line under /root/reference/.
 *
 * Conventions
 *  - Plain pointers and sizes only; no C++/torch types. All functions are `extern "C"`.
 *  - Geometry arrays are COPIED at attach time; the caller keeps ownership of its buffers
 *    (the reference moves Vecs into the builder, scene.rs:158-166).
 *  - Return value CRT_OK (0) or a negative CrtStatus. Nothing unwinds across the boundary. The
 *    reference returns no errors on this path (misses are None/false); where it would panic
 *    (set_geometry on an unknown id, scene.rs:197-201) this ABI returns CRT_ERR_BAD_ID.
 *  - A miss is reported as geom_id == CRT_INVALID_ID (lib.rs:53).
 *  - Device pointers: the *_n entry points take pointers to HBM-resident buffers and a HIP stream
 *    handle (hipStream_t passed as void*; NULL = the default stream). The *1 entry points take host
 *    pointers and synchronise.
 *  - The library requires a gfx950 device. There is NO CPU fallback: without a usable HIP device
 *    every query/render entry point returns CRT_ERR_NO_DEVICE.
 */
#ifndef CRT_H
#define CRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_INVALID_ID 0xFFFFFFFFu /* lib.rs:53 INVALID_ID */
#define CRT_MASK_CAMERA 1u         /* ray.rs:8  */
#define CRT_MASK_SHADOW 2u         /* ray.rs:9  */
#define CRT_MASK_INDIRECT 4u       /* ray.rs:10 */
#define CRT_MASK_ALL 0xFFFFFFFFu   /* ray.rs:11 */

typedef enum {
  CRT_OK = 0,
  CRT_ERR_BAD_ARG = -1,
  CRT_ERR_BAD_ID = -2,     /* scene.rs:197-201 (panic in the reference) */
  CRT_ERR_NO_DEVICE = -3,  /* no gfx950 HIP device / HIP call failed */
  CRT_ERR_STACK = -4,      /* traversal stack capacity exceeded (never on trees the builder emits) */
  CRT_ERR_UNSUPPORTED = -5, /* RoundCurves / CubicCurves (scene.rs:99-106): out of scope */
  CRT_ERR_NO_MEMORY = -6    /* a host allocation failed (the reference aborts); reason in crt_last_error. Nothing unwinds
                             * through this ABI: a C or Rust host could not catch it */
} CrtStatus;

/* crust_rt::Ray (ray.rs:18-23): origin/dir are glam Vec3A (16-byte, w unused), then time, mask. 48 bytes. */
typedef struct CrtRay {
  float origin[3]; float _pad0;
  float dir[3];    float _pad1;
  float time;
  uint32_t mask;
  uint32_t _pad2[2];
} CrtRay;

/* crust_rt::RayHit (scene.rs:134-142). 40 bytes. geom_id == CRT_INVALID_ID on a miss. */
typedef struct CrtRayHit {
  float t;
  float normal[3];   /* ray-facing (scene.rs:356-359) */
  uint32_t front_face;
  float u, v;        /* u weights v1, v weights v2 (triangle.rs:84-86) */
  uint32_t geom_id;
  uint32_t prim_id;
  uint32_t _pad;
} CrtRayHit;

/* bvh.rs:39-57 `traversal-stats` mirror; index 0 = top-level tree, 1 = inside an instance. */
typedef struct CrtTravStats {
  uint64_t queries[2], nodes[2], leaves[2], packets[2], prims[2];
  uint64_t accepted_hits, instance_descents, rays;
  /* SIMD-utilisation counters of the device kernels (no reference counterpart): for each traversal phase,
   * how many times a wave executed it (phase_waves) and how many of its 64 lanes were live in total
   * (phase_lanes); lanes/(64*waves) is the phase's lane utilisation. Phases: 0 scheduling loop, 1 ray
   * fetch + setup, 2 node expansion, 3 packet test, 4 scalar-list primitive, 5 instance exit, 6 emit,
   * 7 on-edge f64 fallback. */
  uint64_t phase_waves[8], phase_lanes[8];
  /* Shader-clock cycles the waves spent in each phase (summed over waves; phase 0 = scheduling between phases).
   * Filled by the phase-scheduled engine only. */
  uint64_t phase_cycles[8];
} CrtTravStats;

typedef struct CrtBuilder CrtBuilder; /* crust_rt::SceneBuilder (scene.rs:147-149) */
typedef struct CrtScene CrtScene;     /* crust_rt::Scene, ref-counted like Arc<Scene> (scene.rs:345-349) */

/* ---- SceneBuilder (scene.rs:151-342) ---- */
CrtBuilder *crt_builder_new(void);                                  /* SceneBuilder::new            scene.rs:152 */
void crt_builder_free(CrtBuilder *b);                               /* drop without commit                       */
int crt_reserve(CrtBuilder *b, size_t additional);                  /* SceneBuilder::reserve        scene.rs:174 */
size_t crt_count(const CrtBuilder *b);                              /* SceneBuilder::count          scene.rs:179 */
/* attach_masked(Geometry::TriangleMesh{vertices, indices, normals}, mask) -> geom_id   scene.rs:163, :88-94.
 * verts: n_verts*3 floats; indices: n_tris*3 u32; normals: NULL or n_normals*3 floats. */
int crt_attach_triangles(CrtBuilder *b, const float *verts, size_t n_verts, const uint32_t *indices, size_t n_tris,
                         const float *normals, size_t n_normals, uint32_t mask, uint32_t *geom_id_out);
/* attach_masked(Geometry::Sphere{center, radius}, mask)                               scene.rs:95-98 */
int crt_attach_sphere(CrtBuilder *b, const float center[3], float radius, uint32_t mask, uint32_t *geom_id_out);
/* attach_masked(Geometry::Instance{scene, transform, transform_end}, mask)            scene.rs:111-121.
 * l2w / l2w_end: glam Affine3A as 12 floats (matrix3 columns x, y, z then translation); l2w_end may be
 * NULL. The builder retains `scene`. */
int crt_attach_instance(CrtBuilder *b, CrtScene *scene, const float l2w[12], const float *l2w_end, uint32_t mask,
                        uint32_t *geom_id_out);
/* attach_masked(SceneBuilder::empty_geometry(), mask)                                 scene.rs:205-211 */
int crt_attach_empty(CrtBuilder *b, uint32_t mask, uint32_t *geom_id_out);
/* set_geometry(id, ...) keeping the slot's mask                                       scene.rs:199-201 */
int crt_set_triangles(CrtBuilder *b, uint32_t id, const float *verts, size_t n_verts, const uint32_t *indices,
                      size_t n_tris, const float *normals, size_t n_normals);
int crt_set_sphere(CrtBuilder *b, uint32_t id, const float center[3], float radius);
int crt_set_instance(CrtBuilder *b, uint32_t id, CrtScene *scene, const float l2w[12], const float *l2w_end);
/* commit(self) -> Scene: consumes the builder (scene.rs:226). Deterministic SBVH build + BVH4 collapse
 * on the host (bvh.rs:300-327) on a bounded number of helper threads; the device image is created on first query.
 * Returns NULL (reason in crt_last_error) when memory or threads run out, or when instances nest deeper than the 8
 * levels the kernels carry frames for (the reference's importer stops at the same depth, usd_import.rs:60). */
CrtScene *crt_commit(CrtBuilder *b);

/* ---- Scene (scene.rs:351-479) ---- */
void crt_scene_retain(CrtScene *s);                                 /* Arc::clone */
void crt_scene_release(CrtScene *s);                                /* drop(Arc)  */
int crt_scene_bounds(const CrtScene *s, float out_min_max[6]);      /* Scene::bounds -> 1 if Some  scene.rs:375 */
uint32_t crt_scene_geometry_count(const CrtScene *s);               /* scene.rs:380 */
int crt_scene_has_motion(const CrtScene *s);                        /* scene.rs:395 */
size_t crt_scene_primitive_count(const CrtScene *s);                /* scene.rs:400 */
/* primitive_breakdown: triangles, spheres, curve_segments, cubic_curve_spans, instances   scene.rs:409 */
int crt_scene_primitive_breakdown(const CrtScene *s, size_t out[5]);
/* unique_primitive_breakdown: the same five counts over what is resident in memory — instanced scenes are
 * descended, each distinct prototype once however many placements share it             scene.rs:422-427 */
int crt_scene_unique_primitive_breakdown(const CrtScene *s, size_t out[5]);
/* primitive_extents -> (count, scene diagonal, mean primitive diagonal, max primitive diagonal): the "BVH that will
 * not cull" diagnostic over the top-level primitives' boxes. Any out pointer may be NULL.   scene.rs:446-455 */
int crt_scene_primitive_extents(const CrtScene *s, size_t *count, float *scene_diagonal, float *mean_diagonal,
                                float *max_diagonal);
/* Host-only self-check of the device image this scene would upload — no GPU needed, nothing is uploaded: every child
 * word of every node decodes to exactly the node, leaf, scalar list or instance slots it stands for (plain, direct and
 * direct-instance forms), the node numbering is a permutation with the queried root at 0 and the instanced trees'
 * roots right behind it, every record index is in range, a moving instance's placements sit where its flags word says.
 * out: nodes | leaf words in plain / direct-index / direct-instance form | instance records | moving instances |
 * instanced roots staged for the LDS window | 1 if direct leaves are on. CRT_ERR_BAD_ARG + crt_last_error on a broken
 * invariant. (No reference counterpart: the image is this library's own layout of scene.rs:226-340's commit.) */
int crt_scene_image_check(CrtScene *s, uint64_t out[8]);
/* Host-only (no GPU needed): which instance of the traversal engine this library selects for the image the scene would
 * upload — ONE function decides it for the renderer, the batched and the single-ray queries — verified against a census
 * of the image: the instance decodes every child word (the four-wave kernels carry no direct-leaf form) and keeps every
 * rarely used per-ray field some primitive can need. want_wide: -1 = the scene's own preference, 0 / 1 = asked for, -2 =
 * exactly what a batched query's launch would pick in this process (the CRT_WIDE A/B request included, which falls back
 * where refused), -3 = what the renderer's launches would (they differ on large flat trees), -4 = the renderer's preference;
 * CRT_ERR_UNSUPPORTED when what was asked for cannot decode the image (nothing would be launched), CRT_ERR_BAD_ARG +
 * crt_last_error on a broken invariant. out: four-wave kernels (1; 2 = their direct-engine instances, which the renderer
 * runs direct-leaf images on) | direct-leaf engine copy | LDS stack entries per ray |
 * nodes staged in LDS | cold mask of the per-stage closest-hit kernel | of the fused kernel | direct words in the image |
 * cold mask the image needs. (No reference counterpart: the reference has one scalar traversal, bvh.rs:441-509.) */
int crt_scene_engine_select(CrtScene *s, int want_wide, uint32_t out[8]);
/* memory_footprint: prim_nodes, boxed_prims, bvh_nodes, leaves, packets, indices (device bytes) scene.rs:459 */
int crt_scene_memory_footprint(CrtScene *s, size_t out[6]);
/* Host copies of the committed tree of THIS scene (local indices), for build-parity checks:
 * counts = nodes, leaves, packets, indices, prims. Pointers stay valid while the scene lives. */
int crt_scene_tree(const CrtScene *s, size_t counts[5], const void **nodes128, const void **leaves16,
                   const void **packets192, const uint32_t **indices);

/* Scene::intersect(&ray, t_min, t_max) -> Option<RayHit>: 1 = hit, 0 = miss, <0 error      scene.rs:354
 * Thread-safe as the reference's queries are ("&self and thread-safe", scene.rs:344): every calling thread stages
 * through its own persistent pinned/device record and its own stream, created on its first call — no allocation,
 * no device-wide synchronisation and no state shared between threads per query. */
int crt_intersect1(CrtScene *s, const CrtRay *ray, float t_min, float t_max, CrtRayHit *hit);
/* Scene::occluded(&ray, t_min, t_max) -> bool: 1 / 0 / <0                                  scene.rs:370 */
int crt_occluded1(CrtScene *s, const CrtRay *ray, float t_min, float t_max);
/* Frees the calling thread's single-ray staging area (a worker thread about to exit calls this; optional). */
void crt_thread_release(void);
/* Batched forms for the wavefront integrator: d_rays / d_hits / d_out are DEVICE pointers,
 * n rays, launched on `stream` without synchronising. d_out: one u32 per ray (1 = occluded). */
int crt_intersect_n(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, CrtRayHit *d_hits,
                    void *stream);
int crt_occluded_n(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                   void *stream);
/* The batched forms do not synchronise, so they cannot report a traversal that overflowed its stack (more than 255
 * pending entries): the kernels OR that into the SCENE's error word. This call drains `stream`, reads and clears
 * the word: CRT_OK, or CRT_ERR_STACK if any launch on this scene since the last call met the condition (the
 * affected rays' results are then undefined). The single-ray and *_stats forms check their own launches. */
int crt_scene_traversal_error(CrtScene *s, void *stream);
/* Same launches with the traversal counters compiled in; counts are added into *host_stats after a
 * stream sync. Take counts from these and timings from the plain forms (bvh.rs:31-38). */
int crt_intersect_n_stats(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, CrtRayHit *d_hits,
                          void *stream, CrtTravStats *host_stats);
int crt_occluded_n_stats(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                         void *stream, CrtTravStats *host_stats);

/* ---- shading seam: the closed Material table (material.rs:26-116, rt_world.rs:111-122) ---- */
enum { CRT_MAT_OPENPBR = 0, CRT_MAT_EMISSIVE = 1 };
/* OpenPBR parameter block (openpbr.rs:66-121); for CRT_MAT_EMISSIVE only emission_color (the emitted
 * radiance, emissive.rs:12-14) is read. */
typedef struct CrtMaterial {
  uint32_t kind;
  uint32_t thin_walled;
  float base_weight; float base_color[3]; float base_diffuse_roughness; float base_metalness;
  float specular_weight; float specular_color[3]; float specular_roughness; float specular_ior;
  float specular_roughness_anisotropy;
  float transmission_weight; float transmission_color[3]; float transmission_depth; float transmission_scatter[3];
  float transmission_scatter_anisotropy; float transmission_dispersion_scale; float transmission_dispersion_abbe_number;
  float subsurface_weight; float subsurface_color[3]; float subsurface_radius; float subsurface_radius_scale[3];
  float subsurface_scatter_anisotropy;
  float fuzz_weight; float fuzz_color[3]; float fuzz_roughness;
  float coat_weight; float coat_color[3]; float coat_roughness; float coat_roughness_anisotropy; float coat_ior;
  float coat_darkening;
  float thin_film_weight; float thin_film_thickness; float thin_film_ior;
  float emission_luminance; float emission_color[3];
  float geometry_opacity;
} CrtMaterial;
void crt_material_default(CrtMaterial *m);                          /* OpenPBR::default   openpbr.rs:130-173 */

enum { CRT_LIGHT_SPHERE = 0, CRT_LIGHT_RECT = 1, CRT_LIGHT_DISTANT = 2, CRT_LIGHT_DOME = 3 };
/* AreaLight{shape, material, geom_id} (light.rs:156-163) with SphereShape (:22-25) / RectShape (:52-57).
 * The two lights at infinity reuse the record, already in their derived form and with geom_id = CRT_INVALID_ID:
 *   DISTANT (DistantLight, light.rs:234-318): normal = unit travel direction, radiance = irradiance,
 *           radius = cos(half angle), center[0] = cone solid angle 2*pi*(1 - cos(half angle))   (light.rs:255-266)
 *   DOME    (DomeLight without an environment map, light.rs:320-390): radiance = tint; uniform over the sphere. */
typedef struct CrtLight {
  uint32_t kind; uint32_t geom_id;
  float radiance[3];
  float center[3]; float radius;
  float origin[3]; float edge_u[3]; float edge_v[3]; float normal[3];
} CrtLight;

/* Camera (camera.rs:8-25), already in its derived form. */
typedef struct CrtCamera {
  float origin[3], lower_left[3], horizontal[3], vertical[3], u[3], v[3];
  float lens_radius;
} CrtCamera;
/* Camera::new(lookfrom, lookat, vup, vfov_deg, aspect, aperture, focus_dist)   camera.rs:27-63 */
void crt_camera_new(CrtCamera *c, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov_deg,
                    float aspect, float aperture, float focus_dist);

/* ---- the shading seam as callable functions: `trait Material` (material.rs:26-116) and `trait Light` (light.rs:120-151)
 * for a host integrator that keeps its own trace_path (tracer.rs:1086-1558) on top of crt_intersect_n / crt_occluded_n.
 * Batched like those: every pointer is a DEVICE pointer, n records, launched on `stream` without synchronising; the
 * arithmetic is the device code crt_render_samples runs (kernels/shade.hip.h), one query per lane. The material / light
 * tables are plain arrays of the records above, uploaded by the caller (crt_renderer_new keeps its own copies).
 * INTEGRATION.md binds them as `impl Material for DeviceMaterial` / `impl Light for DeviceLight`. ---- */

/* One call of a Material method: r_in, rec (HitRecord, hittable.rs:10-36; face_id / face_uv are Ptex-only: out of scope),
 * and the method's own argument. 80 bytes. */
typedef struct CrtShadeQuery {
  float ray_dir[3]; uint32_t material;      /* r_in.direction (unnormalised allowed) | record of the material table   */
  float p[3]; float t;                      /* rec.p, rec.t                                                            */
  float normal[3]; uint32_t front_face;     /* rec.normal (faces the ray, scene.rs:356-359), rec.front_face            */
  float wi[3]; float cos_theta_o;           /* eval: the direction to evaluate | emitted_directional: cos(theta_o)     */
  uint32_t sampler_pattern, sampler_index;  /* scatter_importance: the PathSampler domain handed over by value          */
  uint32_t _pad[2];                         /*   (pattern of the domain, sample index: tracer.rs:1121, openpbr.rs:1042) */
} CrtShadeQuery;
/* Option<ScatterSample> (material.rs:8-22): some = 0 is None (no field below is meaningful then). 48 bytes. */
typedef struct CrtScatterSample {
  float origin[3]; uint32_t some;           /* ray.origin (p, or p + l * 1e-4 across the interface, openpbr.rs:1063)   */
  float dir[3]; float pdf;                  /* ray.direction (unnormalised as the reference leaves it) | pdf >= 1e-4   */
  float value[3]; uint32_t flags;           /* brdf * |cos| | bit 0: delta, bit 1: the ray carries the material's      */
} CrtScatterSample;                         /*   interior medium (Ray::new_in_medium, openpbr.rs:1061-1066)            */
/* Option<(Vec3A, f32)> of Material::eval (material.rs:56-74). 32 bytes. */
typedef struct CrtBsdfEval { float value[3]; float pdf; uint32_t some; uint32_t _pad[3]; } CrtBsdfEval;
/* One call of a Light method. 48 bytes. */
typedef struct CrtLightQuery {
  float from[3]; uint32_t light;            /* the shading point | entry of the light list                              */
  float u, v; uint32_t _pad[2];             /* sample_li: the two unit random numbers                                   */
  float point[3]; uint32_t _pad2;           /* pdf_at_point: light_point | escaped: the (unit) direction                */
} CrtLightQuery;
/* Option<LightSample> (light.rs:90-105); escaped's Option<(radiance, pdf)> uses radiance, pdf and some. 48 bytes. */
typedef struct CrtLightSample {
  float direction[3]; float distance;       /* unit, towards the light | INFINITY for lights at infinity                */
  float radiance[3]; float pdf;
  uint32_t some; uint32_t _pad[3];
} CrtLightSample;

/* Material::scatter_importance(r_in, rec, sampler) -> Option<ScatterSample>     material.rs:40-45; OpenPBR
 * scatter_resolved openpbr.rs:1026-1136 (draws ONE 4-D block from the domain: s[0] lobe, s[1..3] direction, s[3]
 * dispersion channel); Emissive never scatters (emissive.rs:30-38). A query whose material index is out of range
 * answers None (the reference would panic on the index). */
int crt_material_scatter_n(const CrtMaterial *d_materials, size_t n_materials, const CrtShadeQuery *d_queries, size_t n,
                           CrtScatterSample *d_out, void *stream);
/* Material::eval(r_in, rec, wi) -> Option<(value = brdf * |cos|, pdf >= 1e-4)>   material.rs:56-74; openpbr.rs:1138-1158.
 * None iff the material is Emissive or the view direction is below the surface — never because of wi. */
int crt_material_eval_n(const CrtMaterial *d_materials, size_t n_materials, const CrtShadeQuery *d_queries, size_t n,
                        CrtBsdfEval *d_out, void *stream);
/* Material::emitted_directional(cos_theta_o) -> Vec3A                          material.rs:112-115; openpbr.rs:1211-1218
 * (emission seen through the coat); Emissive: its radiance (emissive.rs:25-28). d_rgb: 3 floats per query. */
int crt_material_emitted_n(const CrtMaterial *d_materials, size_t n_materials, const CrtShadeQuery *d_queries, size_t n,
                           float *d_rgb, void *stream);
/* Light::sample_li(from, u, v) -> Option<LightSample>                           light.rs:126, AreaLight :191-204 (sphere
 * :27-36, rect :69-81), DistantLight :285-298, uniform DomeLight :358-383. */
int crt_light_sample_n(const CrtLight *d_lights, size_t n_lights, const CrtLightQuery *d_queries, size_t n,
                       CrtLightSample *d_out, void *stream);
/* Light::pdf_at_point(from, light_point) -> f32                                 light.rs:132, AreaLight :206-208 ->
 * solid_angle_pdf :180-187; 0 for lights at infinity (the trait's default). d_pdf: one float per query. */
int crt_light_pdf_n(const CrtLight *d_lights, size_t n_lights, const CrtLightQuery *d_queries, size_t n, float *d_pdf,
                    void *stream);
/* Light::escaped(from, direction) -> Option<(radiance, pdf)>                    light.rs:141-146, :300-303, :385-388;
 * None (some = 0) for area lights and for directions the light does not cover. out.direction = the query's. */
int crt_light_escaped_n(const CrtLight *d_lights, size_t n_lights, const CrtLightQuery *d_queries, size_t n,
                        CrtLightSample *d_out, void *stream);

enum { CRT_STRATEGY_POWER = 0, CRT_STRATEGY_BALANCE = 1, CRT_STRATEGY_LIGHT = 2, CRT_STRATEGY_BSDF = 3 }; /* tracer.rs:63-75 */
enum { CRT_FILTER_BOX = 0, CRT_FILTER_TRIANGLE = 1 };                                                       /* filter.rs:27-41 */

/* RenderSettings (tracer.rs:640-686). variance_threshold > 0 turns on render_pixel's adaptive early stop
 * (tracer.rs:609-617): per pixel, after every 4th sample once min_spp samples are in, stop when the relative
 * standard error of the mean luminance falls below the threshold. The rule is evaluated sample by sample inside
 * the film fold, so the image does not depend on how the samples are batched; a pixel that stops inside a batch
 * has had the rest of that batch traced in vain (RayStats then counts those rays: use batches of 4 samples after
 * the first min_spp to count exactly what the reference counts). 0 = every pixel takes every sample. */
typedef struct CrtRenderSettings {
  uint32_t width, height;
  uint32_t max_depth;
  int32_t frame;
  int32_t strategy;
  int32_t filter_kind;
  float filter_radius;
  float variance_threshold;
  uint32_t min_spp;          /* RenderSettings::min_samples; values below 2 mean 2 (tracer.rs:526) */
} CrtRenderSettings;

/* RayStats (stats.rs:128-147). */
typedef struct CrtRayStats {
  uint64_t camera_rays, closest_hit, shadow_rays, vertices, rr_tested, rr_killed, ended_escaped, ended_depth;
} CrtRayStats;

/* Pixel-tile sharding (host-only, no device needed): the frame's 16x16 tiles (tracer.rs:424, :1671-1686) are
 * dealt round-robin to `world` ranks; returns how many pixels `rank` owns and, if out != NULL, their linear
 * buffer indices j*width+i in the order the renderer traces and reports them. */
size_t crt_shard_pixels(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t *out);

/* The tile gather of the multi-GPU path for a host that drives the collective itself (RCCL's ncclAllGather / a
 * ncclSend-ncclRecv group over xGMI; bench.py uses torch.distributed): every rank copies its crt_film_resolve output
 * (pixel_count x 3 floats) to the front of a zeroed buffer of crt_shard_padded_count x 3 floats — the largest shard of
 * the frame, rounded up to 256 pixels, the same on every rank — the collective concatenates the ranks' buffers in rank
 * order, and crt_gather_plan_assemble scatters world x padded x 3 floats into the frame (width x height x 3, buffer order
 * j * width + i) with one launch on `stream`. The plan holds the index arrays of crt_shard_pixels in HBM.
 * (The reference renders its tiles on Rayon workers into one shared buffer, tracer.rs:424-459.) */
size_t crt_shard_padded_count(uint32_t width, uint32_t height, uint32_t world);
typedef struct CrtGatherPlan CrtGatherPlan;
CrtGatherPlan *crt_gather_plan_new(uint32_t width, uint32_t height, uint32_t world);
void crt_gather_plan_free(CrtGatherPlan *p);
size_t crt_gather_plan_padded_count(const CrtGatherPlan *p);
int crt_gather_plan_assemble(const CrtGatherPlan *p, const float *d_recv, float *d_frame, void *stream);

typedef struct CrtRenderer CrtRenderer; /* Renderer{camera, world, lights, settings} (tracer.rs:137-148) */

/* Renderer::new: binds the committed scene (retained), one material per geom_id (rt_world.rs:111-122),
 * the light list (light.rs:392-395) and the camera. The renderer owns the pixels crt_shard_pixels(width,
 * height, tile_rank, tile_world) lists (pass 0, 1 for the whole frame). */
CrtRenderer *crt_renderer_new(CrtScene *scene, const CrtMaterial *materials, size_t n_materials,
                              const CrtLight *lights, size_t n_lights, const CrtCamera *camera,
                              const CrtRenderSettings *settings, uint32_t tile_rank, uint32_t tile_world);
void crt_renderer_free(CrtRenderer *r);
/* Number of pixels this renderer owns and their linear indices (j*width+i, buffer order) in sample order. */
size_t crt_renderer_pixel_count(const CrtRenderer *r);
int crt_renderer_pixel_indices(const CrtRenderer *r, uint32_t *out);
/* Traces samples [sample_begin, sample_begin + sample_count) of every owned pixel as one wavefront batch
 * on `stream` and adds them, in sample order, into the renderer's device-resident film sums
 * (render_pixel's `sum += color`, tracer.rs:599). Does not synchronise — except with adaptive stopping, where the
 * size of the next batch (the pixels still sampling) is read back after the fold. */
int crt_render_samples(CrtRenderer *r, uint32_t sample_begin, uint32_t sample_count, void *stream);
/* Film: pixel = sum / weight_sum (tracer.rs:630-634) for the owned pixels, in pixel_indices order,
 * written to a DEVICE buffer of pixel_count*3 floats (d_rgb), or a HOST buffer via crt_film_read. */
int crt_film_resolve(CrtRenderer *r, float *d_rgb, void *stream);
int crt_film_read(CrtRenderer *r, float *host_rgb);
int crt_film_clear(CrtRenderer *r, void *stream);
/* Adaptive stopping: how many owned pixels are still sampling (all of them when variance_threshold == 0), and
 * the samples each owned pixel has taken so far (pixel_indices order; CRT_ERR_UNSUPPORTED without adaptive stopping). */
size_t crt_renderer_active_pixels(const CrtRenderer *r);
int crt_renderer_sample_counts(CrtRenderer *r, uint32_t *host_out);
/* Counters since the last clear (syncs the stream the batches ran on). */
int crt_render_stats(CrtRenderer *r, CrtRayStats *out);
/* Diagnostic for shade's material-class partition: while enabled (enable = 1 / 0; -1 leaves it as is) the vertex step
 * counts, per material class (0 emissive / escaped, 1 base, 2 coat-fuzz-thin-film, 3 transmissive-subsurface), the wave
 * executions that contained a vertex of the class and how many of their 64 lanes held one. out_* (both or neither):
 * the counts since the last read, which also clears them. lanes / (64 * waves) is the class's lane utilisation. */
int crt_renderer_shade_class_stats(CrtRenderer *r, int enable, uint64_t out_waves[4], uint64_t out_lanes[4]);
/* The launch pipeline of the last batch rendered (before the first: the scene's preference): out[0] = 1 fused (one
 * launch runs generate and every bounce's extend, shade and shadow stage of the batch), 0 one launch per stage and
 * bounce; out[1] = 1 when the traversal kernels are the four-workgroups-per-CU instances (flat triangle scenes: small
 * trees and, in the renderer, large ones; crt_scene_engine_select); out[2] = workgroups (= queue segments) per launch.
 * The renderer takes the per-stage form for batches of >= 96 Mi paths and the fused kernel for smaller ones and for the
 * tail of a large one. Environment CRT_FUSED / CRT_WIDE / CRT_STAGE_MIN_PATHS / CRT_GRID_MULT override. */
int crt_renderer_pipeline(const CrtRenderer *r, uint32_t out[3]);
/* Lanes of the last batch rendered (1 before the first): a batch runs as up to 4 (the default; CRT_LANES) sub-batches of
 * consecutive samples — as many as keep 96 Mi paths each (CRT_LANE_MIN_PATHS) — each with its own buffers and counters on
 * its own HIP stream, so one sub-batch's launches overlap the others'; the film fold stays on the caller's stream, lane
 * after lane — samples are summed in the order of one batch and the image bits do not depend on the lane count
 * (tests/test_gpu_render.py). Every lane's buffers are allocated before any lane launches; a failure after that drains
 * the lanes' streams before it is reported.
 * No reference counterpart (the reference renders tiles on Rayon workers, tracer.rs:424-459). */
int crt_renderer_lanes(const CrtRenderer *r);
/* How many lanes later batches may run as: 1 .. 4 (values outside are clamped); returns the count set. A measurement
 * tool asks for 1 to time every launch alone on the chip (bench.py's roofline legs); results do not depend on it. */
int crt_renderer_set_lanes(CrtRenderer *r, int lanes);
/* Live HIP-event timing of the kernels launched by crt_render_samples since the last reset, by class:
 * 0 = extend (closest-hit traversal) — or, in the fused pipeline, the path-loop kernel that runs generate, extend,
 * shade and shadow of a whole batch in one launch — 1 = shade, 2 = shadow (occlusion traversal), 3 = other.
 * out_ms[k] = summed duration, out_launches[k] = launches. Enabled by crt_renderer_profile(r, 1). */
int crt_renderer_profile(CrtRenderer *r, int enable);
int crt_renderer_profile_read(CrtRenderer *r, double out_ms[4], uint64_t out_launches[4]);
/* Traversal counters for one batch, from the stats build of the kernels: host_stats[0] += the extend
 * (closest-hit) launches, host_stats[1] += the shadow (any-hit) launches. Synchronises the stream. */
int crt_render_samples_stats(CrtRenderer *r, uint32_t sample_begin, uint32_t sample_count, void *stream,
                             CrtTravStats host_stats[2]);

/* Library / device info. crt_last_error: text of the last failing HIP call on this thread ("" if none). */
const char *crt_version(void);
const char *crt_last_error(void);
int crt_device_info(char *name_out, size_t name_cap, int *cu_count, size_t *hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* CRT_H */
