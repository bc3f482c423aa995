/* ORACLE — TEST INFRASTRUCTURE ONLY (see ora_math.h header).
 *
 * CPU restatement of the shading side of the hot path:
 *   crates/crust-core/src/material/{brdf,openpbr,emissive,material}.rs
 *   crates/crust-core/src/{light,camera,filter}.rs, crates/utils/src/common.rs
 */
#ifndef ORA_SHADE_H
#define ORA_SHADE_H
#include "ora_math.h"
#include "ora_qmc.h"
#include "ora_rt.h"

enum { ORA_MAT_OPENPBR = 0, ORA_MAT_EMISSIVE = 1 };

/* openpbr.rs:66-121 (OpenPBR) / emissive.rs:12-14 (Emissive: emission_color holds the radiance). */
typedef struct {
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
} OraMaterial;

void ora_material_default(OraMaterial *m);              /* openpbr.rs:130-173 */
void ora_material_diffuse(OraMaterial *m, float r, float g, float b); /* openpbr.rs:177-183 */
void ora_material_emissive(OraMaterial *m, float r, float g, float b);

/* hittable.rs:10-36 (face_id/face_uv are Ptex-only: out of scope) */
typedef struct { v3 p, normal; float t; int front_face; } OraHitRecord;

/* material.rs:8-22 ScatterSample; the continuation ray's origin/direction. */
typedef struct { v3 origin, dir; v3 value; float pdf; int delta; int medium; /* ray enters the interior medium */ } OraScatter;

int ora_mat_scatter(const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec, OraSampler bsdf_dom, OraScatter *out); /* openpbr.rs:1026-1136 */
int ora_mat_eval(const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec, v3 wi, v3 *value, float *pdf);          /* openpbr.rs:1138-1158 */
v3 ora_mat_emitted(const OraMaterial *m);                                                                        /* openpbr.rs:1202-1204 */
v3 ora_mat_emitted_directional(const OraMaterial *m, float cos_theta_o);                                        /* openpbr.rs:1211-1218 */

/* medium.rs:22-160: homogeneous interior medium carried by a ray that refracted into a closed surface. */
typedef struct { v3 sigma_a, sigma_s; float g; } OraMedium;
OraMedium ora_medium_from_transmission(v3 tint, float depth, v3 scatter, float anisotropy);  /* medium.rs:40-66 */
OraMedium ora_medium_from_subsurface(v3 albedo, float radius, v3 radius_scale, float g);      /* medium.rs:77-97 */
OraMedium ora_medium_blend(const OraMedium *a, float wa, const OraMedium *b, float wb);       /* medium.rs:103-114 */
v3 ora_medium_transmittance(const OraMedium *m, float t);                                     /* medium.rs:117-120 */
int ora_medium_is_scattering(const OraMedium *m);                                             /* medium.rs:125-127 */
float ora_medium_sigma_t_max(const OraMedium *m);                                             /* medium.rs:131-133 */
v3 ora_medium_albedo(const OraMedium *m);                                                     /* medium.rs:136-140 */
int ora_interior_medium(const OraMaterial *m, OraMedium *out);                                /* openpbr.rs:225-258; 0 = None */
float ora_hg_phase(float cos_theta, float g);                                                 /* medium.rs:148-152 */
v3 ora_sample_henyey_greenstein(v3 wi, float g, float u1, float u2);                          /* medium.rs:158-184 */

/* Local-frame entry points for known-answer tests. */
v3 ora_eval_all(const OraMaterial *m, v3 v_local, v3 l_local, int entering);  /* openpbr.rs:629-683 */
float ora_pdf_all(const OraMaterial *m, v3 v_local, v3 l_local, int entering); /* openpbr.rs:689-722 */
void ora_lobe_pmf(const OraMaterial *m, float out[5]);                       /* openpbr.rs:317-378 */

/* brdf.rs helpers exposed for tests */
float ora_f0_from_ior(float ior);
void ora_roughness_to_alpha_aniso(float r, float a, float *ax, float *ay);
v3 ora_fresnel_f82_tint(float cos_theta, v3 f0, v3 tint);
float ora_fresnel_dielectric(float cos_i, float eta_i, float eta_t);
float ora_eon_albedo_exact(float mu, float roughness);
float ora_eon_albedo_approx(float mu, float roughness);
v3 ora_eon_diffuse(v3 rho, float roughness, v3 v_local, v3 l_local);
float ora_cauchy_ior(float n_d, float v_d, float lambda_nm);
v3 ora_dispersive_ior(float n_d, float abbe, float scale);
v3 ora_thin_film_fresnel(float cos1, float eta1, float eta_film, float eta2, float thickness_nm);
v3 ora_sample_vndf(v3 v_local, float ax, float ay, float u1, float u2);
v3 ora_coat_passage(const OraMaterial *m, float cos_theta);
void ora_tangent_frame(v3 n, v3 *t, v3 *b);

/* common.rs */
float ora_balance_heuristic(float a, float b); /* common.rs:38-40 */
float ora_power_heuristic(float a, float b);   /* common.rs:45-49 */
v3 ora_cosine_hemisphere(float u, float v);    /* common.rs:128-135 */
v3 ora_concentric_disk(float u, float v);      /* common.rs:158-174 */

/* light.rs: area lights */
/* Lights at infinity reuse the record: DISTANT (light.rs:234-318) keeps its unit travel direction in `normal`, the
 * irradiance in `radiance`, cos(half angle) in `radius` and the cone's solid angle in `center[0]`; DOME (uniform,
 * light.rs:320-390 without an environment map) keeps its tint in `radiance`. geom_id is INVALID for both. */
enum { ORA_LIGHT_SPHERE = 0, ORA_LIGHT_RECT = 1, ORA_LIGHT_DISTANT = 2, ORA_LIGHT_DOME = 3 };
typedef struct {
  uint32_t kind; uint32_t geom_id;
  float radiance[3];
  float center[3]; float radius;                            /* SphereShape light.rs:22-25 */
  float origin[3]; float edge_u[3]; float edge_v[3]; float normal[3]; /* RectShape light.rs:52-57 (normal normalized) */
} OraLight;
typedef struct { v3 direction; float distance; v3 radiance; float pdf; } OraLightSample; /* light.rs:90-105 */
int ora_light_sample_li(const OraLight *l, v3 from, float u, float v, OraLightSample *out); /* light.rs:191-204 */
float ora_light_pdf_at_point(const OraLight *l, v3 from, v3 light_point);                   /* light.rs:206-208 */
int ora_light_escaped(const OraLight *l, v3 direction, v3 *radiance, float *pdf);            /* light.rs:141-146, :300-303, :385-388 */
void ora_light_distant(OraLight *l, v3 direction, v3 irradiance, float angle_deg);           /* DistantLight::new light.rs:255-266 */
v3 ora_align_to_normal(v3 local, v3 normal);                                                 /* common.rs:176-188 */

/* camera.rs */
typedef struct { v3 origin, lower_left, horizontal, vertical, u, v; float lens_radius; } OraCamera;
void ora_camera_new(OraCamera *c, v3 lookfrom, v3 lookat, v3 vup, float vfov_deg, float aspect, float aperture,
                    float focus_dist);                                  /* camera.rs:27-63 */
void ora_camera_get_ray(const OraCamera *c, float s, float t, float lu, float lv, float time, OraRay *out); /* camera.rs:71-84 */

/* filter.rs: box + triangle (analytic) */
enum { ORA_FILTER_BOX = 0, ORA_FILTER_TRIANGLE = 1 };
void ora_filter_sample(int kind, float radius, float u, float *offset, float *weight); /* filter.rs:182-205 */

#endif
