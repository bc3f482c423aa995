/* ORACLE — TEST INFRASTRUCTURE ONLY (see ora_math.h header).
 *
 * CPU restatement of the surface arm of the path-tracing integrator:
 *   crates/crust-core/src/tracer.rs:63-105, 515-636, 828-1035, 1086-1558
 *   crates/crust-core/src/rt_world.rs:207-237
 * Carried media, volume regions and path guiding are out of scope.
 */
#ifndef ORA_PT_H
#define ORA_PT_H
#include "ora_shade.h"

enum { ORA_STRATEGY_POWER = 0, ORA_STRATEGY_BALANCE = 1, ORA_STRATEGY_LIGHT = 2, ORA_STRATEGY_BSDF = 3 }; /* tracer.rs:63-75 */

/* stats.rs:128-147 */
typedef struct {
  uint64_t camera_rays, closest_hit, shadow_rays, vertices, rr_tested, rr_killed, ended_escaped, ended_depth;
} OraRayStats;

typedef struct {
  const OraScene *scene;
  const OraMaterial *materials; /* indexed by geom_id (rt_world.rs:229) */
  uint32_t n_materials;
  const OraLight *lights;
  uint32_t n_lights;
  OraCamera camera;
  uint32_t width, height;
  uint32_t spp, max_depth, min_spp;
  float variance_threshold;
  int32_t frame;
  int32_t strategy;
  int32_t filter_kind;
  float filter_radius;
  /* 0: the reference's estimator, forward walk + backward gather (tracer.rs:1537-1557).
   * 1: the algebraically identical forward accumulation L += beta * (...) that the wavefront
   *    kernels evaluate (same samples, same decisions; float association differs). */
  int32_t forward;
} OraRenderJob;

/* Renders pixel rows [row_begin, row_end) into rgb (width*height*3, row-major in BUFFER order:
 * row j is the reference's Buffer row j, j = 0 at the bottom of the image, buffer.rs:45-49).
 * n_threads workers take 16x16 tiles (tracer.rs:424). */
void ora_render(const OraRenderJob *job, float *rgb, OraRayStats *stats, int n_threads);
/* The whole frame on the calling thread; traversal counters (bvh.rs:39-57) of closest-hit and any-hit queries apart. */
void ora_render_serial_trav(const OraRenderJob *job, float *rgb, OraRayStats *stats, OraTravStats *closest, OraTravStats *any);
/* The pixels idx[0..n) (linear buffer indices j*width+i): rgb[3k..3k+3) = what ora_render writes at idx[k]. */
void ora_render_pixels(const OraRenderJob *job, const uint32_t *idx, size_t n, float *rgb, OraRayStats *stats,
                       int n_threads);
/* One pixel (tracer.rs:515-636); returns the mean and the number of samples taken. */
uint32_t ora_render_pixel(const OraRenderJob *job, uint32_t i, uint32_t j, float rgb[3], OraRayStats *stats);
/* One camera sample of one pixel: radiance before the filter weight (tracer.rs:559-598). */
void ora_render_sample(const OraRenderJob *job, uint32_t i, uint32_t j, uint32_t sample, float rgb[3],
                       OraRayStats *stats);

/* Seam hooks (tests only): the integrator above calls the kernel seam (World::intersect / occluded, rt_world.rs:207-237)
 * and the Material / Light trait methods (material.rs:26-116, light.rs:120-151) through these when they are set — so a
 * test can run THIS per-pixel integrator, unmodified, on somebody else's implementation of the seam (the device's, through
 * libcrt_amd's C ABI) and compare images bit for bit. NULL members fall through to the oracle's own functions. One global
 * table, not thread-safe: render with one thread while hooks are set. Materials and lights are named by their index in
 * the job's tables (materials: by geom_id, rt_world.rs:229). */
typedef struct OraSeamHooks {
  void *ctx;
  int (*intersect)(void *ctx, const OraRay *ray, float t_min, float t_max, OraRayHit *out);
  int (*occluded)(void *ctx, const OraRay *ray, float t_min, float t_max);
  int (*mat_scatter)(void *ctx, uint32_t material, const float *ray_dir, const OraHitRecord *rec, uint32_t dom_pattern,
                     uint32_t dom_index, OraScatter *out);
  int (*mat_eval)(void *ctx, uint32_t material, const float *ray_dir, const OraHitRecord *rec, const float *wi, float *value,
                  float *pdf);
  void (*mat_emitted)(void *ctx, uint32_t material, float cos_theta_o, float *rgb);
  int (*light_sample)(void *ctx, uint32_t light, const float *from, float u, float v, OraLightSample *out);
  float (*light_pdf)(void *ctx, uint32_t light, const float *from, const float *point);
  int (*light_escaped)(void *ctx, uint32_t light, const float *direction, float *radiance, float *pdf);
} OraSeamHooks;
void ora_set_seam_hooks(const OraSeamHooks *hooks); /* NULL: the oracle's own functions again */

float ora_light_weight(int strategy, float light_pdf, float bounce_pdf);  /* tracer.rs:85-92 */
float ora_bounce_weight(int strategy, float bounce_pdf, float light_pdf); /* tracer.rs:97-104 */

#endif
