/* ORACLE — TEST INFRASTRUCTURE ONLY (see ora_math.h header).
 *
 * CPU restatement of tracer.rs (surface arm) and rt_world.rs.
 */
#include "ora_pt.h"
#include <pthread.h>
#include <stdlib.h>

/* tracer.rs:20-30 */
enum { K_CAMERA = 0, K_PATH = 1, K_TIME = 2 };
enum { K_NEE = 0, K_NEE_SHADOW = 1, K_BSDF = 2, K_GUIDE = 3, K_PHASE = 4, K_RR = 5, K_MEDIUM = 6, K_VOLUME = 7 };
#define RR_START_BOUNCE 3 /* tracer.rs:46 */
#define RR_MIN_PROB 0.05f /* tracer.rs:47 */

float ora_light_weight(int s, float light_pdf, float bounce_pdf) { /* tracer.rs:85-92 */
  switch (s) {
    case ORA_STRATEGY_POWER: return ora_power_heuristic(light_pdf, bounce_pdf);
    case ORA_STRATEGY_BALANCE: return ora_balance_heuristic(light_pdf, bounce_pdf);
    case ORA_STRATEGY_LIGHT: return 1.0f;
    default: return 0.0f;
  }
}
float ora_bounce_weight(int s, float bounce_pdf, float light_pdf) { /* tracer.rs:97-104 */
  switch (s) {
    case ORA_STRATEGY_POWER: return ora_power_heuristic(bounce_pdf, light_pdf);
    case ORA_STRATEGY_BALANCE: return ora_balance_heuristic(bounce_pdf, light_pdf);
    case ORA_STRATEGY_LIGHT: return 0.0f;
    default: return 1.0f;
  }
}
static int samples_lights(int s) { return s != ORA_STRATEGY_BSDF; } /* tracer.rs:79-81 */

/* Seam hooks (ora_pt.h): every call the integrator makes across the kernel seam and the Material / Light traits goes
 * through one of the s_* functions below. */
static OraSeamHooks g_hooks;
void ora_set_seam_hooks(const OraSeamHooks *hooks) {
  if (hooks) g_hooks = *hooks; else memset(&g_hooks, 0, sizeof g_hooks);
}
static void put3(float *d, v3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; }
static int s_intersect(const OraRenderJob *job, const OraRay *ray, float t_min, float t_max, OraRayHit *h) {
  return g_hooks.intersect ? g_hooks.intersect(g_hooks.ctx, ray, t_min, t_max, h) : ora_intersect(job->scene, ray, t_min, t_max, h);
}
static int s_occluded(const OraRenderJob *job, const OraRay *ray, float t_min, float t_max) {
  return g_hooks.occluded ? g_hooks.occluded(g_hooks.ctx, ray, t_min, t_max) : ora_occluded(job->scene, ray, t_min, t_max);
}
static int s_mat_scatter(const OraRenderJob *job, const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec, OraSampler dom,
                         OraScatter *out) {
  if (!g_hooks.mat_scatter) return ora_mat_scatter(m, ray_dir, rec, dom, out);
  float d[3]; put3(d, ray_dir);
  return g_hooks.mat_scatter(g_hooks.ctx, (uint32_t)(m - job->materials), d, rec, dom.pattern, dom.index, out);
}
static int s_mat_eval(const OraRenderJob *job, const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec, v3 wi, v3 *value,
                      float *pdf) {
  if (!g_hooks.mat_eval) return ora_mat_eval(m, ray_dir, rec, wi, value, pdf);
  float d[3], w[3], val[3] = {0.0f, 0.0f, 0.0f}; put3(d, ray_dir); put3(w, wi);
  int some = g_hooks.mat_eval(g_hooks.ctx, (uint32_t)(m - job->materials), d, rec, w, val, pdf);
  *value = v3_new(val[0], val[1], val[2]);
  return some;
}
static v3 s_mat_emitted(const OraRenderJob *job, const OraMaterial *m, float cos_o) {
  if (!g_hooks.mat_emitted) return ora_mat_emitted_directional(m, cos_o);
  float rgb[3] = {0.0f, 0.0f, 0.0f};
  g_hooks.mat_emitted(g_hooks.ctx, (uint32_t)(m - job->materials), cos_o, rgb);
  return v3_new(rgb[0], rgb[1], rgb[2]);
}
static int s_light_sample(const OraRenderJob *job, const OraLight *l, v3 from, float u, float v, OraLightSample *out) {
  if (!g_hooks.light_sample) return ora_light_sample_li(l, from, u, v, out);
  float f[3]; put3(f, from);
  return g_hooks.light_sample(g_hooks.ctx, (uint32_t)(l - job->lights), f, u, v, out);
}
static float s_light_pdf(const OraRenderJob *job, const OraLight *l, v3 from, v3 point) {
  if (!g_hooks.light_pdf) return ora_light_pdf_at_point(l, from, point);
  float f[3], p[3]; put3(f, from); put3(p, point);
  return g_hooks.light_pdf(g_hooks.ctx, (uint32_t)(l - job->lights), f, p);
}
static int s_light_escaped(const OraRenderJob *job, const OraLight *l, v3 direction, v3 *emitted, float *pdf) {
  if (!g_hooks.light_escaped) return ora_light_escaped(l, direction, emitted, pdf);
  float d[3], e[3] = {0.0f, 0.0f, 0.0f}; put3(d, direction);
  int some = g_hooks.light_escaped(g_hooks.ctx, (uint32_t)(l - job->lights), d, e, pdf);
  *emitted = v3_new(e[0], e[1], e[2]);
  return some;
}

/* rt_world.rs:191-196 */
typedef struct { OraHitRecord rec; const OraMaterial *mat; uint32_t geom_id, prim_id; } WorldHit;

static int world_intersect(const OraRenderJob *job, const OraRay *ray, float t_min, float t_max, WorldHit *out) {
  OraRayHit h; /* rt_world.rs:207-232 */
  if (!s_intersect(job, ray, t_min, t_max, &h)) return 0;
  out->rec.p = v3_add(ray->origin, v3_scale(ray->dir, h.t));
  out->rec.normal = h.normal;
  out->rec.t = h.t;
  out->rec.front_face = h.front_face;
  out->mat = &job->materials[h.geom_id];
  out->geom_id = h.geom_id;
  out->prim_id = h.prim_id;
  return 1;
}

/* tracer.rs:899-906 (surface arm of PrevVertex) */
typedef struct { int valid; v3 ray_dir; OraHitRecord rec; const OraMaterial *mat; v3 dir; float pdf; int delta; } PrevBounce;

/* tracer.rs:832-863 */
typedef struct { v3 atten, segment_emit, emit_here, nee, factor, next_emit; float next_emit_weight; } VertexRec;

static const OraLight *find_by_geom(const OraRenderJob *job, uint32_t geom_id) { /* light.rs:436-440 */
  for (uint32_t i = 0; i < job->n_lights; i++)
    if (job->lights[i].geom_id == geom_id) return &job->lights[i];
  return NULL;
}

static float bounce_emission_weight(const OraRenderJob *job, const PrevBounce *p, const WorldHit *hit) { /* tracer.rs:930-953 */
  v3 val; float pdf;
  if (p->delta || !s_mat_eval(job, p->mat, p->ray_dir, &p->rec, p->dir, &val, &pdf)) return 1.0f;
  const OraLight *light = find_by_geom(job, hit->geom_id);
  if (!light) return 1.0f;
  float light_pdf = ora_max(s_light_pdf(job, light, p->rec.p, hit->rec.p) / (float)job->n_lights, 1e-6f);
  return ora_bounce_weight(job->strategy, p->pdf, light_pdf);
}

static v3 sky(v3 unit_direction) { /* tracer.rs:1334-1336 */
  float t = 0.5f * (unit_direction.y + 1.0f);
  return v3_add(v3_scale(v3_new(1.0f, 1.0f, 1.0f), 1.0f - t), v3_scale(v3_new(0.5f, 0.7f, 1.0f), t));
}

#define ORA_MAX_RECORDS 4096

/* tracer.rs:1086-1558: surface arm + carried interior medium (no volume regions, no guiding). */
static v3 trace_path(const OraRenderJob *job, const OraRay *r, OraSampler sampler, VertexRec *records,
                     OraRayStats *stats) {
  const int forward = job->forward;
  OraSampler path = ora_new_domain(sampler, K_PATH);
  uint32_t n_rec = 0;
  OraRay ray = *r;
  int32_t remaining = (int32_t)job->max_depth;
  PrevBounce prev; memset(&prev, 0, sizeof prev);
  v3 beta = v3_splat(1.0f);
  v3 terminal = v3_splat(0.0f);
  v3 L = v3_splat(0.0f); /* forward mode only */
  const v3 one = v3_splat(1.0f);
  OraMedium med; int has_med = 0; /* Ray::medium (ray.rs): the interior the current ray travels in */
  memset(&med, 0, sizeof med);

  for (;;) {
    OraSampler v = ora_new_domain(path, (int)n_rec);
    if (remaining <= 0) { /* tracer.rs:1123-1149 */
      stats->ended_depth++;
      if (prev.valid) {
        stats->closest_hit++;
        WorldHit hit;
        if (world_intersect(job, &ray, 0.001f, ORA_INF, &hit)) {
          float cos_o = ora_abs(v3_dot(v3_normalize(ray.dir), hit.rec.normal));
          v3 emitted = s_mat_emitted(job, hit.mat, cos_o);
          if (v3_len2(emitted) > 0.0f) {
            if (has_med) emitted = v3_mul(emitted, ora_medium_transmittance(&med, hit.rec.t)); /* tracer.rs:1134-1136 */
            float w = bounce_emission_weight(job, &prev, &hit);
            if (forward) L = v3_add(L, v3_mul(beta, v3_scale(emitted, w)));
            else { records[n_rec - 1].next_emit = emitted; records[n_rec - 1].next_emit_weight = w; }
          }
        }
      }
      break;
    }

    stats->closest_hit++;
    WorldHit hit;
    int has_hit = world_intersect(job, &ray, 0.001f, ORA_INF, &hit); /* tracer.rs:1152 */
    const float t_surf = has_hit ? hit.rec.t : ORA_INF;
    /* free-flight candidate in a scattering carried medium (tracer.rs:1159-1165) */
    float t_med = ORA_INF;
    if (has_med && ora_medium_is_scattering(&med)) {
      float sigma_t_max = ora_max(ora_medium_sigma_t_max(&med), 1e-4f);
      t_med = -(ora_logf(ora_draw_rnd1(ora_new_domain(v, K_MEDIUM)))) / sigma_t_max;
    }
    if (t_med < t_surf) { /* === carried-medium scatter vertex (tracer.rs:1256-1319) === */
      float sigma_bar = ora_max(ora_medium_sigma_t_max(&med), 1e-4f);
      v3 pos = v3_add(ray.origin, v3_scale(ray.dir, t_med));
      float phase_uv[4];
      ora_draw_sample4(ora_new_domain(v, K_PHASE), phase_uv);
      v3 dir = ora_sample_henyey_greenstein(v3_normalize(ray.dir), med.g, phase_uv[0], phase_uv[1]);
      v3 e = v3_scale(v3_sub(v3_splat(sigma_bar), v3_add(med.sigma_a, med.sigma_s)), t_med);
      v3 factor = v3_mul(v3_divs(med.sigma_s, sigma_bar), v3_new(ora_expf(e.x), ora_expf(e.y), ora_expf(e.z)));
      VertexRec vrec;
      vrec.atten = one; vrec.segment_emit = v3_splat(0.0f); vrec.emit_here = v3_splat(0.0f); vrec.nee = v3_splat(0.0f);
      vrec.factor = factor; vrec.next_emit = v3_splat(0.0f); vrec.next_emit_weight = 1.0f;
      beta = v3_mul(beta, v3_mul(one, factor));
      int survived = 1;
      if (n_rec >= RR_START_BOUNCE) {
        stats->rr_tested++;
        float p_survive = ora_clamp(v3_max_elem(beta), RR_MIN_PROB, 1.0f);
        if (p_survive < 1.0f) {
          if (ora_draw_rnd1(ora_new_domain(v, K_RR)) >= p_survive) {
            survived = 0;
            stats->rr_killed++;
            vrec.factor = v3_splat(0.0f);
          } else {
            vrec.factor = v3_divs(vrec.factor, p_survive);
            beta = v3_divs(beta, p_survive);
          }
        }
      }
      stats->vertices++;
      if (n_rec >= ORA_MAX_RECORDS) abort();
      records[n_rec++] = vrec;
      if (!survived) break;
      ray.origin = pos; ray.dir = dir; ray.mask = ORA_MASK_INDIRECT; /* same medium, time kept */
      remaining -= 1;
      prev.valid = 0;
      continue;
    }
    if (!has_hit) { /* tracer.rs:1321-1342 */
      stats->ended_escaped++;
      v3 unit_direction = v3_normalize(ray.dir);
      /* escaped_emission (tracer.rs:966-1009): lights at infinity covering the direction, MIS-weighted against
       * the NEE of the previous vertex; area lights never cover an escaping direction. */
      v3 background = v3_splat(0.0f);
      int covered = 0;
      if (job->n_lights > 0) {
        int competing = prev.valid && !prev.delta; /* a vertex that scattered is evaluable: eval() is Some */
        float n_lights = (float)job->n_lights;
        for (uint32_t k = 0; k < job->n_lights; k++) {
          v3 emitted; float pdf;
          if (!s_light_escaped(job, &job->lights[k], unit_direction, &emitted, &pdf)) continue;
          covered = 1;
          float weight = 1.0f;
          if (competing && samples_lights(job->strategy)) {
            float light_pdf = ora_max(pdf / n_lights, 1e-6f);
            weight = ora_bounce_weight(job->strategy, prev.pdf, light_pdf);
          }
          background = v3_add(background, v3_scale(emitted, weight));
        }
      }
      if (!covered) background = v3_add(background, sky(unit_direction));
      if (forward) L = v3_add(L, v3_mul(beta, background));
      else terminal = v3_add(v3_splat(0.0f), v3_mul(one, background)); /* vol_emit + vol_tr * background */
      break;
    }
    const OraHitRecord rec = hit.rec;
    const OraMaterial *mat = hit.mat;
    /* tracer.rs:1352-1361: a scattering medium already paid e^{-sigma_bar t} through the free-flight
     * competition, so only the chromatic correction remains; a clear one keeps pure Beer-Lambert. */
    v3 atten = one;
    if (has_med) {
      if (ora_medium_is_scattering(&med)) {
        float sigma_bar = ora_max(ora_medium_sigma_t_max(&med), 1e-4f);
        v3 e = v3_scale(v3_sub(v3_splat(sigma_bar), v3_add(med.sigma_a, med.sigma_s)), rec.t);
        atten = v3_mul(one, v3_new(ora_expf(e.x), ora_expf(e.y), ora_expf(e.z)));
      } else {
        atten = v3_mul(one, ora_medium_transmittance(&med, rec.t));
      }
    }

    /* tracer.rs:1369-1381 */
    float cos_o = ora_abs(v3_dot(v3_normalize(ray.dir), rec.normal));
    v3 emitted = s_mat_emitted(job, mat, cos_o);
    v3 emit_here = v3_splat(0.0f);
    if (prev.valid) {
      if (v3_len2(emitted) > 0.0f) {
        float w = bounce_emission_weight(job, &prev, &hit);
        if (forward) L = v3_add(L, v3_mul(beta, v3_scale(v3_mul(atten, emitted), w)));
        else { records[n_rec - 1].next_emit = v3_mul(atten, emitted); records[n_rec - 1].next_emit_weight = w; }
      }
    } else {
      emit_here = emitted;
    }

    /* === 1. NEE (tracer.rs:1394-1445) === */
    v3 nee = v3_splat(0.0f);
    float nee_s[4];
    ora_draw_sample4(ora_new_domain(v, K_NEE), nee_s);
    if (samples_lights(job->strategy) && job->n_lights > 0) {
      /* light.rs:421-429 pick */
      size_t li = (size_t)(nee_s[0] * (float)job->n_lights);
      if (li > job->n_lights - 1) li = job->n_lights - 1;
      const OraLight *light = &job->lights[li];
      OraLightSample ls;
      if (s_light_sample(job, light, rec.p, nee_s[1], nee_s[2], &ls)) {
        float n_lights = (float)job->n_lights;
        OraRay shadow_ray = {rec.p, ls.direction, ray.time, ORA_MASK_SHADOW};
        stats->shadow_rays++; /* tracer.rs:1026 */
        int occluded = s_occluded(job, &shadow_ray, 0.001f, ls.distance - 0.001f);
        if (!occluded) {
          float cosine = ora_abs(v3_dot(rec.normal, ls.direction));
          float light_pdf = ora_max(ls.pdf / n_lights, 1e-6f);
          v3 brdf_value; float brdf_pdf;
          if (s_mat_eval(job, mat, ray.dir, &rec, ls.direction, &brdf_value, &brdf_pdf)) {
            float weight = ora_light_weight(job->strategy, light_pdf, brdf_pdf);
            v3 c = v3_scale(v3_mul(ls.radiance, brdf_value), cosine);
            c = v3_mul(c, one); /* shadow_tr == ONE */
            nee = v3_add(nee, v3_divs(v3_scale(c, weight), light_pdf));
          }
        }
      }
    }

    VertexRec vrec;
    vrec.atten = atten; vrec.segment_emit = v3_splat(0.0f); vrec.emit_here = emit_here; vrec.nee = nee;
    vrec.factor = v3_splat(0.0f); vrec.next_emit = v3_splat(0.0f); vrec.next_emit_weight = 1.0f;
    if (forward) { /* two separate adds: the second is what a deferred shadow test contributes */
      const v3 ba = v3_mul(beta, atten);
      L = v3_add(L, v3_mul(ba, emit_here));
      L = v3_add(L, v3_mul(ba, nee));
    }

    /* === 2. bounce (tracer.rs:1459-1523) === */
    OraScatter sample;
    if (s_mat_scatter(job, mat, ray.dir, &rec, ora_new_domain(v, K_BSDF), &sample)) {
      v3 dir = v3_normalize(sample.dir);
      float cosine = sample.delta ? 1.0f : ora_abs(v3_dot(rec.normal, dir));
      v3 factor = v3_divs(v3_scale(sample.value, cosine), sample.pdf);
      beta = v3_mul(beta, v3_mul(atten, factor));
      int survived = 1;
      if (n_rec >= RR_START_BOUNCE) {
        stats->rr_tested++;
        float p_survive = ora_clamp(v3_max_elem(beta), RR_MIN_PROB, 1.0f);
        if (p_survive < 1.0f) {
          if (ora_draw_rnd1(ora_new_domain(v, K_RR)) >= p_survive) {
            survived = 0;
            stats->rr_killed++;
          } else {
            factor = v3_divs(factor, p_survive);
            beta = v3_divs(beta, p_survive);
          }
        }
      }
      if (survived) {
        vrec.factor = factor;
        prev.valid = 1; prev.ray_dir = ray.dir; prev.rec = rec; prev.mat = mat; prev.dir = dir; prev.pdf = sample.pdf;
        prev.delta = sample.delta;
        stats->vertices++;
        if (n_rec >= ORA_MAX_RECORDS) abort();
        records[n_rec++] = vrec;
        ray.origin = sample.origin; ray.dir = sample.dir; ray.mask = ORA_MASK_INDIRECT; /* time kept */
        has_med = sample.medium ? ora_interior_medium(mat, &med) : 0; /* materials build the ray: openpbr.rs:1061-1066 */
        remaining -= 1;
        continue;
      }
    }
    stats->vertices++;
    if (n_rec >= ORA_MAX_RECORDS) abort();
    records[n_rec++] = vrec;
    break;
  }
  if (forward) return L;

  v3 radiance = terminal; /* tracer.rs:1537-1557 */
  for (uint32_t k = n_rec; k-- > 0;) {
    const VertexRec *vr = &records[k];
    v3 inner = v3_add(v3_scale(vr->next_emit, vr->next_emit_weight), radiance);
    v3 mid = v3_add(v3_add(vr->emit_here, vr->nee), v3_mul(vr->factor, inner));
    radiance = v3_add(vr->segment_emit, v3_mul(vr->atten, mid));
  }
  return radiance;
}

static float luminance(v3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; } /* guiding/mod.rs:18-20 */

static v3 camera_sample(const OraRenderJob *job, uint32_t i, uint32_t j, uint32_t sample, VertexRec *records,
                        OraRayStats *stats, float *w_out) { /* tracer.rs:559-598 */
  int tile = (int)(i >> 8) + (int)(j >> 8) * 4096; /* tracer.rs:543 */
  OraSampler root = ora_new_domain(ora_sampler_new((int)i, (int)j, job->frame, (int)sample), tile);
  float cam[4];
  ora_draw_sample4(ora_new_domain(root, K_CAMERA), cam);
  float fx, wx, fy, wy;
  ora_filter_sample(job->filter_kind, job->filter_radius, cam[0], &fx, &wx);
  ora_filter_sample(job->filter_kind, job->filter_radius, cam[1], &fy, &wy);
  float u = ((float)i + fx) / (float)job->width;
  float v = ((float)j + fy) / (float)job->height;
  float time = 0.0f;
  if (ora_has_motion(job->scene)) { /* tracer.rs:579-583 */
    float t4[4];
    ora_draw_sample4(ora_new_domain(root, K_TIME), t4);
    time = t4[0];
  }
  OraRay r;
  ora_camera_get_ray(&job->camera, u, v, cam[2], cam[3], time, &r);
  stats->camera_rays++;
  v3 color = trace_path(job, &r, root, records, stats);
  *w_out = wx * wy;
  return color;
}

void ora_render_sample(const OraRenderJob *job, uint32_t i, uint32_t j, uint32_t sample, float rgb[3],
                       OraRayStats *stats) {
  VertexRec *records = (VertexRec *)malloc(sizeof(VertexRec) * ORA_MAX_RECORDS);
  float w;
  v3 c = camera_sample(job, i, j, sample, records, stats, &w);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
  free(records);
}

static uint32_t render_pixel(const OraRenderJob *job, uint32_t i, uint32_t j, VertexRec *records, float rgb[3],
                             OraRayStats *stats) { /* tracer.rs:515-636 */
  v3 sum = v3_splat(0.0f);
  float weight_sum = 0.0f;
  double lum_sum = 0.0, lum_sq = 0.0;
  double threshold = (double)job->variance_threshold;
  uint32_t min_spp = job->min_spp > 2 ? job->min_spp : 2;
  uint32_t taken = 0;
  for (uint32_t sample = 0; sample < job->spp; sample++) {
    float w = 0.0f;
    v3 radiance = camera_sample(job, i, j, sample, records, stats, &w);
    v3 color = v3_scale(radiance, w);
    sum = v3_add(sum, color);
    weight_sum += w;
    double lum = (double)luminance(color);
    lum_sum += lum;
    lum_sq += lum * lum;
    taken = sample + 1;
    if (threshold > 0.0 && taken >= min_spp && taken % 4 == 0) { /* tracer.rs:609-617 (final pass: adaptive) */
      double n = (double)taken;
      double var_of_mean = (lum_sq - lum_sum * lum_sum / n) / (n - 1.0) / n;
      if (!(var_of_mean > 0.0)) var_of_mean = 0.0;
      double mean = lum_sum / n;
      if (!(mean > 1e-4)) mean = 1e-4;
      if (sqrt(var_of_mean) / mean < threshold) break;
    }
  }
  v3 mean = weight_sum > 0.0f ? v3_divs(sum, weight_sum) : v3_divs(sum, (float)taken);
  rgb[0] = mean.x; rgb[1] = mean.y; rgb[2] = mean.z;
  return taken;
}

uint32_t ora_render_pixel(const OraRenderJob *job, uint32_t i, uint32_t j, float rgb[3], OraRayStats *stats) {
  VertexRec *records = (VertexRec *)malloc(sizeof(VertexRec) * ORA_MAX_RECORDS);
  uint32_t taken = render_pixel(job, i, j, records, rgb, stats);
  free(records);
  return taken;
}

typedef struct {
  const OraRenderJob *job; float *rgb; OraRayStats stats;
  uint32_t *next_tile; pthread_mutex_t *mu; uint32_t tiles_x, tiles_y;
} Worker;

static void stats_merge(OraRayStats *a, const OraRayStats *b) { /* stats.rs:175-184 */
  a->camera_rays += b->camera_rays; a->closest_hit += b->closest_hit; a->shadow_rays += b->shadow_rays;
  a->vertices += b->vertices; a->rr_tested += b->rr_tested; a->rr_killed += b->rr_killed;
  a->ended_escaped += b->ended_escaped; a->ended_depth += b->ended_depth;
}

static void *worker_main(void *arg) {
  Worker *w = (Worker *)arg;
  const OraRenderJob *job = w->job;
  VertexRec *records = (VertexRec *)malloc(sizeof(VertexRec) * ORA_MAX_RECORDS);
  for (;;) {
    pthread_mutex_lock(w->mu);
    uint32_t t = (*w->next_tile)++;
    pthread_mutex_unlock(w->mu);
    if (t >= w->tiles_x * w->tiles_y) break;
    uint32_t tx = (t % w->tiles_x) * 16, ty = (t / w->tiles_x) * 16; /* tracer.rs:1671-1686, tile size 16 */
    uint32_t x1 = tx + 16 < job->width ? tx + 16 : job->width;
    uint32_t y1 = ty + 16 < job->height ? ty + 16 : job->height;
    for (uint32_t j = ty; j < y1; j++)
      for (uint32_t i = tx; i < x1; i++)
        render_pixel(job, i, j, records, w->rgb + 3 * ((size_t)j * job->width + i), &w->stats);
  }
  free(records);
  return NULL;
}

/* The whole frame on the CALLING thread, with the traversal counters of the two query kinds kept apart: the stats sink
 * of ora_rt.c is thread-local, and World::intersect (closest hit) and World::occluded (any hit) are separate kernels on
 * the device (k_extend / k_shadow). closest / any: zeroed by the caller, either may be NULL. */
void ora_render_serial_trav(const OraRenderJob *job, float *rgb, OraRayStats *stats, OraTravStats *closest, OraTravStats *any) {
  VertexRec *records = (VertexRec *)malloc(sizeof(VertexRec) * ORA_MAX_RECORDS);
  OraRayStats total; memset(&total, 0, sizeof total);
  ora_set_trav_stats(closest);
  ora_set_trav_stats_any(any);
  for (uint32_t j = 0; j < job->height; j++)
    for (uint32_t i = 0; i < job->width; i++)
      render_pixel(job, i, j, records, rgb + 3 * ((size_t)j * job->width + i), &total);
  ora_set_trav_stats(NULL);
  ora_set_trav_stats_any(NULL);
  if (stats) *stats = total;
  free(records);
}

void ora_render(const OraRenderJob *job, float *rgb, OraRayStats *stats, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  uint32_t next = 0;
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  Worker *ws = (Worker *)calloc((size_t)n_threads, sizeof(Worker));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int k = 0; k < n_threads; k++) {
    ws[k].job = job; ws[k].rgb = rgb; ws[k].next_tile = &next; ws[k].mu = &mu;
    ws[k].tiles_x = (job->width + 15) / 16; ws[k].tiles_y = (job->height + 15) / 16;
    pthread_create(&th[k], NULL, worker_main, &ws[k]);
  }
  OraRayStats total; memset(&total, 0, sizeof total);
  for (int k = 0; k < n_threads; k++) { pthread_join(th[k], NULL); stats_merge(&total, &ws[k].stats); }
  if (stats) *stats = total;
  free(ws); free(th);
}

/* A list of pixels (linear buffer indices j*width+i), n_threads workers taking 64-pixel chunks: rgb[3*k] is the
 * mean of pixel idx[k], exactly what ora_render writes at that pixel. For frames too large to render whole in a
 * test (3840x2160 at the bench's 64 spp): per-pixel independence (tracer.rs:543, :559-560) makes any subset exact. */
typedef struct {
  const OraRenderJob *job; const uint32_t *idx; size_t n; float *rgb; OraRayStats stats;
  size_t *next; pthread_mutex_t *mu;
} PixWorker;

static void *pix_worker_main(void *arg) {
  PixWorker *w = (PixWorker *)arg;
  VertexRec *records = (VertexRec *)malloc(sizeof(VertexRec) * ORA_MAX_RECORDS);
  for (;;) {
    pthread_mutex_lock(w->mu);
    size_t k0 = *w->next;
    *w->next = k0 + 64;
    pthread_mutex_unlock(w->mu);
    if (k0 >= w->n) break;
    size_t k1 = k0 + 64 < w->n ? k0 + 64 : w->n;
    for (size_t k = k0; k < k1; k++)
      render_pixel(w->job, w->idx[k] % w->job->width, w->idx[k] / w->job->width, records, w->rgb + 3 * k, &w->stats);
  }
  free(records);
  return NULL;
}

void ora_render_pixels(const OraRenderJob *job, const uint32_t *idx, size_t n, float *rgb, OraRayStats *stats,
                       int n_threads) {
  if (n_threads < 1) n_threads = 1;
  size_t next = 0;
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  PixWorker *ws = (PixWorker *)calloc((size_t)n_threads, sizeof(PixWorker));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int k = 0; k < n_threads; k++) {
    ws[k].job = job; ws[k].idx = idx; ws[k].n = n; ws[k].rgb = rgb; ws[k].next = &next; ws[k].mu = &mu;
    pthread_create(&th[k], NULL, pix_worker_main, &ws[k]);
  }
  OraRayStats total; memset(&total, 0, sizeof total);
  for (int k = 0; k < n_threads; k++) { pthread_join(th[k], NULL); stats_merge(&total, &ws[k].stats); }
  if (stats) *stats = total;
  free(ws); free(th);
}
