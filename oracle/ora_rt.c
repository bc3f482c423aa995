/* ORACLE — TEST INFRASTRUCTURE ONLY (see ora_math.h header).
 *
 * CPU restatement of crates/crust-rt/src/{aabb,triangle,prim,bvh,scene}.rs.
 * Plain scalar C, one ray at a time, compiled -ffp-contract=off. The 4-wide
 * Vec4 code of the reference is restated lane by lane: every lane performs
 * the same IEEE operations in the same order as the SSE2 lanes do.
 */
#include "ora_rt.h"
#include <stdlib.h>

/* ------------------------------------------------------------------ */
/* small containers                                                    */
/* ------------------------------------------------------------------ */
#define VEC_PUSH(arr, n, cap, T, val)                              \
  do {                                                             \
    if ((n) == (cap)) {                                            \
      (cap) = (cap) ? (cap) * 2 : 16;                              \
      (arr) = (T *)realloc((arr), (cap) * sizeof(T));              \
    }                                                              \
    (arr)[(n)++] = (val);                                          \
  } while (0)

/* ------------------------------------------------------------------ */
/* affine helpers (glam Affine3A / Mat3A, SSE2 code paths)             */
/* ------------------------------------------------------------------ */
static v3 mat3_mul_v(const OraMat3 *m, v3 r) {
  v3 res = v3_scale(m->x, r.x);
  res = v3_add(res, v3_scale(m->y, r.y));
  res = v3_add(res, v3_scale(m->z, r.z));
  return res;
}
static v3 affine_point(const OraAffine *a, v3 p) {
  OraMat3 m = {a->x, a->y, a->z};
  return v3_add(mat3_mul_v(&m, p), a->t);
}
static v3 affine_vector(const OraAffine *a, v3 p) {
  OraMat3 m = {a->x, a->y, a->z};
  return mat3_mul_v(&m, p);
}
static OraMat3 mat3_transpose(const OraMat3 *m) {
  OraMat3 r;
  r.x = v3_new(m->x.x, m->y.x, m->z.x);
  r.y = v3_new(m->x.y, m->y.y, m->z.y);
  r.z = v3_new(m->x.z, m->y.z, m->z.z);
  return r;
}
/* glam Mat3A::inverse: cross products scaled by 1/det, transposed. */
static OraMat3 mat3_inverse(const OraMat3 *m) {
  v3 tmp0 = v3_cross(m->y, m->z);
  v3 tmp1 = v3_cross(m->z, m->x);
  v3 tmp2 = v3_cross(m->x, m->y);
  float det = v3_dot(m->z, tmp2);
  float inv_det = 1.0f / det;
  OraMat3 c = {v3_scale(tmp0, inv_det), v3_scale(tmp1, inv_det), v3_scale(tmp2, inv_det)};
  return mat3_transpose(&c);
}
static OraAffine affine_inverse(const OraAffine *a) {
  OraMat3 m = {a->x, a->y, a->z};
  OraMat3 inv = mat3_inverse(&m);
  v3 t = v3_neg(mat3_mul_v(&inv, a->t));
  OraAffine r = {inv.x, inv.y, inv.z, t};
  return r;
}
static OraAffine affine_from12(const float m[12]) {
  OraAffine a = {v3_new(m[0], m[1], m[2]), v3_new(m[3], m[4], m[5]), v3_new(m[6], m[7], m[8]),
                 v3_new(m[9], m[10], m[11])};
  return a;
}
void ora_affine_inverse(const float m[12], float out[12]) {
  OraAffine a = affine_from12(m);
  OraAffine r = affine_inverse(&a);
  float o[12] = {r.x.x, r.x.y, r.x.z, r.y.x, r.y.y, r.y.z, r.z.x, r.z.y, r.z.z, r.t.x, r.t.y, r.t.z};
  memcpy(out, o, sizeof o);
}
/* prim.rs:285-294 lerp_affine */
static OraAffine lerp_affine(const OraAffine *a, const OraAffine *b, float t) {
  OraAffine r = {v3_lerp(a->x, b->x, t), v3_lerp(a->y, b->y, t), v3_lerp(a->z, b->z, t), v3_lerp(a->t, b->t, t)};
  return r;
}

/* ------------------------------------------------------------------ */
/* aabb.rs                                                             */
/* ------------------------------------------------------------------ */
static OraAabb aabb_union(OraAabb a, OraAabb b) { /* aabb.rs:15-20 */
  OraAabb r = {v3_min(a.mn, b.mn), v3_max(a.mx, b.mx)};
  return r;
}
static OraAabb triangle_aabb(v3 v0, v3 v1, v3 v2) { /* aabb.rs:47-58 */
  v3 mn = v3_min(v3_min(v0, v1), v2);
  v3 mx = v3_max(v3_max(v0, v1), v2);
  const float PAD = 1e-4f;
  for (int a = 0; a < 3; a++) {
    if (v3_get(mx, a) - v3_get(mn, a) < PAD) {
      v3_set(&mn, a, v3_get(mn, a) - PAD);
      v3_set(&mx, a, v3_get(mx, a) + PAD);
    }
  }
  OraAabb r = {mn, mx};
  return r;
}
/* prim.rs:298-319 transformed_aabb */
static OraAabb transformed_aabb(const OraAabb *local, const OraAffine *m) {
  v3 mn = v3_splat(ORA_INF), mx = v3_splat(-ORA_INF);
  for (int i = 0; i < 8; i++) {
    v3 corner = v3_new((i & 1) == 0 ? local->mn.x : local->mx.x, (i & 2) == 0 ? local->mn.y : local->mx.y,
                       (i & 4) == 0 ? local->mn.z : local->mx.z);
    v3 p = affine_point(m, corner);
    mn = v3_min(mn, p);
    mx = v3_max(mx, p);
  }
  const float PAD = 1e-4f;
  for (int a = 0; a < 3; a++) {
    if (v3_get(mx, a) - v3_get(mn, a) < PAD) {
      v3_set(&mn, a, v3_get(mn, a) - PAD);
      v3_set(&mx, a, v3_get(mx, a) + PAD);
    }
  }
  OraAabb r = {mn, mx};
  return r;
}

/* ------------------------------------------------------------------ */
/* triangle.rs                                                         */
/* ------------------------------------------------------------------ */
typedef struct { int kx, ky, kz; float sx, sy, sz; } RayShear; /* triangle.rs:23-38 */

static RayShear shear_new(const OraRay *ray) { /* triangle.rs:41-80 */
  RayShear s;
  v3 d = ray->dir;
  float adx = ora_abs(d.x), ady = ora_abs(d.y), adz = ora_abs(d.z);
  int kz;
  if (adx > ady) kz = (adx > adz) ? 0 : 2;
  else if (ady > adz) kz = 1;
  else kz = 2;
  int kx = (kz + 1) % 3, ky = (kz + 2) % 3;
  if (v3_get(d, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
  s.kx = kx; s.ky = ky; s.kz = kz;
  s.sx = v3_get(d, kx) / v3_get(d, kz);
  s.sy = v3_get(d, ky) / v3_get(d, kz);
  s.sz = 1.0f / v3_get(d, kz);
  return s;
}

/* triangle.rs:110-172 triangle_intersect_sheared */
static int tri_intersect_sheared(const RayShear *sh, v3 origin, v3 v0, v3 v1, v3 v2, float t_min, float t_max,
                                 float *to, float *uo, float *vo) {
  int kx = sh->kx, ky = sh->ky, kz = sh->kz;
  float sx = sh->sx, sy = sh->sy, sz = sh->sz;
  v3 a = v3_sub(v0, origin), b = v3_sub(v1, origin), c = v3_sub(v2, origin);
  float ax = v3_get(a, kx) - sx * v3_get(a, kz);
  float ay = v3_get(a, ky) - sy * v3_get(a, kz);
  float bx = v3_get(b, kx) - sx * v3_get(b, kz);
  float by = v3_get(b, ky) - sy * v3_get(b, kz);
  float cx = v3_get(c, kx) - sx * v3_get(c, kz);
  float cy = v3_get(c, ky) - sy * v3_get(c, kz);
  float e0 = bx * cy - by * cx;
  float e1 = cx * ay - cy * ax;
  float e2 = ax * by - ay * bx;
  if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
    e0 = (float)((double)bx * (double)cy - (double)by * (double)cx);
    e1 = (float)((double)cx * (double)ay - (double)cy * (double)ax);
    e2 = (float)((double)ax * (double)by - (double)ay * (double)bx);
  }
  if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return 0;
  float det = e0 + e1 + e2;
  if (det == 0.0f) return 0;
  float az = sz * v3_get(a, kz), bz = sz * v3_get(b, kz), cz = sz * v3_get(c, kz);
  float t_scaled = e0 * az + e1 * bz + e2 * cz;
  if (det < 0.0f && (t_scaled > t_min * det || t_scaled < t_max * det)) return 0;
  if (det > 0.0f && (t_scaled < t_min * det || t_scaled > t_max * det)) return 0;
  float inv_det = 1.0f / det;
  *to = t_scaled * inv_det;
  *uo = e1 * inv_det;
  *vo = e2 * inv_det;
  return 1;
}

static int tri_intersect(const OraRay *ray, v3 v0, v3 v1, v3 v2, float t_min, float t_max, float *t, float *u,
                         float *v) { /* triangle.rs:97-106 */
  RayShear sh = shear_new(ray);
  return tri_intersect_sheared(&sh, ray->origin, v0, v1, v2, t_min, t_max, t, u, v);
}
int ora_triangle_intersect(const OraRay *ray, const float v0[3], const float v1[3], const float v2[3], float t_min,
                           float t_max, float tuv[3]) {
  return tri_intersect(ray, v3_new(v0[0], v0[1], v0[2]), v3_new(v1[0], v1[1], v1[2]), v3_new(v2[0], v2[1], v2[2]),
                       t_min, t_max, &tuv[0], &tuv[1], &tuv[2]);
}

typedef struct { v3 v0, v1, v2; uint32_t pi, mask; } TriRef;

static void tri4_new(OraTri4 *p, const TriRef *tris, int n) { /* triangle.rs:217-253 */
  memset(p, 0, sizeof *p);
  p->mask_and = 0xffffffffu;
  for (int lane = 0; lane < 4; lane++) {
    const TriRef *t = &tris[lane < n - 1 ? lane : n - 1];
    for (int axis = 0; axis < 3; axis++) {
      p->v[0][axis][lane] = v3_get(t->v0, axis);
      p->v[1][axis][lane] = v3_get(t->v1, axis);
      p->v[2][axis][lane] = v3_get(t->v2, axis);
    }
    p->prim[lane] = 0xffffffffu;
    if (lane < n) {
      p->prim[lane] = t->pi;
      p->masks[lane] = t->mask;
      p->active |= 1u << lane;
      p->mask_and &= t->mask;
      p->mask_or |= t->mask;
    }
  }
}

static uint32_t tri4_visible(const OraTri4 *p, uint32_t ray_mask) { /* triangle.rs:257-271 */
  if (ray_mask & p->mask_and) return p->active;
  if ((ray_mask & p->mask_or) == 0) return 0;
  uint32_t m = 0;
  for (int lane = 0; lane < 4; lane++)
    if ((p->active & (1u << lane)) && (p->masks[lane] & ray_mask)) m |= 1u << lane;
  return m;
}

typedef struct { uint32_t hits, fallback; float t[4], u[4], v[4]; } Hit4;

/* triangle.rs:276-348 Tri4::intersect, lane by lane. */
static Hit4 tri4_intersect(const OraTri4 *p, const RayShear *sh, v3 origin, uint32_t ray_mask, float t_min,
                           float t_max) {
  Hit4 h;
  memset(&h, 0, sizeof h);
  uint32_t m = tri4_visible(p, ray_mask);
  if (m == 0) return h;
  int kx = sh->kx, ky = sh->ky, kz = sh->kz;
  float okx = v3_get(origin, kx), oky = v3_get(origin, ky), okz = v3_get(origin, kz);
  uint32_t fallback = 0, negpos = 0, detz = 0, range = 0;
  float ts_[4], e1_[4], e2_[4], det_[4];
  for (int l = 0; l < 4; l++) {
    float akz = p->v[0][kz][l] - okz;
    float bkz = p->v[1][kz][l] - okz;
    float ckz = p->v[2][kz][l] - okz;
    float ax = (p->v[0][kx][l] - okx) - sh->sx * akz;
    float ay = (p->v[0][ky][l] - oky) - sh->sy * akz;
    float bx = (p->v[1][kx][l] - okx) - sh->sx * bkz;
    float by = (p->v[1][ky][l] - oky) - sh->sy * bkz;
    float cx = (p->v[2][kx][l] - okx) - sh->sx * ckz;
    float cy = (p->v[2][ky][l] - oky) - sh->sy * ckz;
    float e0 = bx * cy - by * cx;
    float e1 = cx * ay - cy * ax;
    float e2 = ax * by - ay * bx;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) fallback |= 1u << l;
    int neg = (e0 < 0.0f) || (e1 < 0.0f) || (e2 < 0.0f);
    int pos = (e0 > 0.0f) || (e1 > 0.0f) || (e2 > 0.0f);
    if (neg && pos) negpos |= 1u << l;
    float det = e0 + e1 + e2;
    if (det == 0.0f) detz |= 1u << l;
    float t_scaled = e0 * (sh->sz * akz) + e1 * (sh->sz * bkz) + e2 * (sh->sz * ckz);
    float abs_det = ora_abs(det);
    float ts = det < 0.0f ? -t_scaled : t_scaled;
    if (ts >= t_min * abs_det && ts <= t_max * abs_det) range |= 1u << l;
    ts_[l] = t_scaled; e1_[l] = e1; e2_[l] = e2; det_[l] = det;
  }
  fallback &= m;
  m &= ~fallback;
  m &= ~negpos;
  m &= ~detz;
  h.fallback = fallback;
  if (m == 0) return h;
  m &= range;
  if (m == 0) return h;
  h.hits = m;
  for (int l = 0; l < 4; l++) {
    float inv_det = 1.0f / det_[l];
    h.t[l] = ts_[l] * inv_det;
    h.u[l] = e1_[l] * inv_det;
    h.v[l] = e2_[l] * inv_det;
  }
  return h;
}

void ora_tri4_intersect(const OraRay *ray, const float *tris, const uint32_t *masks, int n, uint32_t ray_mask,
                        float t_min, float t_max, uint32_t *hits, uint32_t *fallback, uint32_t *active, float t[4],
                        float u[4], float v[4]) {
  TriRef refs[4];
  for (int i = 0; i < n; i++) {
    const float *f = tris + 9 * i;
    refs[i].v0 = v3_new(f[0], f[1], f[2]);
    refs[i].v1 = v3_new(f[3], f[4], f[5]);
    refs[i].v2 = v3_new(f[6], f[7], f[8]);
    refs[i].pi = (uint32_t)i;
    refs[i].mask = masks[i];
  }
  OraTri4 p;
  tri4_new(&p, refs, n);
  RayShear sh = shear_new(ray);
  Hit4 h = tri4_intersect(&p, &sh, ray->origin, ray_mask, t_min, t_max);
  *hits = h.hits; *fallback = h.fallback; *active = p.active;
  memcpy(t, h.t, sizeof h.t); memcpy(u, h.u, sizeof h.u); memcpy(v, h.v, sizeof h.v);
}

/* triangle.rs:367-429 clip_triangle_aabb */
static int clip_triangle_aabb(v3 v0, v3 v1, v3 v2, int axis, float mn, float mx, OraAabb *out) {
  v3 poly[8];
  int n = 3;
  poly[0] = v0; poly[1] = v1; poly[2] = v2;
  for (int pass = 0; pass < 2; pass++) {
    float bound = pass == 0 ? mn : mx;
    int keep_ge = pass == 0;
    v3 o[8];
    int m = 0;
    for (int i = 0; i < n; i++) {
      v3 a = poly[i], b = poly[(i + 1) % n];
      float da, db;
      if (keep_ge) { da = v3_get(a, axis) - bound; db = v3_get(b, axis) - bound; }
      else { da = bound - v3_get(a, axis); db = bound - v3_get(b, axis); }
      if (da >= 0.0f) o[m++] = a;
      if ((da > 0.0f) != (db > 0.0f) && da != db) {
        float t = da / (da - db);
        o[m++] = v3_add(a, v3_scale(v3_sub(b, a), t));
      }
    }
    memcpy(poly, o, sizeof o);
    n = m;
    if (n == 0) return 0;
  }
  v3 lo = poly[0], hi = poly[0];
  for (int i = 1; i < n; i++) { lo = v3_min(lo, poly[i]); hi = v3_max(hi, poly[i]); }
  v3_set(&lo, axis, ora_max(v3_get(lo, axis), mn));
  v3_set(&hi, axis, ora_min(v3_get(hi, axis), mx));
  const float PAD = 1e-4f;
  for (int a = 0; a < 3; a++) {
    if (v3_get(hi, a) - v3_get(lo, a) < PAD) {
      v3_set(&lo, a, v3_get(lo, a) - PAD);
      v3_set(&hi, a, v3_get(hi, a) + PAD);
    }
  }
  out->mn = lo; out->mx = hi;
  return 1;
}
int ora_clip_triangle_aabb(const float v0[3], const float v1[3], const float v2[3], int axis, float mn, float mx,
                           float out[6]) {
  OraAabb b;
  if (!clip_triangle_aabb(v3_new(v0[0], v0[1], v0[2]), v3_new(v1[0], v1[1], v1[2]), v3_new(v2[0], v2[1], v2[2]), axis,
                          mn, mx, &b))
    return 0;
  out[0] = b.mn.x; out[1] = b.mn.y; out[2] = b.mn.z; out[3] = b.mx.x; out[4] = b.mx.y; out[5] = b.mx.z;
  return 1;
}

/* ------------------------------------------------------------------ */
/* prim.rs                                                             */
/* ------------------------------------------------------------------ */
enum { PRIM_TRI = 0, PRIM_SPHERE = 1, PRIM_INSTANCE = 2 };

typedef struct {
  int kind;
  uint32_t geom_id, prim_id, mask;
  /* triangle (prim.rs:60-70) */
  v3 v0, v1, v2;
  int has_normals;
  v3 n0, n1, n2;
  /* sphere (prim.rs:125-130) */
  v3 center;
  float radius;
  /* instance (prim.rs:261-280) */
  OraScene *scene;
  OraAffine l2w, w2l;
  OraMat3 normal_mat;
  int has_end;
  OraAffine l2w_end;
  OraAabb bounds;
} Prim;

typedef struct {
  OraWideNode *wide; size_t n_wide;
  OraLeaf *leaves; size_t n_leaves;
  OraTri4 *packets; size_t n_packets;
  uint32_t *indices; size_t n_indices;
  Prim *prims; size_t n_prims;
  int has_bbox;
  OraAabb root_bbox;
} Bvh;

struct OraScene { Bvh bvh; uint32_t n_geoms; int has_motion; };

static __thread OraTravStats *g_stats = NULL;
static __thread int g_depth = 0;
void ora_set_trav_stats(OraTravStats *st) { g_stats = st; }
static __thread OraTravStats *g_stats_any = NULL;
void ora_set_trav_stats_any(OraTravStats *st) { g_stats_any = st; }
#define TSTAT(field, n) do { if (g_stats) g_stats->field[g_depth > 0] += (n); } while (0)
/* Optional per-ray phase trace for scheduling studies (profiles/simulate_scheduling.py): one char per unit of
 * traversal work in execution order — N node, P packet, s sphere, I instance entry, X instance exit. */
static __thread char *g_trace = NULL;
static __thread size_t g_trace_cap = 0, g_trace_len = 0;
void ora_set_trace(char *buf, size_t cap) { g_trace = buf; g_trace_cap = cap; g_trace_len = 0; }
size_t ora_trace_len(void) { return g_trace_len; }
#define TRACE(c) do { if (g_trace && g_trace_len < g_trace_cap) g_trace[g_trace_len++] = (c); } while (0)

static int bvh_hit(const Bvh *b, const OraRay *ray, float t_min, float t_max, OraPrimHit *out);
static int bvh_hit_any(const Bvh *b, const OraRay *ray, float t_min, float t_max);

/* prim.rs:76-95 */
static int tri_hit_from_bary(const Prim *p, float t, float u, float v, OraPrimHit *out) {
  v3 outward;
  if (p->has_normals) {
    v3 n = v3_add(v3_add(v3_scale(p->n0, 1.0f - u - v), v3_scale(p->n1, u)), v3_scale(p->n2, v));
    outward = v3_normalize(n);
  } else {
    v3 n = v3_cross(v3_sub(p->v1, p->v0), v3_sub(p->v2, p->v0));
    if (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f) return 0;
    outward = v3_normalize(n);
  }
  out->t = t; out->outward = outward; out->u = u; out->v = v;
  out->geom_id = p->geom_id; out->prim_id = p->prim_id;
  return 1;
}

static OraAabb prim_bbox(const Prim *p) {
  switch (p->kind) {
    case PRIM_TRI: return triangle_aabb(p->v0, p->v1, p->v2); /* prim.rs:112-114 */
    case PRIM_SPHERE: { /* prim.rs:163-168 */
      OraAabb r = {v3_sub(p->center, v3_splat(p->radius)), v3_add(p->center, v3_splat(p->radius))};
      return r;
    }
    default: return p->bounds;
  }
}

/* prim.rs:39-48 default clipped_aabb; :116-118 triangle override */
static int prim_clipped_aabb(const Prim *p, int axis, float mn, float mx, OraAabb *out) {
  if (p->kind == PRIM_TRI) return clip_triangle_aabb(p->v0, p->v1, p->v2, axis, mn, mx, out);
  OraAabb b = prim_bbox(p);
  if (v3_get(b.mn, axis) > mx || v3_get(b.mx, axis) < mn) return 0;
  v3_set(&b.mn, axis, ora_max(v3_get(b.mn, axis), mn));
  v3_set(&b.mx, axis, ora_min(v3_get(b.mx, axis), mx));
  *out = b;
  return 1;
}

static v3 ray_at(const OraRay *r, float t) { return v3_add(r->origin, v3_scale(r->dir, t)); } /* ray.rs:52-54 */

/* prim.rs:323-331 transforms_at */
static void inst_transforms_at(const Prim *p, float time, OraAffine *w2l, OraMat3 *nm) {
  if (p->has_end && time > 0.0f) {
    OraAffine l = lerp_affine(&p->l2w, &p->l2w_end, time);
    *w2l = affine_inverse(&l);
    OraMat3 m = {w2l->x, w2l->y, w2l->z};
    *nm = mat3_transpose(&m);
  } else {
    *w2l = p->w2l;
    *nm = p->normal_mat;
  }
}

static int prim_hit(const Prim *p, const OraRay *ray, float t_min, float t_max, OraPrimHit *out) {
  if ((ray->mask & p->mask) == 0) return 0; /* prim.rs:52-54 */
  switch (p->kind) {
    case PRIM_TRI: { /* prim.rs:99-105 */
      float t, u, v;
      if (!tri_intersect(ray, p->v0, p->v1, p->v2, t_min, t_max, &t, &u, &v)) return 0;
      return tri_hit_from_bary(p, t, u, v, out);
    }
    case PRIM_SPHERE: { /* prim.rs:133-161 */
      TRACE('s');
      v3 oc = v3_sub(ray->origin, p->center);
      float a = v3_len2(ray->dir);
      float half_b = v3_dot(oc, ray->dir);
      float c = v3_len2(oc) - p->radius * p->radius;
      float disc = half_b * half_b - a * c;
      if (disc < 0.0f) return 0;
      float sqrt_d = sqrtf(disc);
      float root = (-half_b - sqrt_d) / a;
      if (root <= t_min || root >= t_max) {
        root = (-half_b + sqrt_d) / a;
        if (root <= t_min || root >= t_max) return 0;
      }
      out->t = root;
      out->outward = v3_divs(v3_sub(ray_at(ray, root), p->center), p->radius);
      out->u = 0.0f; out->v = 0.0f; out->geom_id = p->geom_id; out->prim_id = 0;
      return 1;
    }
    default: { /* prim.rs:345-365 */
      OraAffine w2l; OraMat3 nm;
      inst_transforms_at(p, ray->time, &w2l, &nm);
      OraRay local = {affine_point(&w2l, ray->origin), affine_vector(&w2l, ray->dir), ray->time, ray->mask};
      if (g_stats) g_stats->instance_descents++;
      TRACE('I');
      g_depth++;
      int hit = bvh_hit(&p->scene->bvh, &local, t_min, t_max, out);
      g_depth--;
      TRACE('X');
      if (!hit) return 0;
      out->outward = v3_normalize(mat3_mul_v(&nm, out->outward));
      out->geom_id = p->geom_id;
      return 1;
    }
  }
}

static int prim_hit_any(const Prim *p, const OraRay *ray, float t_min, float t_max) {
  if ((ray->mask & p->mask) == 0) return 0;
  switch (p->kind) {
    case PRIM_TRI: { /* prim.rs:107-110 */
      float t, u, v;
      return tri_intersect(ray, p->v0, p->v1, p->v2, t_min, t_max, &t, &u, &v);
    }
    case PRIM_SPHERE: { /* prim.rs:31-33 default: hit().is_some() */
      OraPrimHit h;
      return prim_hit(p, ray, t_min, t_max, &h);
    }
    default: { /* prim.rs:367-378 */
      OraAffine w2l; OraMat3 nm;
      inst_transforms_at(p, ray->time, &w2l, &nm);
      OraRay local = {affine_point(&w2l, ray->origin), affine_vector(&w2l, ray->dir), ray->time, ray->mask};
      if (g_stats) g_stats->instance_descents++;
      g_depth++;
      int occ = bvh_hit_any(&p->scene->bvh, &local, t_min, t_max);
      g_depth--;
      return occ;
    }
  }
}

/* ------------------------------------------------------------------ */
/* bvh.rs — build                                                      */
/* ------------------------------------------------------------------ */
#define MAX_DEPTH 60          /* bvh.rs:142 */
#define MIN_LEAF 2            /* bvh.rs:144 */
#define MIN_LEAF_PACKED 4     /* bvh.rs:153 */
#define MAX_LEAF 8            /* bvh.rs:156 */
#define BINS 12               /* bvh.rs:158 */
#define SBVH_ALPHA 1e-5f      /* bvh.rs:164 */
#define SBVH_MAX_DEPTH 32     /* bvh.rs:167 */
#define EMPTY_LANE 0xffffffffu

typedef struct { OraAabb bbox; uint32_t idx; } PrimRef;             /* bvh.rs:281-284 */
typedef struct { OraAabb bbox; uint32_t first_or_right, count; } BNode; /* bvh.rs:171-178 */

typedef struct {
  const Prim *prims;
  BNode *nodes; size_t n_nodes, cap_nodes;
  uint32_t *indices; size_t n_indices, cap_indices;
} BuildCtx;

static v3 ref_centroid(const PrimRef *r) { return v3_scale(v3_add(r->bbox.mn, r->bbox.mx), 0.5f); } /* bvh.rs:287-289 */
static float surface_area(const OraAabb *b) { /* bvh.rs:811-814 */
  v3 d = v3_sub(b->mx, b->mn);
  return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
static OraAabb union_all(const PrimRef *refs, size_t n) { /* bvh.rs:816-820 */
  OraAabb acc = refs[0].bbox;
  for (size_t i = 1; i < n; i++) acc = aabb_union(acc, refs[i].bbox);
  return acc;
}
static int intersect_aabb(const OraAabb *a, const OraAabb *b, OraAabb *out) { /* bvh.rs:823-831 */
  v3 lo = v3_max(a->mn, b->mn), hi = v3_min(a->mx, b->mx);
  if (lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z) { out->mn = lo; out->mx = hi; return 1; }
  return 0;
}

typedef struct { int valid; int axis; float cmin, scale; int split_bin; float cost; OraAabb lb, rb; } ObjSplit;
typedef struct { int valid; int axis; float pos, cost; } SpatSplit;

/* Rust `x as usize` for f32: truncation toward zero, saturating, NaN -> 0. */
static size_t f32_as_usize(float x) {
  if (!(x > 0.0f)) return 0;
  if (x >= 1.8446744e19f) return (size_t)-1;
  return (size_t)x;
}
static int obj_bin_of(const PrimRef *r, int axis, float cmin, float scale) { /* bvh.rs:916, :1225 */
  size_t b = f32_as_usize((v3_get(ref_centroid(r), axis) - cmin) * scale);
  return (int)(b < BINS - 1 ? b : BINS - 1);
}

static ObjSplit best_object_split(const PrimRef *refs, size_t n) { /* bvh.rs:894-969 */
  ObjSplit best; memset(&best, 0, sizeof best);
  v3 cmin = ref_centroid(&refs[0]), cmax = cmin;
  for (size_t i = 1; i < n; i++) { v3 c = ref_centroid(&refs[i]); cmin = v3_min(cmin, c); cmax = v3_max(cmax, c); }
  v3 ext = v3_sub(cmax, cmin);
  int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
  if (v3_get(ext, axis) <= 1e-6f) return best;
  float scale = (float)BINS / v3_get(ext, axis);
  float cm = v3_get(cmin, axis);
  size_t counts[BINS]; OraAabb bounds[BINS]; int has[BINS];
  memset(counts, 0, sizeof counts); memset(has, 0, sizeof has);
  for (size_t i = 0; i < n; i++) {
    int b = obj_bin_of(&refs[i], axis, cm, scale);
    counts[b]++;
    bounds[b] = has[b] ? aabb_union(bounds[b], refs[i].bbox) : refs[i].bbox;
    has[b] = 1;
  }
  for (int split = 0; split < BINS - 1; split++) {
    OraAabb lb, rb; int hl = 0, hr = 0; size_t lc = 0, rc = 0;
    for (int b = 0; b <= split; b++) {
      lc += counts[b];
      if (has[b]) { lb = hl ? aabb_union(lb, bounds[b]) : bounds[b]; hl = 1; }
    }
    for (int b = split + 1; b < BINS; b++) {
      rc += counts[b];
      if (has[b]) { rb = hr ? aabb_union(rb, bounds[b]) : bounds[b]; hr = 1; }
    }
    if (lc == 0 || rc == 0) continue;
    float cost = surface_area(&lb) * (float)lc + surface_area(&rb) * (float)rc;
    if (!best.valid || cost < best.cost) {
      best.valid = 1; best.axis = axis; best.cmin = cm; best.scale = scale; best.split_bin = split;
      best.cost = cost; best.lb = lb; best.rb = rb;
    }
  }
  return best;
}

static SpatSplit best_spatial_split(const Prim *prims, const PrimRef *refs, size_t n, const OraAabb *bbox) {
  /* bvh.rs:974-1054 */
  SpatSplit best; memset(&best, 0, sizeof best);
  v3 ext = v3_sub(bbox->mx, bbox->mn);
  int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
  if (v3_get(ext, axis) <= 1e-6f) return best;
  float lo = v3_get(bbox->mn, axis);
  float width = v3_get(ext, axis) / (float)BINS;
#define SBIN(x) ({ size_t _b = f32_as_usize(((x) - lo) / width); (int)(_b > BINS - 1 ? BINS - 1 : _b); })
  size_t entry[BINS], exitc[BINS]; OraAabb bounds[BINS]; int has[BINS];
  memset(entry, 0, sizeof entry); memset(exitc, 0, sizeof exitc); memset(has, 0, sizeof has);
  for (size_t i = 0; i < n; i++) {
    const PrimRef *r = &refs[i];
    int b0 = SBIN(v3_get(r->bbox.mn, axis));
    int b1 = SBIN(v3_get(r->bbox.mx, axis));
    entry[b0]++; exitc[b1]++;
    if (b0 == b1) {
      bounds[b0] = has[b0] ? aabb_union(bounds[b0], r->bbox) : r->bbox; has[b0] = 1;
      continue;
    }
    for (int b = b0; b <= b1; b++) {
      float bin_lo = lo + (float)b * width, bin_hi = lo + (float)(b + 1) * width;
      OraAabb c, ci;
      if (prim_clipped_aabb(&prims[r->idx], axis, bin_lo, bin_hi, &c) && intersect_aabb(&c, &r->bbox, &ci)) {
        bounds[b] = has[b] ? aabb_union(bounds[b], ci) : ci; has[b] = 1;
      }
    }
  }
#undef SBIN
  for (int split = 0; split < BINS - 1; split++) {
    OraAabb lb, rb; int hl = 0, hr = 0; size_t lc = 0, rc = 0;
    for (int b = 0; b <= split; b++) {
      lc += entry[b];
      if (has[b]) { lb = hl ? aabb_union(lb, bounds[b]) : bounds[b]; hl = 1; }
    }
    for (int b = split + 1; b < BINS; b++) {
      rc += exitc[b];
      if (has[b]) { rb = hr ? aabb_union(rb, bounds[b]) : bounds[b]; hr = 1; }
    }
    if (lc == 0 || rc == 0) continue;
    /* The reference `expect`s bounds here; a side with refs but no surviving clipped bounds would panic
     * (abort in release). Not reachable on valid geometry. */
    if (!hl || !hr) abort();
    float cost = surface_area(&lb) * (float)lc + surface_area(&rb) * (float)rc;
    if (!best.valid || cost < best.cost) {
      best.valid = 1; best.axis = axis; best.pos = lo + (float)(split + 1) * width; best.cost = cost;
    }
  }
  return best;
}

static size_t min_leaf_for(const Prim *prims, const PrimRef *refs, size_t n) { /* bvh.rs:1201-1210 */
  for (size_t i = 0; i < n; i++)
    if (prims[refs[i].idx].kind != PRIM_TRI) return MIN_LEAF;
  return MIN_LEAF_PACKED;
}

static void emit_leaf(BuildCtx *c, OraAabb bbox, const PrimRef *refs, size_t n) { /* bvh.rs:833-842 */
  BNode nd = {bbox, (uint32_t)c->n_indices, (uint32_t)n};
  VEC_PUSH(c->nodes, c->n_nodes, c->cap_nodes, BNode, nd);
  for (size_t i = 0; i < n; i++) VEC_PUSH(c->indices, c->n_indices, c->cap_indices, uint32_t, refs[i].idx);
}

static void partition_by_bin(PrimRef *refs, size_t n, const ObjSplit *o, PrimRef **l, size_t *nl, PrimRef **r,
                             size_t *nr) { /* bvh.rs:1224-1237 */
  size_t cl = 0;
  for (size_t i = 0; i < n; i++)
    if (obj_bin_of(&refs[i], o->axis, o->cmin, o->scale) <= o->split_bin) cl++;
  *l = (PrimRef *)malloc((cl ? cl : 1) * sizeof(PrimRef));
  *r = (PrimRef *)malloc(((n - cl) ? (n - cl) : 1) * sizeof(PrimRef));
  size_t a = 0, b = 0;
  for (size_t i = 0; i < n; i++) {
    if (obj_bin_of(&refs[i], o->axis, o->cmin, o->scale) <= o->split_bin) (*l)[a++] = refs[i];
    else (*r)[b++] = refs[i];
  }
  *nl = a; *nr = b;
}

/* Emits the subtree for refs in depth-first order. The reference builds local Subtrees and splices them
 * with merge() (bvh.rs:846-872); the spliced result is exactly this preorder: node, left subtree, right
 * subtree, with leaf offsets running left to right. Takes ownership of refs. */
static void build_subtree(BuildCtx *c, PrimRef *refs, size_t n, size_t depth, float root_area);

static void emit_internal(BuildCtx *c, OraAabb bbox, PrimRef *l, size_t nl, PrimRef *r, size_t nr, size_t depth,
                          float root_area) {
  size_t slot = c->n_nodes;
  BNode nd = {bbox, 0, 0};
  VEC_PUSH(c->nodes, c->n_nodes, c->cap_nodes, BNode, nd);
  build_subtree(c, l, nl, depth + 1, root_area);
  c->nodes[slot].first_or_right = (uint32_t)c->n_nodes;
  build_subtree(c, r, nr, depth + 1, root_area);
}

/* bvh.rs:1173-1196 */
static void object_partition_or_leaf(BuildCtx *c, PrimRef *refs, size_t n, OraAabb bbox, const ObjSplit *o,
                                     size_t depth, float root_area) {
  if (o->valid && n > min_leaf_for(c->prims, refs, n)) {
    PrimRef *l, *r; size_t nl, nr;
    partition_by_bin(refs, n, o, &l, &nl, &r, &nr);
    if (nl == 0 || nr == 0) {
      /* all = l ++ r */
      PrimRef *all = (PrimRef *)malloc((n ? n : 1) * sizeof(PrimRef));
      memcpy(all, l, nl * sizeof(PrimRef));
      memcpy(all + nl, r, nr * sizeof(PrimRef));
      emit_leaf(c, bbox, all, n);
      free(all); free(l); free(r); free(refs);
      return;
    }
    free(refs);
    emit_internal(c, bbox, l, nl, r, nr, depth, root_area);
    return;
  }
  emit_leaf(c, bbox, refs, n);
  free(refs);
}

static void build_subtree(BuildCtx *c, PrimRef *refs, size_t n, size_t depth, float root_area) { /* bvh.rs:1059-1169 */
  OraAabb bbox = union_all(refs, n);
  size_t count = n;
  size_t min_leaf = min_leaf_for(c->prims, refs, n);
  if (count <= min_leaf || depth >= MAX_DEPTH) { emit_leaf(c, bbox, refs, n); free(refs); return; }

  ObjSplit object = best_object_split(refs, n);
  SpatSplit spatial; memset(&spatial, 0, sizeof spatial);
  if (object.valid && depth < SBVH_MAX_DEPTH) {
    OraAabb ov;
    float overlap = intersect_aabb(&object.lb, &object.rb, &ov) ? surface_area(&ov) : 0.0f;
    if (overlap / root_area > SBVH_ALPHA) {
      SpatSplit s = best_spatial_split(c->prims, refs, n, &bbox);
      if (s.valid && s.cost < object.cost) spatial = s;
    }
  }

  PrimRef *l, *r; size_t nl = 0, nr = 0;
  if (spatial.valid) {
    const SpatSplit *s = &spatial;
    l = (PrimRef *)malloc((n + 1) * sizeof(PrimRef));
    r = (PrimRef *)malloc((n + 1) * sizeof(PrimRef));
    for (size_t i = 0; i < n; i++) {
      PrimRef rr = refs[i];
      if (v3_get(rr.bbox.mx, s->axis) <= s->pos) l[nl++] = rr;
      else if (v3_get(rr.bbox.mn, s->axis) >= s->pos) r[nr++] = rr;
      else {
        const Prim *prim = &c->prims[rr.idx];
        OraAabb cb, ci;
        if (prim_clipped_aabb(prim, s->axis, -ORA_INF, s->pos, &cb) && intersect_aabb(&cb, &rr.bbox, &ci)) {
          PrimRef nr_ = {ci, rr.idx}; l[nl++] = nr_;
        }
        if (prim_clipped_aabb(prim, s->axis, s->pos, ORA_INF, &cb) && intersect_aabb(&cb, &rr.bbox, &ci)) {
          PrimRef nr_ = {ci, rr.idx}; r[nr++] = nr_;
        }
      }
    }
    free(refs);
    if (nl == 0 || nr == 0) {
      PrimRef *all = (PrimRef *)malloc((nl + nr + 1) * sizeof(PrimRef));
      memcpy(all, l, nl * sizeof(PrimRef));
      memcpy(all + nl, r, nr * sizeof(PrimRef));
      free(l); free(r);
      object_partition_or_leaf(c, all, nl + nr, bbox, &object, depth, root_area);
      return;
    }
  } else if (object.valid) {
    if (count <= MAX_LEAF && object.cost >= surface_area(&bbox) * (float)count) {
      emit_leaf(c, bbox, refs, n); free(refs); return;
    }
    partition_by_bin(refs, n, &object, &l, &nl, &r, &nr);
    free(refs);
  } else {
    size_t mid = count / 2;
    nl = mid; nr = n - mid;
    l = (PrimRef *)malloc((nl ? nl : 1) * sizeof(PrimRef));
    r = (PrimRef *)malloc((nr ? nr : 1) * sizeof(PrimRef));
    memcpy(l, refs, nl * sizeof(PrimRef));
    memcpy(r, refs + mid, nr * sizeof(PrimRef));
    free(refs);
  }
  emit_internal(c, bbox, l, nl, r, nr, depth, root_area);
}

/* ---- collapse (bvh.rs:1299-1375) ---- */
typedef struct {
  OraWideNode *wide; size_t n_wide, cap_wide;
  OraLeaf *leaves; size_t n_leaves, cap_leaves;
  OraTri4 *packets; size_t n_packets, cap_packets;
  uint32_t *indices; size_t n_indices, cap_indices;
} Collapse;

static OraWideNode wide_empty(void) { /* bvh.rs:230-241 */
  OraWideNode w; memset(&w, 0, sizeof w);
  for (int a = 0; a < 3; a++) for (int l = 0; l < 4; l++) { w.bmin[a][l] = ORA_INF; w.bmax[a][l] = ORA_INF; }
  for (int l = 0; l < 4; l++) w.child[l] = EMPTY_LANE;
  return w;
}
static void wide_set_lane(OraWideNode *w, int lane, const OraAabb *b) { /* bvh.rs:243-251 */
  w->bmin[0][lane] = b->mn.x; w->bmin[1][lane] = b->mn.y; w->bmin[2][lane] = b->mn.z;
  w->bmax[0][lane] = b->mx.x; w->bmax[1][lane] = b->mx.y; w->bmax[2][lane] = b->mx.z;
  w->flags |= 1u << lane;
}

static uint32_t push_leaf(Collapse *c, const uint32_t *range, size_t n, const Prim *prims) { /* bvh.rs:1256-1290 */
  uint32_t pkt_first = (uint32_t)c->n_packets, idx_first = (uint32_t)c->n_indices, idx_count = 0;
  TriRef batch[4]; int nb = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t pi = range[i];
    const Prim *p = &prims[pi];
    if (p->kind == PRIM_TRI) {
      TriRef t = {p->v0, p->v1, p->v2, pi, p->mask};
      batch[nb++] = t;
      if (nb == 4) {
        OraTri4 pk; tri4_new(&pk, batch, 4);
        VEC_PUSH(c->packets, c->n_packets, c->cap_packets, OraTri4, pk);
        nb = 0;
      }
    } else {
      VEC_PUSH(c->indices, c->n_indices, c->cap_indices, uint32_t, pi);
      idx_count++;
    }
  }
  if (nb > 0) {
    OraTri4 pk; tri4_new(&pk, batch, nb);
    VEC_PUSH(c->packets, c->n_packets, c->cap_packets, OraTri4, pk);
  }
  OraLeaf lf = {pkt_first, (uint32_t)c->n_packets - pkt_first, idx_first, idx_count};
  VEC_PUSH(c->leaves, c->n_leaves, c->cap_leaves, OraLeaf, lf);
  return (uint32_t)c->n_leaves - 1;
}

static uint32_t collapse_node(Collapse *c, const BNode *bin, const uint32_t *indices, const Prim *prims,
                              uint32_t b_idx) { /* bvh.rs:1328-1375 */
  size_t slot = c->n_wide;
  OraWideNode e = wide_empty();
  VEC_PUSH(c->wide, c->n_wide, c->cap_wide, OraWideNode, e);
  uint32_t kids[4] = {0, 0, 0, 0};
  kids[0] = b_idx + 1;
  kids[1] = bin[b_idx].first_or_right;
  int n_kids = 2;
  while (n_kids < 4) {
    int best = -1; float best_a = 0.0f;
    for (int i = 0; i < n_kids; i++) {
      uint32_t k = kids[i];
      if (bin[k].count == 0) {
        float a = surface_area(&bin[k].bbox);
        if (best < 0 || a > best_a) { best = i; best_a = a; }
      }
    }
    if (best < 0) break;
    uint32_t k = kids[best];
    kids[best] = k + 1;
    kids[n_kids] = bin[k].first_or_right;
    n_kids++;
  }
  for (int lane = 0; lane < n_kids; lane++) {
    uint32_t k = kids[lane];
    OraAabb bounds = bin[k].bbox;
    wide_set_lane(&c->wide[slot], lane, &bounds);
    if (bin[k].count > 0) {
      uint32_t li = push_leaf(c, indices + bin[k].first_or_right, bin[k].count, prims);
      c->wide[slot].child[lane] = li;
      c->wide[slot].flags |= 1u << (4 + lane);
    } else {
      uint32_t ci = collapse_node(c, bin, indices, prims, k);
      c->wide[slot].child[lane] = ci;
    }
  }
  return (uint32_t)slot;
}

static void bvh_new(Bvh *out, Prim *prims, size_t n_prims) { /* bvh.rs:300-327 */
  memset(out, 0, sizeof *out);
  out->prims = prims; out->n_prims = n_prims;
  if (n_prims == 0) return;
  PrimRef *refs = (PrimRef *)malloc(n_prims * sizeof(PrimRef));
  for (size_t i = 0; i < n_prims; i++) { refs[i].bbox = prim_bbox(&prims[i]); refs[i].idx = (uint32_t)i; }
  OraAabb root = union_all(refs, n_prims);
  BuildCtx bc; memset(&bc, 0, sizeof bc); bc.prims = prims;
  build_subtree(&bc, refs, n_prims, 0, surface_area(&root));
  Collapse c; memset(&c, 0, sizeof c);
  if (bc.nodes[0].count > 0) { /* single-leaf tree, bvh.rs:1309-1317 */
    OraWideNode w = wide_empty();
    wide_set_lane(&w, 0, &bc.nodes[0].bbox);
    w.child[0] = push_leaf(&c, bc.indices + bc.nodes[0].first_or_right, bc.nodes[0].count, prims);
    w.flags |= 1u << 4;
    VEC_PUSH(c.wide, c.n_wide, c.cap_wide, OraWideNode, w);
  } else {
    collapse_node(&c, bc.nodes, bc.indices, prims, 0);
  }
  free(bc.nodes); free(bc.indices);
  out->wide = c.wide; out->n_wide = c.n_wide;
  out->leaves = c.leaves; out->n_leaves = c.n_leaves;
  out->packets = c.packets; out->n_packets = c.n_packets;
  out->indices = c.indices; out->n_indices = c.n_indices;
  out->has_bbox = 1; out->root_bbox = root;
}

/* ------------------------------------------------------------------ */
/* bvh.rs — traversal                                                  */
/* ------------------------------------------------------------------ */
static v3 safe_inv3(v3 d) { /* bvh.rs:662-668 */
  const float TINY = 1e-20f, HUGE_ = 1e20f;
  v3 r;
  r.x = ora_abs(d.x) < TINY ? ora_copysign(HUGE_, d.x) : 1.0f / d.x;
  r.y = ora_abs(d.y) < TINY ? ora_copysign(HUGE_, d.y) : 1.0f / d.y;
  r.z = ora_abs(d.z) < TINY ? ora_copysign(HUGE_, d.z) : 1.0f / d.z;
  return r;
}

/* bvh.rs:790-808 slab4, one lane. */
static void slab_lane(const OraWideNode *nd, int l, v3 o, v3 inv, float t_min, float t_max, float *tnear, float *tfar) {
  float t0x = (nd->bmin[0][l] - o.x) * inv.x, t1x = (nd->bmax[0][l] - o.x) * inv.x;
  float t0y = (nd->bmin[1][l] - o.y) * inv.y, t1y = (nd->bmax[1][l] - o.y) * inv.y;
  float t0z = (nd->bmin[2][l] - o.z) * inv.z, t1z = (nd->bmax[2][l] - o.z) * inv.z;
  float tn = ora_sse_max(ora_sse_max(ora_sse_max(ora_sse_min(t0x, t1x), ora_sse_min(t0y, t1y)), ora_sse_min(t0z, t1z)), t_min);
  float tf = ora_sse_min(ora_sse_min(ora_sse_min(ora_sse_max(t0x, t1x), ora_sse_max(t0y, t1y)), ora_sse_max(t0z, t1z)), t_max);
  *tnear = tn; *tfar = tf;
}

/* bvh.rs:514-572 */
static int intersect_leaf(const Bvh *b, uint32_t leaf_idx, const OraRay *ray, const RayShear *sh, float t_min,
                          float t_max, OraPrimHit *out) {
  const OraLeaf *leaf = &b->leaves[leaf_idx];
  TSTAT(leaves, 1); TSTAT(packets, leaf->pkt_count); TSTAT(prims, leaf->idx_count);
  float closest = t_max;
  int found = 0;
  for (uint32_t k = 0; k < leaf->pkt_count; k++) {
    const OraTri4 *pk = &b->packets[leaf->pkt_first + k];
    TRACE('P');
    Hit4 h = tri4_intersect(pk, sh, ray->origin, ray->mask, t_min, closest);
    for (int lane = 0; lane < 4; lane++) {
      if (!(h.hits & (1u << lane))) continue;
      if (h.t[lane] > closest) continue; /* bvh.rs:542 */
      OraPrimHit ph;
      if (tri_hit_from_bary(&b->prims[pk->prim[lane]], h.t[lane], h.u[lane], h.v[lane], &ph)) {
        closest = ph.t; *out = ph; found = 1;
        if (g_stats) g_stats->accepted_hits++;
      }
    }
    for (int lane = 0; lane < 4; lane++) {
      if (!(h.fallback & (1u << lane))) continue;
      OraPrimHit ph;
      if (prim_hit(&b->prims[pk->prim[lane]], ray, t_min, closest, &ph)) {
        closest = ph.t; *out = ph; found = 1;
        if (g_stats) g_stats->accepted_hits++;
      }
    }
  }
  for (uint32_t k = 0; k < leaf->idx_count; k++) {
    uint32_t pi = b->indices[leaf->idx_first + k];
    OraPrimHit ph;
    if (prim_hit(&b->prims[pi], ray, t_min, closest, &ph)) {
      closest = ph.t; *out = ph; found = 1;
      if (g_stats) g_stats->accepted_hits++;
    }
  }
  return found;
}

/* bvh.rs:617-653 */
static int occlude_leaf(const Bvh *b, uint32_t leaf_idx, const OraRay *ray, const RayShear *sh, float t_min,
                        float t_max) {
  const OraLeaf *leaf = &b->leaves[leaf_idx];
  TSTAT(leaves, 1); TSTAT(packets, leaf->pkt_count); TSTAT(prims, leaf->idx_count);
  for (uint32_t k = 0; k < leaf->pkt_count; k++) {
    const OraTri4 *pk = &b->packets[leaf->pkt_first + k];
    Hit4 h = tri4_intersect(pk, sh, ray->origin, ray->mask, t_min, t_max);
    if (h.hits) return 1;
    for (int lane = 0; lane < 4; lane++)
      if ((h.fallback & (1u << lane)) && prim_hit_any(&b->prims[pk->prim[lane]], ray, t_min, t_max)) return 1;
  }
  for (uint32_t k = 0; k < leaf->idx_count; k++)
    if (prim_hit_any(&b->prims[b->indices[leaf->idx_first + k]], ray, t_min, t_max)) return 1;
  return 0;
}

/* Unbounded stack (the reference's inline-32 + Vec spill behaves as one LIFO, bvh.rs:711-755). */
#define ORA_STACK 256

static int bvh_hit(const Bvh *b, const OraRay *ray, float t_min, float t_max, OraPrimHit *out) { /* bvh.rs:441-509 */
  if (b->n_wide == 0) return 0;
  TSTAT(queries, 1);
  float closest = t_max;
  int found = 0;
  v3 inv = safe_inv3(ray->dir);
  RayShear sh; memset(&sh, 0, sizeof sh);
  if (b->n_packets) sh = shear_new(ray);
  uint32_t stack[ORA_STACK]; int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    uint32_t node_idx = stack[--sp];
    TSTAT(nodes, 1);
    TRACE('N');
    const OraWideNode *nd = &b->wide[node_idx];
    float tn[4]; uint32_t mask = 0;
    for (int l = 0; l < 4; l++) {
      float tf;
      slab_lane(nd, l, ray->origin, inv, t_min, closest, &tn[l], &tf);
      if (tn[l] <= tf) mask |= 1u << l;
    }
    mask &= nd->flags & 0xfu;
    if (!mask) continue;
    float ot[4]; int ol[4]; int n_hit = 0;
    for (int l = 0; l < 4; l++) {
      if (!(mask & (1u << l))) continue;
      float t = tn[l];
      int i = n_hit;
      while (i > 0 && ot[i - 1] > t) { ot[i] = ot[i - 1]; ol[i] = ol[i - 1]; i--; }
      ot[i] = t; ol[i] = l;
      n_hit++;
    }
    for (int i = 0; i < n_hit; i++) {
      int l = ol[i];
      if (nd->flags & (1u << (4 + l))) {
        OraPrimHit ph;
        if (intersect_leaf(b, nd->child[l], ray, &sh, t_min, closest, &ph)) { closest = ph.t; *out = ph; found = 1; }
      }
    }
    for (int i = n_hit - 1; i >= 0; i--) {
      int l = ol[i];
      if (!(nd->flags & (1u << (4 + l)))) {
        if (sp >= ORA_STACK) abort();
        stack[sp++] = nd->child[l];
        if (g_stats && (uint64_t)sp > g_stats->stack_high_water) g_stats->stack_high_water = (uint64_t)sp;
      }
    }
  }
  return found;
}

static int bvh_hit_any(const Bvh *b, const OraRay *ray, float t_min, float t_max) { /* bvh.rs:585-611 */
  if (b->n_wide == 0) return 0;
  TSTAT(queries, 1);
  v3 inv = safe_inv3(ray->dir);
  RayShear sh; memset(&sh, 0, sizeof sh);
  if (b->n_packets) sh = shear_new(ray);
  uint32_t stack[ORA_STACK]; int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    uint32_t node_idx = stack[--sp];
    TSTAT(nodes, 1);
    const OraWideNode *nd = &b->wide[node_idx];
    uint32_t mask = 0;
    for (int l = 0; l < 4; l++) {
      float tn, tf;
      slab_lane(nd, l, ray->origin, inv, t_min, t_max, &tn, &tf);
      if (tn <= tf) mask |= 1u << l;
    }
    mask &= nd->flags & 0xfu;
    for (int l = 0; l < 4; l++) {
      if (!(mask & (1u << l))) continue;
      if (nd->flags & (1u << (4 + l))) {
        if (occlude_leaf(b, nd->child[l], ray, &sh, t_min, t_max)) return 1;
      } else {
        if (sp >= ORA_STACK) abort();
        stack[sp++] = nd->child[l];
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* scene.rs                                                            */
/* ------------------------------------------------------------------ */
enum { G_MESH = 0, G_SPHERE = 1, G_INSTANCE = 2 };
typedef struct {
  int kind; uint32_t mask;
  float *verts; size_t nverts; uint32_t *idx; size_t ntris; float *normals; size_t nnormals;
  v3 center; float radius;
  OraScene *scene; OraAffine l2w; int has_end; OraAffine l2w_end;
} Geom;
struct OraBuilder { Geom *g; size_t n, cap; };

OraBuilder *ora_builder_new(void) { return (OraBuilder *)calloc(1, sizeof(OraBuilder)); }
size_t ora_builder_count(const OraBuilder *b) { return b->n; }

static void geom_clear(Geom *g) { free(g->verts); free(g->idx); free(g->normals); uint32_t m = g->mask; memset(g, 0, sizeof *g); g->mask = m; }
static void geom_set_mesh(Geom *g, const float *verts, size_t nverts, const uint32_t *idx, size_t ntris,
                          const float *normals, size_t nnormals) {
  g->kind = G_MESH;
  g->nverts = nverts; g->ntris = ntris; g->nnormals = normals ? nnormals : 0;
  g->verts = (float *)malloc((nverts ? nverts : 1) * 3 * sizeof(float));
  if (nverts) memcpy(g->verts, verts, nverts * 3 * sizeof(float));
  g->idx = (uint32_t *)malloc((ntris ? ntris : 1) * 3 * sizeof(uint32_t));
  if (ntris) memcpy(g->idx, idx, ntris * 3 * sizeof(uint32_t));
  g->normals = NULL;
  if (normals) {
    g->normals = (float *)malloc((nnormals ? nnormals : 1) * 3 * sizeof(float));
    if (nnormals) memcpy(g->normals, normals, nnormals * 3 * sizeof(float));
  }
}
static uint32_t push_geom(OraBuilder *b, uint32_t mask) {
  Geom g; memset(&g, 0, sizeof g); g.mask = mask;
  VEC_PUSH(b->g, b->n, b->cap, Geom, g);
  return (uint32_t)b->n - 1;
}
uint32_t ora_attach_triangles(OraBuilder *b, const float *verts, size_t nverts, const uint32_t *idx, size_t ntris,
                              const float *normals, size_t nnormals, uint32_t mask) {
  uint32_t id = push_geom(b, mask);
  geom_set_mesh(&b->g[id], verts, nverts, idx, ntris, normals, nnormals);
  return id;
}
uint32_t ora_attach_empty(OraBuilder *b, uint32_t mask) { return ora_attach_triangles(b, NULL, 0, NULL, 0, NULL, 0, mask); }
uint32_t ora_attach_sphere(OraBuilder *b, float cx, float cy, float cz, float r, uint32_t mask) {
  uint32_t id = push_geom(b, mask);
  b->g[id].kind = G_SPHERE; b->g[id].center = v3_new(cx, cy, cz); b->g[id].radius = r;
  return id;
}
uint32_t ora_attach_instance(OraBuilder *b, OraScene *scene, const float l2w[12], const float *l2w_end, uint32_t mask) {
  uint32_t id = push_geom(b, mask);
  Geom *g = &b->g[id];
  g->kind = G_INSTANCE; g->scene = scene; g->l2w = affine_from12(l2w);
  g->has_end = l2w_end != NULL;
  if (l2w_end) g->l2w_end = affine_from12(l2w_end);
  return id;
}
int ora_set_triangles(OraBuilder *b, uint32_t id, const float *verts, size_t nverts, const uint32_t *idx, size_t ntris,
                      const float *normals, size_t nnormals) {
  if (id >= b->n) return -1;
  geom_clear(&b->g[id]);
  geom_set_mesh(&b->g[id], verts, nverts, idx, ntris, normals, nnormals);
  return 0;
}
int ora_set_sphere(OraBuilder *b, uint32_t id, float cx, float cy, float cz, float r) {
  if (id >= b->n) return -1;
  geom_clear(&b->g[id]);
  b->g[id].kind = G_SPHERE; b->g[id].center = v3_new(cx, cy, cz); b->g[id].radius = r;
  return 0;
}
int ora_set_instance(OraBuilder *b, uint32_t id, OraScene *scene, const float l2w[12], const float *l2w_end) {
  if (id >= b->n) return -1;
  geom_clear(&b->g[id]);
  Geom *g = &b->g[id];
  g->kind = G_INSTANCE; g->scene = scene; g->l2w = affine_from12(l2w);
  g->has_end = l2w_end != NULL;
  if (l2w_end) g->l2w_end = affine_from12(l2w_end);
  return 0;
}

OraScene *ora_commit(OraBuilder *b) { /* scene.rs:226-341 */
  size_t total = 0;
  for (size_t i = 0; i < b->n; i++) total += b->g[i].kind == G_MESH ? b->g[i].ntris : 1;
  Prim *prims = (Prim *)calloc(total ? total : 1, sizeof(Prim));
  size_t np = 0;
  int has_motion = 0;
  for (size_t gi = 0; gi < b->n; gi++) {
    Geom *g = &b->g[gi];
    uint32_t geom_id = (uint32_t)gi;
    if (g->kind == G_MESH) {
      for (size_t ti = 0; ti < g->ntris; ti++) {
        size_t i0 = g->idx[3 * ti], i1 = g->idx[3 * ti + 1], i2 = g->idx[3 * ti + 2];
        if (i0 >= g->nverts || i1 >= g->nverts || i2 >= g->nverts) continue;
        Prim *p = &prims[np++];
        p->kind = PRIM_TRI; p->geom_id = geom_id; p->prim_id = (uint32_t)ti; p->mask = g->mask;
        p->v0 = v3_new(g->verts[3 * i0], g->verts[3 * i0 + 1], g->verts[3 * i0 + 2]);
        p->v1 = v3_new(g->verts[3 * i1], g->verts[3 * i1 + 1], g->verts[3 * i1 + 2]);
        p->v2 = v3_new(g->verts[3 * i2], g->verts[3 * i2 + 1], g->verts[3 * i2 + 2]);
        if (g->normals && i0 < g->nnormals && i1 < g->nnormals && i2 < g->nnormals) {
          p->has_normals = 1;
          p->n0 = v3_new(g->normals[3 * i0], g->normals[3 * i0 + 1], g->normals[3 * i0 + 2]);
          p->n1 = v3_new(g->normals[3 * i1], g->normals[3 * i1 + 1], g->normals[3 * i1 + 2]);
          p->n2 = v3_new(g->normals[3 * i2], g->normals[3 * i2 + 1], g->normals[3 * i2 + 2]);
        }
      }
    } else if (g->kind == G_SPHERE) {
      Prim *p = &prims[np++];
      p->kind = PRIM_SPHERE; p->geom_id = geom_id; p->mask = g->mask; p->center = g->center; p->radius = g->radius;
    } else {
      if (!g->scene->bvh.has_bbox) continue; /* empty instanced scene, scene.rs:307-309 */
      OraAabb inner = g->scene->bvh.root_bbox;
      Prim *p = &prims[np++];
      p->kind = PRIM_INSTANCE; p->geom_id = geom_id; p->mask = g->mask; p->scene = g->scene;
      p->l2w = g->l2w;
      p->w2l = affine_inverse(&g->l2w);
      OraMat3 m = {p->w2l.x, p->w2l.y, p->w2l.z};
      p->normal_mat = mat3_transpose(&m);
      p->has_end = g->has_end; p->l2w_end = g->l2w_end;
      p->bounds = g->has_end ? aabb_union(transformed_aabb(&inner, &g->l2w), transformed_aabb(&inner, &g->l2w_end))
                             : transformed_aabb(&inner, &g->l2w);
      has_motion |= g->has_end || g->scene->has_motion;
    }
  }
  OraScene *s = (OraScene *)calloc(1, sizeof(OraScene));
  s->n_geoms = (uint32_t)b->n;
  s->has_motion = has_motion;
  bvh_new(&s->bvh, prims, np);
  for (size_t i = 0; i < b->n; i++) { free(b->g[i].verts); free(b->g[i].idx); free(b->g[i].normals); }
  free(b->g); free(b);
  return s;
}

void ora_scene_free(OraScene *s) {
  if (!s) return;
  free(s->bvh.wide); free(s->bvh.leaves); free(s->bvh.packets); free(s->bvh.indices); free(s->bvh.prims);
  free(s);
}

static void orient(const OraRay *ray, const OraPrimHit *h, OraRayHit *out) { /* scene.rs:355-365 */
  int front = v3_dot(ray->dir, h->outward) < 0.0f;
  out->t = h->t;
  out->normal = front ? h->outward : v3_neg(h->outward);
  out->front_face = front;
  out->u = h->u; out->v = h->v; out->geom_id = h->geom_id; out->prim_id = h->prim_id;
}
int ora_intersect(const OraScene *s, const OraRay *ray, float t_min, float t_max, OraRayHit *out) {
  OraPrimHit h;
  g_depth = 0;
  if (!bvh_hit(&s->bvh, ray, t_min, t_max, &h)) return 0;
  orient(ray, &h, out);
  return 1;
}
int ora_occluded(const OraScene *s, const OraRay *ray, float t_min, float t_max) {
  g_depth = 0;
  if (!g_stats_any) return bvh_hit_any(&s->bvh, ray, t_min, t_max);
  OraTravStats *keep = g_stats;  /* any-hit queries count in their own sink while one is set */
  g_stats = g_stats_any;
  const int occ = bvh_hit_any(&s->bvh, ray, t_min, t_max);
  g_stats = keep;
  return occ;
}
int ora_scene_bounds(const OraScene *s, float out[6]) {
  if (!s->bvh.has_bbox) return 0;
  out[0] = s->bvh.root_bbox.mn.x; out[1] = s->bvh.root_bbox.mn.y; out[2] = s->bvh.root_bbox.mn.z;
  out[3] = s->bvh.root_bbox.mx.x; out[4] = s->bvh.root_bbox.mx.y; out[5] = s->bvh.root_bbox.mx.z;
  return 1;
}
uint32_t ora_geometry_count(const OraScene *s) { return s->n_geoms; }
int ora_has_motion(const OraScene *s) { return s->has_motion; }
size_t ora_primitive_count(const OraScene *s) { return s->bvh.n_prims; }
void ora_primitive_breakdown(const OraScene *s, size_t out[5]) { /* scene.rs:409 */
  memset(out, 0, 5 * sizeof out[0]);
  for (size_t i = 0; i < s->bvh.n_prims; i++) {
    const int k = s->bvh.prims[i].kind;
    out[k == PRIM_TRI ? 0 : k == PRIM_SPHERE ? 1 : 4]++;
  }
}
/* scene.rs:446-455 + bvh.rs:335-345: (count, scene diagonal, mean and max diagonal of the top-level primitives' boxes) */
size_t ora_primitive_extents(const OraScene *s, float out[3]) {
  float sum = 0.0f, mx = 0.0f;
  for (size_t i = 0; i < s->bvh.n_prims; i++) {
    const OraAabb b = prim_bbox(&s->bvh.prims[i]);
    const float d = v3_len(v3_sub(b.mx, b.mn));
    sum += d;
    mx = d > mx ? d : mx;
  }
  out[0] = s->bvh.has_bbox ? v3_len(v3_sub(s->bvh.root_bbox.mx, s->bvh.root_bbox.mn)) : 0.0f;
  out[1] = s->bvh.n_prims == 0 ? 0.0f : sum / (float)s->bvh.n_prims;
  out[2] = mx;
  return s->bvh.n_prims;
}
/* bvh.rs:397-416: every primitive of this tree once; an instanced scene is entered the first time it is met. */
typedef struct { const OraScene **seen; size_t n, cap; } Visited;
static void accumulate_unique(const OraScene *s, Visited *vis, size_t acc[5]) {
  for (size_t i = 0; i < s->bvh.n_prims; i++) {
    const Prim *p = &s->bvh.prims[i];
    if (p->kind == PRIM_TRI) acc[0]++;
    else if (p->kind == PRIM_SPHERE) acc[1]++;
    else {
      acc[4]++;
      int known = 0;
      for (size_t k = 0; k < vis->n && !known; k++) known = vis->seen[k] == p->scene;
      if (known) continue;
      if (vis->n == vis->cap) {
        vis->cap = vis->cap ? 2 * vis->cap : 16;
        vis->seen = (const OraScene **)realloc((void *)vis->seen, vis->cap * sizeof *vis->seen);
      }
      vis->seen[vis->n++] = p->scene;
      accumulate_unique(p->scene, vis, acc);
    }
  }
}
void ora_unique_primitive_breakdown(const OraScene *s, size_t out[5]) { /* scene.rs:422-427 */
  Visited vis = {NULL, 0, 0};
  memset(out, 0, 5 * sizeof out[0]);
  accumulate_unique(s, &vis, out);
  free((void *)vis.seen);
}

int ora_linear_scan(const OraScene *s, const OraRay *ray, float t_min, float t_max, OraRayHit *out) {
  float closest = t_max; int found = 0; OraPrimHit best;
  for (size_t i = 0; i < s->bvh.n_prims; i++) {
    OraPrimHit h;
    if (prim_hit(&s->bvh.prims[i], ray, t_min, closest, &h)) { closest = h.t; best = h; found = 1; }
  }
  if (found) orient(ray, &best, out);
  return found;
}

static OraRay ray_from8(const float *r) {
  OraRay ray = {v3_new(r[0], r[1], r[2]), v3_new(r[3], r[4], r[5]), r[6], ora_f2u(r[7])};
  return ray;
}
void ora_intersect_n(const OraScene *s, const float *rays, size_t n, float t_min, float t_max, float *hit_f,
                     uint32_t *hit_ids, uint8_t *front) {
  for (size_t i = 0; i < n; i++) {
    OraRay ray = ray_from8(rays + 8 * i);
    OraRayHit h;
    if (ora_intersect(s, &ray, t_min, t_max, &h)) {
      float *f = hit_f + 6 * i;
      f[0] = h.t; f[1] = h.normal.x; f[2] = h.normal.y; f[3] = h.normal.z; f[4] = h.u; f[5] = h.v;
      hit_ids[2 * i] = h.geom_id; hit_ids[2 * i + 1] = h.prim_id;
      front[i] = (uint8_t)h.front_face;
    } else {
      memset(hit_f + 6 * i, 0, 6 * sizeof(float));
      hit_ids[2 * i] = ORA_INVALID_ID; hit_ids[2 * i + 1] = ORA_INVALID_ID;
      front[i] = 0;
    }
  }
}
void ora_occluded_n(const OraScene *s, const float *rays, size_t n, float t_min, float t_max, uint8_t *out) {
  for (size_t i = 0; i < n; i++) {
    OraRay ray = ray_from8(rays + 8 * i);
    out[i] = (uint8_t)ora_occluded(s, &ray, t_min, t_max);
  }
}

size_t ora_bvh_counts(const OraScene *s, size_t out[5]) {
  out[0] = s->bvh.n_wide; out[1] = s->bvh.n_leaves; out[2] = s->bvh.n_packets; out[3] = s->bvh.n_indices;
  out[4] = s->bvh.n_prims;
  return s->bvh.n_wide;
}
const OraWideNode *ora_bvh_nodes(const OraScene *s) { return s->bvh.wide; }
const OraLeaf *ora_bvh_leaves(const OraScene *s) { return s->bvh.leaves; }
const OraTri4 *ora_bvh_packets(const OraScene *s) { return s->bvh.packets; }
const uint32_t *ora_bvh_indices(const OraScene *s) { return s->bvh.indices; }
