/* ORACLE — TEST INFRASTRUCTURE ONLY (see ora_math.h header).
 *
 * CPU restatement of the crust-rt kernel layer:
 *   crates/crust-rt/src/{ray,aabb,triangle,prim,bvh,scene}.rs
 * Each function cites the reference lines it follows.
 */
#ifndef ORA_RT_H
#define ORA_RT_H

#include "ora_math.h"
#include <stddef.h>

#define ORA_MASK_CAMERA 1u   /* ray.rs:8 */
#define ORA_MASK_SHADOW 2u   /* ray.rs:9 */
#define ORA_MASK_INDIRECT 4u /* ray.rs:10 */
#define ORA_MASK_ALL 0xffffffffu
#define ORA_INVALID_ID 0xffffffffu /* lib.rs:53 */

typedef struct { v3 origin, dir; float time; uint32_t mask; } OraRay; /* ray.rs:18-23 */
typedef struct { v3 mn, mx; } OraAabb;                                 /* aabb.rs:5-8 */
/* glam Affine3A: matrix3 columns + translation. */
typedef struct { v3 x, y, z, t; } OraAffine;
typedef struct { v3 x, y, z; } OraMat3;

/* prim.rs:18-25 */
typedef struct { float t; v3 outward; float u, v; uint32_t geom_id, prim_id; } OraPrimHit;
/* scene.rs:134-142 */
typedef struct { float t; v3 normal; int front_face; float u, v; uint32_t geom_id, prim_id; } OraRayHit;

/* bvh.rs:53-57 traversal-stats mirror: index 0 = top level, 1 = inside an instance. */
typedef struct {
  uint64_t queries[2], nodes[2], leaves[2], packets[2], prims[2];
  uint64_t accepted_hits, instance_descents, stack_high_water;
} OraTravStats;

typedef struct OraScene OraScene;
typedef struct OraBuilder OraBuilder;

OraBuilder *ora_builder_new(void);                                       /* scene.rs:152 */
uint32_t ora_attach_triangles(OraBuilder *b, const float *verts, size_t nverts, const uint32_t *idx,
                              size_t ntris, const float *normals, size_t nnormals, uint32_t mask); /* scene.rs:163 */
uint32_t ora_attach_sphere(OraBuilder *b, float cx, float cy, float cz, float r, uint32_t mask);
uint32_t ora_attach_instance(OraBuilder *b, OraScene *scene, const float l2w[12], const float *l2w_end,
                             uint32_t mask);
uint32_t ora_attach_empty(OraBuilder *b, uint32_t mask);                 /* scene.rs:205 */
/* set_geometry keeps the slot's mask (scene.rs:199-201); returns -1 for an unknown id (the reference panics). */
int ora_set_triangles(OraBuilder *b, uint32_t id, const float *verts, size_t nverts, const uint32_t *idx,
                      size_t ntris, const float *normals, size_t nnormals);
int ora_set_sphere(OraBuilder *b, uint32_t id, float cx, float cy, float cz, float r);
int ora_set_instance(OraBuilder *b, uint32_t id, OraScene *scene, const float l2w[12], const float *l2w_end);
size_t ora_builder_count(const OraBuilder *b);                           /* scene.rs:179 */
OraScene *ora_commit(OraBuilder *b);                                     /* scene.rs:226 (consumes b) */
void ora_scene_free(OraScene *s);

int ora_intersect(const OraScene *s, const OraRay *ray, float t_min, float t_max, OraRayHit *out); /* scene.rs:354 */
int ora_occluded(const OraScene *s, const OraRay *ray, float t_min, float t_max);                  /* scene.rs:370 */
int ora_scene_bounds(const OraScene *s, float out[6]);                   /* scene.rs:375 */
uint32_t ora_geometry_count(const OraScene *s);                          /* scene.rs:380 */
int ora_has_motion(const OraScene *s);                                   /* scene.rs:395 */
size_t ora_primitive_count(const OraScene *s);                           /* scene.rs:400 */
/* out: triangles, spheres, curve_segments, cubic_curve_spans, instances */
void ora_primitive_breakdown(const OraScene *s, size_t out[5]);          /* scene.rs:409 */
size_t ora_primitive_extents(const OraScene *s, float out[3]);              /* scene.rs:446-455 -> count; out = scene, mean, max diagonal */
void ora_unique_primitive_breakdown(const OraScene *s, size_t out[5]);   /* scene.rs:422-427, bvh.rs:397-416 */

/* Batched helpers for tests / cpu_baseline (rays: 8 floats o,d,time,mask-bits; hits: 8 x 4-byte
 * t,nx,ny,nz,u,v,geom,prim + front flags array). */
void ora_intersect_n(const OraScene *s, const float *rays, size_t n, float t_min, float t_max, float *hit_f,
                     uint32_t *hit_ids, uint8_t *front);
void ora_occluded_n(const OraScene *s, const float *rays, size_t n, float t_min, float t_max, uint8_t *out);
/* Brute-force closest hit over the scene's top-level primitives (bvh.rs:1460-1470 linear_scan). */
int ora_linear_scan(const OraScene *s, const OraRay *ray, float t_min, float t_max, OraRayHit *out);

/* Thread-local stats sink (NULL = off). */
void ora_set_trav_stats(OraTravStats *st);
/* A second thread-local sink for the any-hit queries (ora_occluded): while set, they count there instead. */
void ora_set_trav_stats_any(OraTravStats *st);

/* ---- structure introspection (for the build-parity tests) ---- */
typedef struct {
  float bmin[3][4];
  float bmax[3][4];
  uint32_t child[4];
  uint32_t flags;
  uint32_t pad[3];
} OraWideNode; /* bvh.rs:196-215 (128 bytes) */
typedef struct { uint32_t pkt_first, pkt_count, idx_first, idx_count; } OraLeaf; /* bvh.rs:220-227 */
typedef struct {
  float v[3][3][4]; /* v[vertex][axis][lane] */
  uint32_t prim[4];
  uint32_t active, mask_and, mask_or;
  uint32_t masks[4];
  uint32_t pad;
} OraTri4; /* triangle.rs:182-196 (192 bytes) */

size_t ora_bvh_counts(const OraScene *s, size_t out[5]); /* nodes, leaves, packets, indices, prims */
const OraWideNode *ora_bvh_nodes(const OraScene *s);
const OraLeaf *ora_bvh_leaves(const OraScene *s);
const OraTri4 *ora_bvh_packets(const OraScene *s);
const uint32_t *ora_bvh_indices(const OraScene *s);

/* ---- unit-level entry points for the known-answer tests ---- */
int ora_triangle_intersect(const OraRay *ray, const float v0[3], const float v1[3], const float v2[3], float t_min,
                           float t_max, float tuv[3]); /* triangle.rs:97-106 */
/* Tri4::new + Tri4::intersect on up to 4 triangles (triangle.rs:217-348); tris = n*(9 floats), masks[n]. */
void ora_tri4_intersect(const OraRay *ray, const float *tris, const uint32_t *masks, int n, uint32_t ray_mask,
                        float t_min, float t_max, uint32_t *hits, uint32_t *fallback, uint32_t *active, float t[4],
                        float u[4], float v[4]);
int ora_clip_triangle_aabb(const float v0[3], const float v1[3], const float v2[3], int axis, float mn, float mx,
                           float out[6]); /* triangle.rs:367-429 */
void ora_affine_inverse(const float m[12], float out[12]);

#endif
