/* Scene files of the example hosts (crt_host.c, ../host_rccl/crt_rccl_host.cpp): the content crust-render_amd/usda.py's
 * build_world feeds through the Python mirror, as one little-endian blob (tests/host_c_scene.py writes them) —
 *   "CRTS", n_protos, n_geoms
 *   prototypes: kind 0 mesh (n_verts, n_tris, verts, indices) | 1 sphere (center, radius) | 2 instances (n x proto, mask, l2w[12])
 *   geometries: kind 0 mesh | 1 sphere | 2 instance (proto, has_end, l2w[12], [l2w_end[12]]) | 3 empty; each: mask, then CrtMaterial
 *   n_lights x CrtLight; Camera::new's 13 arguments; CrtRenderSettings; spp, batch
 * — and the reference's SceneBuilder call sequence for it (scene.rs:152-341, rt_world.rs:111-185). Plain C99 / C++. */
#ifndef CRT_EXAMPLE_SCENE_FILE_H
#define CRT_EXAMPLE_SCENE_FILE_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "crt.h"

typedef struct { const unsigned char *p, *end; } Reader;
static int take(Reader *r, void *dst, size_t n) {
  if ((size_t)(r->end - r->p) < n) return 0;
  memcpy(dst, r->p, n); r->p += n; return 1;
}
static const void *span(Reader *r, size_t n) {
  if ((size_t)(r->end - r->p) < n) return NULL;
  const void *q = r->p; r->p += n; return q;
}
static uint32_t u32(Reader *r, int *ok) { uint32_t v = 0; if (!take(r, &v, 4)) *ok = 0; return v; }

static int fail_lib(const char *what) {
  fprintf(stderr, "crt_host: %s failed: %s\n", what, crt_last_error());
  return 3;
}

/* mesh: n_verts, n_tris, verts, indices | sphere: center, radius */
static int attach_mesh(Reader *r, CrtBuilder *b, uint32_t mask) {
  int ok = 1;
  const uint32_t nv = u32(r, &ok), nt = u32(r, &ok);
  if (!ok) return 0;
  const float *verts = (const float *)span(r, (size_t)nv * 12);
  const uint32_t *idx = (const uint32_t *)span(r, (size_t)nt * 12);
  uint32_t id;
  return verts && idx && crt_attach_triangles(b, verts, nv, idx, nt, NULL, 0, mask, &id) == CRT_OK;
}
static int attach_sphere(Reader *r, CrtBuilder *b, uint32_t mask) {
  float cr[4];
  uint32_t id;
  return take(r, cr, 16) && crt_attach_sphere(b, cr, cr[3], mask, &id) == CRT_OK;
}

typedef struct {
  unsigned char *buf;
  CrtScene **protos; uint32_t n_protos;
  CrtScene *scene;
  CrtMaterial *materials; uint32_t n_geoms;
  const CrtLight *lights; uint32_t n_lights;
  CrtCamera camera;
  CrtRenderSettings settings;
  uint32_t spp, batch;
} World;

/* 0 ok, 2 bad file, 3 library error (already reported on stderr) */
static int world_load(const char *path, World *w) {
  memset(w, 0, sizeof *w);
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); return 2; }
  fseek(f, 0, SEEK_END);
  const long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *buf = (unsigned char *)malloc((size_t)size + 16);
  if (!buf || fread(buf, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "crt_host: cannot read %s\n", path); fclose(f); return 2; }
  fclose(f);
  w->buf = buf;
  Reader rd = {buf, buf + size};
  int ok = 1;
  if (u32(&rd, &ok) != 0x53545243u /* "CRTS" */) { fprintf(stderr, "crt_host: not a scene file\n"); return 2; }
  const uint32_t n_protos = u32(&rd, &ok), n_geoms = u32(&rd, &ok);
  if (!ok) return 2;

  /* prototype scenes (MeshArena::committed_scene, usd_import.rs:891-909): a mesh, a sphere, or earlier prototypes placed */
  CrtScene **protos = (CrtScene **)calloc(n_protos ? n_protos : 1, sizeof *protos);
  w->protos = protos; w->n_protos = n_protos;
  for (uint32_t i = 0; i < n_protos; i++) {
    CrtBuilder *b = crt_builder_new();
    if (!b) return fail_lib("crt_builder_new");
    const uint32_t kind = u32(&rd, &ok);
    if (kind == 0) ok = ok && attach_mesh(&rd, b, CRT_MASK_ALL);
    else if (kind == 1) ok = ok && attach_sphere(&rd, b, CRT_MASK_ALL);
    else {
      const uint32_t n = u32(&rd, &ok);
      for (uint32_t k = 0; ok && k < n; k++) {
        const uint32_t proto = u32(&rd, &ok), mask = u32(&rd, &ok);
        const float *l2w = (const float *)span(&rd, 48);
        uint32_t id;
        ok = ok && l2w && proto < i && crt_attach_instance(b, protos[proto], l2w, NULL, mask, &id) == CRT_OK;
      }
    }
    if (!ok) { fprintf(stderr, "crt_host: prototype %u: bad record (%s)\n", i, crt_last_error()); crt_builder_free(b); return 2; }
    protos[i] = crt_commit(b);
    if (!protos[i]) return fail_lib("crt_commit (prototype)");
  }

  /* the world: one geometry and one material per geom_id (WorldBuilder::attach_masked, rt_world.rs:111-185) */
  CrtBuilder *b = crt_builder_new();
  if (!b) return fail_lib("crt_builder_new");
  CrtMaterial *mats = (CrtMaterial *)calloc(n_geoms ? n_geoms : 1, sizeof *mats);
  w->materials = mats; w->n_geoms = n_geoms;
  if (crt_reserve(b, n_geoms) != CRT_OK) return fail_lib("crt_reserve");
  for (uint32_t g = 0; g < n_geoms; g++) {
    const uint32_t kind = u32(&rd, &ok), mask = u32(&rd, &ok);
    uint32_t id = 0;
    if (kind == 0) ok = ok && attach_mesh(&rd, b, mask);
    else if (kind == 1) ok = ok && attach_sphere(&rd, b, mask);
    else if (kind == 2) {
      const uint32_t proto = u32(&rd, &ok), has_end = u32(&rd, &ok);
      const float *l2w = (const float *)span(&rd, 48);
      const float *l2w_end = has_end ? (const float *)span(&rd, 48) : NULL;
      ok = ok && l2w && (!has_end || l2w_end) && proto < n_protos &&
           crt_attach_instance(b, protos[proto], l2w, l2w_end, mask, &id) == CRT_OK;
    } else ok = ok && crt_attach_empty(b, mask, &id) == CRT_OK;
    ok = ok && take(&rd, &mats[g], sizeof(CrtMaterial));
    if (!ok) { fprintf(stderr, "crt_host: geometry %u: bad record (%s)\n", g, crt_last_error()); crt_builder_free(b); return 2; }
  }
  if (crt_count(b) != n_geoms) { fprintf(stderr, "crt_host: builder holds %zu geometries, file says %u\n", crt_count(b), n_geoms); return 2; }
  w->scene = crt_commit(b);
  if (!w->scene) return fail_lib("crt_commit");

  w->n_lights = u32(&rd, &ok);
  w->lights = (const CrtLight *)span(&rd, (size_t)w->n_lights * sizeof(CrtLight));
  float cam[13];
  ok = ok && (w->lights || !w->n_lights) && take(&rd, cam, sizeof cam) && take(&rd, &w->settings, sizeof w->settings);
  w->spp = u32(&rd, &ok); w->batch = u32(&rd, &ok);
  if (!ok || w->batch == 0) { fprintf(stderr, "crt_host: truncated scene file\n"); return 2; }
  crt_camera_new(&w->camera, cam, cam + 3, cam + 6, cam[9], cam[10], cam[11], cam[12]);
  return 0;
}
static void world_free(World *w) {
  if (w->scene) crt_scene_release(w->scene);
  for (uint32_t i = 0; i < w->n_protos; i++) if (w->protos[i]) crt_scene_release(w->protos[i]);
  free(w->protos); free(w->materials); free(w->buf);
  memset(w, 0, sizeof *w);
}
#endif
