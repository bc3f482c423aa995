/* A host WITHOUT Python, torch or HIP calls of its own: plain C over include/crt.h.
 *
 *   crt_host <scene.bin> <film.bin>
 *
 * Reads a scene description (the same content crust-render_amd/usda.py's build_world feeds through the Python mirror:
 * prototype scenes, geometries with their ray masks and materials, lights, camera, RenderSettings), drives the
 * reference's own call sequence through the C ABI —
 *   SceneBuilder::new / attach_masked / commit          scene.rs:152-341   crt_builder_new, crt_attach_*, crt_commit
 *   Camera::new                                          camera.rs:27-63    crt_camera_new
 *   Renderer::new + render                               tracer.rs:137-148, :405-470   crt_renderer_new, crt_render_samples
 *   film resolve, RayStats                               tracer.rs:630-634, stats.rs:128-147   crt_film_read, crt_render_stats
 * — and writes the frame (width x height x 3 floats, buffer order) followed by the eight RayStats counters.
 * tests/test_gpu_host_c.py compares that file bit for bit with the Python host's image and with the oracle's.
 * What a Rust maintainer binds is exactly this sequence. Exit codes: 0 ok, 2 bad file, 3 library error
 * (message from crt_last_error: without a gfx950 device nothing is rendered — there is no CPU fallback). */
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

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: crt_host <scene.bin> <film.bin>\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END);
  const long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *buf = (unsigned char *)malloc((size_t)size + 16);
  if (!buf || fread(buf, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "crt_host: cannot read %s\n", argv[1]); return 2; }
  fclose(f);
  Reader rd = {buf, buf + size};
  int ok = 1;
  if (u32(&rd, &ok) != 0x53545243u /* "CRTS" */) { fprintf(stderr, "crt_host: not a scene file\n"); return 2; }
  const uint32_t n_protos = u32(&rd, &ok), n_geoms = u32(&rd, &ok);
  if (!ok) return 2;

  /* prototype scenes (MeshArena::committed_scene, usd_import.rs:891-909): a mesh, a sphere, or earlier prototypes placed */
  CrtScene **protos = (CrtScene **)calloc(n_protos ? n_protos : 1, sizeof *protos);
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
    if (!ok) { fprintf(stderr, "crt_host: prototype %u: bad record (%s)\n", i, crt_last_error()); return 2; }
    protos[i] = crt_commit(b);
    if (!protos[i]) return fail_lib("crt_commit (prototype)");
  }

  /* the world: one geometry and one material per geom_id (WorldBuilder::attach_masked, rt_world.rs:111-185) */
  CrtBuilder *b = crt_builder_new();
  if (!b) return fail_lib("crt_builder_new");
  CrtMaterial *mats = (CrtMaterial *)calloc(n_geoms ? n_geoms : 1, sizeof *mats);
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
    if (!ok) { fprintf(stderr, "crt_host: geometry %u: bad record (%s)\n", g, crt_last_error()); return 2; }
  }
  if (crt_count(b) != n_geoms) { fprintf(stderr, "crt_host: builder holds %zu geometries, file says %u\n", crt_count(b), n_geoms); return 2; }
  CrtScene *scene = crt_commit(b);
  if (!scene) return fail_lib("crt_commit");

  const uint32_t n_lights = u32(&rd, &ok);
  const CrtLight *lights = (const CrtLight *)span(&rd, (size_t)n_lights * sizeof(CrtLight));
  float cam[13];
  CrtRenderSettings settings;
  ok = ok && (lights || !n_lights) && take(&rd, cam, sizeof cam) && take(&rd, &settings, sizeof settings);
  const uint32_t spp = u32(&rd, &ok), batch = u32(&rd, &ok);
  if (!ok || batch == 0) { fprintf(stderr, "crt_host: truncated scene file\n"); return 2; }
  CrtCamera camera;
  crt_camera_new(&camera, cam, cam + 3, cam + 6, cam[9], cam[10], cam[11], cam[12]);

  CrtRenderer *r = crt_renderer_new(scene, mats, n_geoms, lights, n_lights, &camera, &settings, 0, 1);
  if (!r) return fail_lib("crt_renderer_new");
  for (uint32_t s = 0; s < spp; s += batch)
    if (crt_render_samples(r, s, spp - s < batch ? spp - s : batch, NULL) != CRT_OK) return fail_lib("crt_render_samples");

  const size_t n_pix = crt_renderer_pixel_count(r);
  if (n_pix != (size_t)settings.width * settings.height) { fprintf(stderr, "crt_host: %zu pixels owned\n", n_pix); return 3; }
  float *rgb = (float *)malloc(n_pix * 12), *frame = (float *)calloc(n_pix, 12);
  uint32_t *pix = (uint32_t *)malloc(n_pix * 4);
  CrtRayStats stats;
  if (!rgb || !frame || !pix) return 2;
  if (crt_film_read(r, rgb) != CRT_OK) return fail_lib("crt_film_read");
  if (crt_renderer_pixel_indices(r, pix) != CRT_OK) return fail_lib("crt_renderer_pixel_indices");
  if (crt_render_stats(r, &stats) != CRT_OK) return fail_lib("crt_render_stats");
  for (size_t k = 0; k < n_pix; k++) memcpy(frame + 3 * (size_t)pix[k], rgb + 3 * k, 12);  /* tile order -> buffer order */

  FILE *o = fopen(argv[2], "wb");
  if (!o || fwrite(frame, 12, n_pix, o) != n_pix || fwrite(&stats, sizeof stats, 1, o) != 1) { perror(argv[2]); return 2; }
  fclose(o);
  uint32_t pipe[3] = {0, 0, 0};
  crt_renderer_pipeline(r, pipe);
  printf("crt_host: %s: %ux%u, %u spp in batches of %u, %llu closest-hit + %llu shadow rays, last batch %s\n", crt_version(),
         settings.width, settings.height, spp, batch, (unsigned long long)stats.closest_hit, (unsigned long long)stats.shadow_rays,
         pipe[0] ? "fused" : "per stage");
  crt_renderer_free(r);
  crt_scene_release(scene);
  for (uint32_t i = 0; i < n_protos; i++) crt_scene_release(protos[i]);
  free(protos); free(mats); free(rgb); free(frame); free(pix); free(buf);
  return 0;
}
