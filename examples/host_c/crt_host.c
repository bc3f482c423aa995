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
#include "scene_file.h"

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: crt_host <scene.bin> <film.bin>\n"); return 2; }
  World w;
  const int rc = world_load(argv[1], &w);
  if (rc) return rc;
  const CrtRenderSettings settings = w.settings;
  const uint32_t spp = w.spp, batch = w.batch;

  CrtRenderer *r = crt_renderer_new(w.scene, w.materials, w.n_geoms, w.lights, w.n_lights, &w.camera, &settings, 0, 1);
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
  world_free(&w);
  free(rgb); free(frame); free(pix);
  return 0;
}
