// extern "C" surface of libcrt_amd.so: builder / scene / query entry points (include/crt.h).
// The renderer entry points live in render.cpp.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <unordered_set>

#include "crt_internal.h"

namespace crt {
int traversal_error_check(void *stream);
const char *last_error_text();
}

using namespace crt;

namespace {

Affine affine_from12(const float m[12]) {
  return Affine{f3(m[0], m[1], m[2]), f3(m[3], m[4], m[5]), f3(m[6], m[7], m[8]), f3(m[9], m[10], m[11])};
}

void fill_mesh(Geom &g, const float *verts, size_t n_verts, const uint32_t *indices, size_t n_tris, const float *normals,
               size_t n_normals) {
  g.kind = G_MESH;
  g.verts.assign(verts, verts + (verts ? 3 * n_verts : 0));
  g.idx.assign(indices, indices + (indices ? 3 * n_tris : 0));
  g.has_normals = normals != nullptr;
  g.normals.clear();
  if (normals) g.normals.assign(normals, normals + 3 * n_normals);
  g.scene.reset();
}
void fill_sphere(Geom &g, const float c[3], float r) {
  g = Geom{G_SPHERE, g.mask};
  g.center = f3(c[0], c[1], c[2]);
  g.radius = r;
}
void fill_instance(Geom &g, CrtScene *scene, const float l2w[12], const float *l2w_end) {
  g = Geom{G_INSTANCE, g.mask};
  g.scene = scene->p;
  g.l2w = affine_from12(l2w);
  g.has_end = l2w_end != nullptr;
  if (l2w_end) g.l2w_end = affine_from12(l2w_end);
}

}  // namespace

extern "C" {

const char *crt_version(void) { return "crt_amd 0.1 (gfx950)"; }
const char *crt_last_error(void) { return crt::last_error_text(); }

int crt_device_info(char *name_out, size_t name_cap, int *cu_count, size_t *hbm_bytes) {
  if (!device_ok()) return CRT_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return CRT_ERR_NO_DEVICE;
  if (name_out && name_cap) {
    std::strncpy(name_out, prop.gcnArchName, name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return CRT_OK;
}

CrtBuilder *crt_builder_new(void) { return new (std::nothrow) CrtBuilder(); }
void crt_builder_free(CrtBuilder *b) { delete b; }
int crt_reserve(CrtBuilder *b, size_t additional) {
  if (!b) return CRT_ERR_BAD_ARG;
  b->b.geoms.reserve(b->b.geoms.size() + additional);
  return CRT_OK;
}
size_t crt_count(const CrtBuilder *b) { return b ? b->b.geoms.size() : 0; }

int crt_attach_triangles(CrtBuilder *b, const float *verts, size_t n_verts, const uint32_t *indices, size_t n_tris,
                         const float *normals, size_t n_normals, uint32_t mask, uint32_t *geom_id_out) {
  if (!b || (n_verts && !verts) || (n_tris && !indices)) return CRT_ERR_BAD_ARG;
  Geom g;
  g.mask = mask;
  fill_mesh(g, verts, n_verts, indices, n_tris, normals, n_normals);
  b->b.geoms.push_back(std::move(g));
  if (geom_id_out) *geom_id_out = uint32_t(b->b.geoms.size() - 1);
  return CRT_OK;
}
int crt_attach_sphere(CrtBuilder *b, const float center[3], float radius, uint32_t mask, uint32_t *geom_id_out) {
  if (!b || !center) return CRT_ERR_BAD_ARG;
  Geom g;
  g.mask = mask;
  fill_sphere(g, center, radius);
  b->b.geoms.push_back(std::move(g));
  if (geom_id_out) *geom_id_out = uint32_t(b->b.geoms.size() - 1);
  return CRT_OK;
}
int crt_attach_instance(CrtBuilder *b, CrtScene *scene, const float l2w[12], const float *l2w_end, uint32_t mask,
                        uint32_t *geom_id_out) {
  if (!b || !scene || !l2w) return CRT_ERR_BAD_ARG;
  Geom g;
  g.mask = mask;
  fill_instance(g, scene, l2w, l2w_end);
  b->b.geoms.push_back(std::move(g));
  if (geom_id_out) *geom_id_out = uint32_t(b->b.geoms.size() - 1);
  return CRT_OK;
}
int crt_attach_empty(CrtBuilder *b, uint32_t mask, uint32_t *geom_id_out) {
  return crt_attach_triangles(b, nullptr, 0, nullptr, 0, nullptr, 0, mask, geom_id_out);
}
int crt_set_triangles(CrtBuilder *b, uint32_t id, const float *verts, size_t n_verts, const uint32_t *indices,
                      size_t n_tris, const float *normals, size_t n_normals) {
  if (!b || (n_verts && !verts) || (n_tris && !indices)) return CRT_ERR_BAD_ARG;
  if (id >= b->b.geoms.size()) return CRT_ERR_BAD_ID;
  fill_mesh(b->b.geoms[id], verts, n_verts, indices, n_tris, normals, n_normals);
  return CRT_OK;
}
int crt_set_sphere(CrtBuilder *b, uint32_t id, const float center[3], float radius) {
  if (!b || !center) return CRT_ERR_BAD_ARG;
  if (id >= b->b.geoms.size()) return CRT_ERR_BAD_ID;
  fill_sphere(b->b.geoms[id], center, radius);
  return CRT_OK;
}
int crt_set_instance(CrtBuilder *b, uint32_t id, CrtScene *scene, const float l2w[12], const float *l2w_end) {
  if (!b || !scene || !l2w) return CRT_ERR_BAD_ARG;
  if (id >= b->b.geoms.size()) return CRT_ERR_BAD_ID;
  fill_instance(b->b.geoms[id], scene, l2w, l2w_end);
  return CRT_OK;
}

CrtScene *crt_commit(CrtBuilder *b) {
  if (!b) return nullptr;
  CrtScene *s = new (std::nothrow) CrtScene();
  if (s) s->p = commit(std::move(b->b));
  delete b;
  return s;
}

void crt_scene_retain(CrtScene *s) {
  if (s) s->refs.fetch_add(1);
}
void crt_scene_release(CrtScene *s) {
  if (s && s->refs.fetch_sub(1) == 1) delete s;
}

int crt_scene_bounds(const CrtScene *s, float out[6]) {
  if (!s || !out) return CRT_ERR_BAD_ARG;
  const Bvh &b = s->p->bvh;
  if (!b.has_bbox) return 0;
  out[0] = b.root_bbox.mn.x; out[1] = b.root_bbox.mn.y; out[2] = b.root_bbox.mn.z;
  out[3] = b.root_bbox.mx.x; out[4] = b.root_bbox.mx.y; out[5] = b.root_bbox.mx.z;
  return 1;
}
uint32_t crt_scene_geometry_count(const CrtScene *s) { return s ? s->p->n_geoms : 0; }
int crt_scene_has_motion(const CrtScene *s) { return s ? (s->p->has_motion ? 1 : 0) : 0; }
size_t crt_scene_primitive_count(const CrtScene *s) { return s ? s->p->bvh.prims.size() : 0; }
int crt_scene_primitive_breakdown(const CrtScene *s, size_t out[5]) {
  if (!s || !out) return CRT_ERR_BAD_ARG;
  out[0] = out[1] = out[2] = out[3] = out[4] = 0;
  for (const Prim &p : s->p->bvh.prims) {
    if (p.kind == PRIM_TRI) out[0]++;
    else if (p.kind == PRIM_SPHERE) out[1]++;
    else out[4]++;
  }
  return CRT_OK;
}
namespace {
void accumulate_unique(const Scene &sc, std::unordered_set<const Scene *> &visited, size_t acc[5]) {  // bvh.rs:397-416
  for (const Prim &p : sc.bvh.prims) {
    if (p.kind == PRIM_TRI) acc[0]++;
    else if (p.kind == PRIM_SPHERE) acc[1]++;
    else {
      acc[4]++;
      if (visited.insert(p.scene.get()).second) accumulate_unique(*p.scene, visited, acc);
    }
  }
}
}  // namespace
int crt_scene_unique_primitive_breakdown(const CrtScene *s, size_t out[5]) {
  if (!s || !out) return CRT_ERR_BAD_ARG;
  out[0] = out[1] = out[2] = out[3] = out[4] = 0;
  std::unordered_set<const Scene *> visited;
  accumulate_unique(*s->p, visited, out);
  return CRT_OK;
}
int crt_scene_memory_footprint(CrtScene *s, size_t out[6]) {
  if (!s || !out) return CRT_ERR_BAD_ARG;
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  const DeviceImage &d = *s->p->dev;
  out[0] = d.bytes[4];               // prim records
  out[1] = d.bytes[5] + d.bytes[6];  // instance records + shading normals
  out[2] = d.bytes[0];
  out[3] = d.bytes[1];
  out[4] = d.bytes[2];
  out[5] = d.bytes[3];
  return CRT_OK;
}
int crt_scene_tree(const CrtScene *s, size_t counts[5], const void **nodes128, const void **leaves16,
                   const void **packets192, const uint32_t **indices) {
  if (!s || !counts) return CRT_ERR_BAD_ARG;
  const Bvh &b = s->p->bvh;
  counts[0] = b.wide.size(); counts[1] = b.leaves.size(); counts[2] = b.packets.size();
  counts[3] = b.indices.size(); counts[4] = b.prims.size();
  if (nodes128) *nodes128 = b.wide.data();
  if (leaves16) *leaves16 = b.leaves.data();
  if (packets192) *packets192 = b.packets.data();
  if (indices) *indices = b.indices.data();
  return CRT_OK;
}

int crt_intersect_n(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, CrtRayHit *d_hits,
                    void *stream) {
  if (!s || (n && (!d_rays || !d_hits))) return CRT_ERR_BAD_ARG;
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  return launch_intersect_n(s->p->dev->view, d_rays, n, t_min, t_max, d_hits, stream, nullptr);
}
int crt_occluded_n(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                   void *stream) {
  if (!s || (n && (!d_rays || !d_out))) return CRT_ERR_BAD_ARG;
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  return launch_occluded_n(s->p->dev->view, d_rays, n, t_min, t_max, d_out, stream, nullptr);
}

static int with_stats(CrtScene *s, void *stream, CrtTravStats *host_stats,
                      int (*launch)(const DevScene &, void *, CrtTravStats *, void *), void *ctx) {
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  CrtTravStats *d = nullptr;
  if (hipMalloc(&d, sizeof(CrtTravStats)) != hipSuccess) return CRT_ERR_NO_DEVICE;
  (void)hipMemsetAsync(d, 0, sizeof(CrtTravStats), (hipStream_t)stream);
  rc = launch(s->p->dev->view, stream, d, ctx);
  CrtTravStats h;
  if (rc == CRT_OK && hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
    rc = CRT_ERR_NO_DEVICE;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = CRT_ERR_NO_DEVICE;
  (void)hipFree(d);
  if (rc == CRT_OK && host_stats) {
    for (int k = 0; k < 2; k++) {
      host_stats->queries[k] += h.queries[k]; host_stats->nodes[k] += h.nodes[k];
      host_stats->leaves[k] += h.leaves[k]; host_stats->packets[k] += h.packets[k];
      host_stats->prims[k] += h.prims[k];
    }
    host_stats->accepted_hits += h.accepted_hits;
    host_stats->instance_descents += h.instance_descents;
    host_stats->rays += h.rays;
    for (int k = 0; k < 8; k++) { host_stats->phase_waves[k] += h.phase_waves[k]; host_stats->phase_lanes[k] += h.phase_lanes[k]; host_stats->phase_cycles[k] += h.phase_cycles[k]; }
  }
  return rc;
}

struct NArgs { const CrtRay *rays; size_t n; float t_min, t_max; void *out; };

int crt_intersect_n_stats(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, CrtRayHit *d_hits,
                          void *stream, CrtTravStats *host_stats) {
  if (!s || (n && (!d_rays || !d_hits))) return CRT_ERR_BAD_ARG;
  NArgs a{d_rays, n, t_min, t_max, d_hits};
  return with_stats(
      s, stream, host_stats,
      [](const DevScene &v, void *st, CrtTravStats *d, void *c) {
        NArgs *a = static_cast<NArgs *>(c);
        return launch_intersect_n(v, a->rays, a->n, a->t_min, a->t_max, static_cast<CrtRayHit *>(a->out), st, d);
      },
      &a);
}
int crt_occluded_n_stats(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                         void *stream, CrtTravStats *host_stats) {
  if (!s || (n && (!d_rays || !d_out))) return CRT_ERR_BAD_ARG;
  NArgs a{d_rays, n, t_min, t_max, d_out};
  return with_stats(
      s, stream, host_stats,
      [](const DevScene &v, void *st, CrtTravStats *d, void *c) {
        NArgs *a = static_cast<NArgs *>(c);
        return launch_occluded_n(v, a->rays, a->n, a->t_min, a->t_max, static_cast<uint32_t *>(a->out), st, d);
      },
      &a);
}

// Single-ray forms: stage through a small device buffer and synchronise. Meant for drop-in use by a
// per-pixel host integrator and for API-semantics tests, not for throughput.
int crt_intersect1(CrtScene *s, const CrtRay *ray, float t_min, float t_max, CrtRayHit *hit) {
  if (!s || !ray || !hit) return CRT_ERR_BAD_ARG;
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  char *d = nullptr;
  if (hipMalloc(&d, sizeof(CrtRay) + sizeof(CrtRayHit)) != hipSuccess) return CRT_ERR_NO_DEVICE;
  CrtRay *dr = reinterpret_cast<CrtRay *>(d);
  CrtRayHit *dh = reinterpret_cast<CrtRayHit *>(d + sizeof(CrtRay));
  rc = hipMemcpy(dr, ray, sizeof(CrtRay), hipMemcpyHostToDevice) == hipSuccess ? CRT_OK : CRT_ERR_NO_DEVICE;
  if (rc == CRT_OK) rc = launch_intersect_n(s->p->dev->view, dr, 1, t_min, t_max, dh, nullptr, nullptr);
  if (rc == CRT_OK) rc = traversal_error_check(nullptr);
  if (rc == CRT_OK && hipMemcpy(hit, dh, sizeof(CrtRayHit), hipMemcpyDeviceToHost) != hipSuccess) rc = CRT_ERR_NO_DEVICE;
  (void)hipFree(d);
  if (rc != CRT_OK) return rc;
  return hit->geom_id != CRT_INVALID_ID ? 1 : 0;
}
int crt_occluded1(CrtScene *s, const CrtRay *ray, float t_min, float t_max) {
  if (!s || !ray) return CRT_ERR_BAD_ARG;
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  char *d = nullptr;
  if (hipMalloc(&d, sizeof(CrtRay) + sizeof(uint32_t)) != hipSuccess) return CRT_ERR_NO_DEVICE;
  CrtRay *dr = reinterpret_cast<CrtRay *>(d);
  uint32_t *dout = reinterpret_cast<uint32_t *>(d + sizeof(CrtRay));
  uint32_t h = 0;
  rc = hipMemcpy(dr, ray, sizeof(CrtRay), hipMemcpyHostToDevice) == hipSuccess ? CRT_OK : CRT_ERR_NO_DEVICE;
  if (rc == CRT_OK) rc = launch_occluded_n(s->p->dev->view, dr, 1, t_min, t_max, dout, nullptr, nullptr);
  if (rc == CRT_OK) rc = traversal_error_check(nullptr);
  if (rc == CRT_OK && hipMemcpy(&h, dout, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) rc = CRT_ERR_NO_DEVICE;
  (void)hipFree(d);
  if (rc != CRT_OK) return rc;
  return h ? 1 : 0;
}

size_t crt_shard_pixels(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t *out) {
  if (world == 0 || rank >= world) return 0;
  const uint32_t tx = (width + 15) / 16, ty = (height + 15) / 16;
  size_t n = 0;
  for (uint32_t t = 0; t < tx * ty; t++) {
    if (t % world != rank) continue;
    const uint32_t x0 = (t % tx) * 16, y0 = (t / tx) * 16;
    for (uint32_t y = y0; y < y0 + 16 && y < height; y++)
      for (uint32_t x = x0; x < x0 + 16 && x < width; x++) {
        if (out) out[n] = y * width + x;
        n++;
      }
  }
  return n;
}

void crt_material_default(CrtMaterial *m) {  // openpbr.rs:130-173
  if (!m) return;
  std::memset(m, 0, sizeof *m);
  m->kind = CRT_MAT_OPENPBR;
  m->base_weight = 1.0f;
  m->base_color[0] = m->base_color[1] = m->base_color[2] = 0.8f;
  m->specular_weight = 1.0f;
  m->specular_color[0] = m->specular_color[1] = m->specular_color[2] = 1.0f;
  m->specular_roughness = 0.3f;
  m->specular_ior = 1.5f;
  m->transmission_color[0] = m->transmission_color[1] = m->transmission_color[2] = 1.0f;
  m->transmission_dispersion_abbe_number = 20.0f;
  m->subsurface_color[0] = m->subsurface_color[1] = m->subsurface_color[2] = 0.8f;
  m->subsurface_radius = 1.0f;
  m->subsurface_radius_scale[0] = 1.0f; m->subsurface_radius_scale[1] = 0.5f; m->subsurface_radius_scale[2] = 0.25f;
  m->fuzz_color[0] = m->fuzz_color[1] = m->fuzz_color[2] = 1.0f;
  m->fuzz_roughness = 0.5f;
  m->coat_color[0] = m->coat_color[1] = m->coat_color[2] = 1.0f;
  m->coat_ior = 1.6f;
  m->coat_darkening = 1.0f;
  m->thin_film_thickness = 0.5f;
  m->thin_film_ior = 1.4f;
  m->emission_color[0] = m->emission_color[1] = m->emission_color[2] = 1.0f;
  m->geometry_opacity = 1.0f;
}

}  // extern "C"
