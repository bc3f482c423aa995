// extern "C" surface of libcrt_amd.so: builder / scene / query entry points (include/crt.h).
// The renderer entry points live in render.cpp.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <unordered_set>

#include "crt_internal.h"

namespace crt {
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
  return abi_guard("crt_reserve", [&] {
    if (additional > b->b.geoms.max_size() - b->b.geoms.size()) throw std::length_error("capacity overflow");  // Vec::reserve panics on it
    b->b.geoms.reserve(b->b.geoms.size() + additional);
    return (int)CRT_OK;
  });
}
size_t crt_count(const CrtBuilder *b) { return b ? b->b.geoms.size() : 0; }

int crt_attach_triangles(CrtBuilder *b, const float *verts, size_t n_verts, const uint32_t *indices, size_t n_tris,
                         const float *normals, size_t n_normals, uint32_t mask, uint32_t *geom_id_out) {
  if (!b || (n_verts && !verts) || (n_tris && !indices)) return CRT_ERR_BAD_ARG;
  return abi_guard("crt_attach_triangles", [&] {
    Geom g;
    g.mask = mask;
    fill_mesh(g, verts, n_verts, indices, n_tris, normals, n_normals);
    b->b.geoms.push_back(std::move(g));
    if (geom_id_out) *geom_id_out = uint32_t(b->b.geoms.size() - 1);
    return (int)CRT_OK;
  });
}
int crt_attach_sphere(CrtBuilder *b, const float center[3], float radius, uint32_t mask, uint32_t *geom_id_out) {
  if (!b || !center) return CRT_ERR_BAD_ARG;
  return abi_guard("crt_attach_sphere", [&] {
    Geom g;
    g.mask = mask;
    fill_sphere(g, center, radius);
    b->b.geoms.push_back(std::move(g));
    if (geom_id_out) *geom_id_out = uint32_t(b->b.geoms.size() - 1);
    return (int)CRT_OK;
  });
}
int crt_attach_instance(CrtBuilder *b, CrtScene *scene, const float l2w[12], const float *l2w_end, uint32_t mask,
                        uint32_t *geom_id_out) {
  if (!b || !scene || !l2w) return CRT_ERR_BAD_ARG;
  return abi_guard("crt_attach_instance", [&] {
    Geom g;
    g.mask = mask;
    fill_instance(g, scene, l2w, l2w_end);
    b->b.geoms.push_back(std::move(g));
    if (geom_id_out) *geom_id_out = uint32_t(b->b.geoms.size() - 1);
    return (int)CRT_OK;
  });
}
int crt_attach_empty(CrtBuilder *b, uint32_t mask, uint32_t *geom_id_out) {
  return crt_attach_triangles(b, nullptr, 0, nullptr, 0, nullptr, 0, mask, geom_id_out);
}
int crt_set_triangles(CrtBuilder *b, uint32_t id, const float *verts, size_t n_verts, const uint32_t *indices,
                      size_t n_tris, const float *normals, size_t n_normals) {
  if (!b || (n_verts && !verts) || (n_tris && !indices)) return CRT_ERR_BAD_ARG;
  if (id >= b->b.geoms.size()) return CRT_ERR_BAD_ID;
  return abi_guard("crt_set_triangles", [&] {
    Geom g;  // built aside: the slot keeps its old geometry if an allocation fails
    g.mask = b->b.geoms[id].mask;
    fill_mesh(g, verts, n_verts, indices, n_tris, normals, n_normals);
    b->b.geoms[id] = std::move(g);
    return (int)CRT_OK;
  });
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
  CrtScene *s = nullptr;
  try {  // nothing may unwind through the C ABI: allocation or thread-creation failures become NULL + crt_last_error
    s = new CrtScene();
    s->p = commit(std::move(b->b));
    if (!s->p) { delete s; s = nullptr; }  // refused (instance nesting): commit left the reason in crt_last_error
  } catch (const std::exception &e) {
    set_error_text("commit: %s", e.what());
    delete s;
    s = nullptr;
  } catch (...) {
    set_error_text("commit: unknown failure");
    delete s;
    s = nullptr;
  }
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
int crt_scene_primitive_extents(const CrtScene *s, size_t *count, float *scene_diagonal, float *mean_diagonal,
                                float *max_diagonal) {  // scene.rs:446-455, bvh.rs:335-345
  if (!s) return CRT_ERR_BAD_ARG;
  const Bvh &b = s->p->bvh;
  auto length = [](F3 v) { return std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); };  // Vec3A::length
  float sum = 0.0f, mx = 0.0f;
  for (const Prim &p : b.prims) {
    const Aabb bb = prim_bbox(p);
    const float d = length(bb.mx - bb.mn);
    sum += d;
    mx = d > mx ? d : mx;  // f32::max
  }
  const size_t n = b.prims.size();
  if (count) *count = n;
  if (scene_diagonal) *scene_diagonal = b.has_bbox ? length(b.root_bbox.mx - b.root_bbox.mn) : 0.0f;
  if (mean_diagonal) *mean_diagonal = n == 0 ? 0.0f : sum / float(n);
  if (max_diagonal) *max_diagonal = mx;
  return CRT_OK;
}
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
int crt_scene_image_check(CrtScene *s, uint64_t out[8]) {
  if (!s || !out) return CRT_ERR_BAD_ARG;
  try {
    return scene_image_check(*s->p, out);
  } catch (const std::exception &e) {
    set_error_text("crt_scene_image_check: %s", e.what());
    return CRT_ERR_BAD_ARG;
  }
}
int crt_scene_engine_select(CrtScene *s, int want_wide, uint32_t out[8]) {
  if (!s || !out) return CRT_ERR_BAD_ARG;
  try {
    return scene_engine_select(*s->p, want_wide, out);
  } catch (const std::exception &e) {
    set_error_text("crt_scene_engine_select: %s", e.what());
    return CRT_ERR_BAD_ARG;
  }
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
  return launch_intersect_n(s->p->dev->view, d_rays, n, t_min, t_max, d_hits, stream, nullptr, s->p->dev->err);
}
int crt_occluded_n(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                   void *stream) {
  if (!s || (n && (!d_rays || !d_out))) return CRT_ERR_BAD_ARG;
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  return launch_occluded_n(s->p->dev->view, d_rays, n, t_min, t_max, d_out, stream, nullptr, s->p->dev->err);
}

// Reads and clears a device error word after the stream has drained: CRT_OK, or CRT_ERR_STACK when a traversal of
// the launches since the last read overflowed its stack (bit 0) or met instance nesting beyond the frames (bit 1).
static int take_error_word(uint32_t *d_err, void *stream) {
  uint32_t h = 0;
  if (hipMemcpyAsync(&h, d_err, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return CRT_ERR_NO_DEVICE;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return CRT_ERR_NO_DEVICE;
  if (!h) return CRT_OK;
  (void)hipMemsetAsync(d_err, 0, sizeof h, (hipStream_t)stream);
  set_error_text("traversal error word 0x%x (1 = stack overflow, 2 = instance nesting)", h);
  return CRT_ERR_STACK;
}

int crt_scene_traversal_error(CrtScene *s, void *stream) {
  if (!s) return CRT_ERR_BAD_ARG;
  if (!s->p->dev) return CRT_OK;  // never queried
  return take_error_word(s->p->dev->err, stream);
}

static int with_stats(CrtScene *s, void *stream, CrtTravStats *host_stats,
                      int (*launch)(const DevScene &, void *, CrtTravStats *, uint32_t *, void *), void *ctx) {
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  CrtTravStats *d = nullptr;
  if (hipMalloc(&d, sizeof(CrtTravStats)) != hipSuccess) return CRT_ERR_NO_DEVICE;
  (void)hipMemsetAsync(d, 0, sizeof(CrtTravStats), (hipStream_t)stream);
  rc = launch(s->p->dev->view, stream, d, s->p->dev->err, ctx);
  CrtTravStats h;
  if (rc == CRT_OK && hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
    rc = CRT_ERR_NO_DEVICE;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = CRT_ERR_NO_DEVICE;
  (void)hipFree(d);
  if (rc == CRT_OK) rc = take_error_word(s->p->dev->err, stream);  // these forms synchronise anyway: surface it
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
      [](const DevScene &v, void *st, CrtTravStats *d, uint32_t *e, void *c) {
        NArgs *a = static_cast<NArgs *>(c);
        return launch_intersect_n(v, a->rays, a->n, a->t_min, a->t_max, static_cast<CrtRayHit *>(a->out), st, d, e);
      },
      &a);
}
int crt_occluded_n_stats(CrtScene *s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                         void *stream, CrtTravStats *host_stats) {
  if (!s || (n && (!d_rays || !d_out))) return CRT_ERR_BAD_ARG;
  NArgs a{d_rays, n, t_min, t_max, d_out};
  return with_stats(
      s, stream, host_stats,
      [](const DevScene &v, void *st, CrtTravStats *d, uint32_t *e, void *c) {
        NArgs *a = static_cast<NArgs *>(c);
        return launch_occluded_n(v, a->rays, a->n, a->t_min, a->t_max, static_cast<uint32_t *>(a->out), st, d, e);
      },
      &a);
}

// Single-ray forms, for drop-in use by a per-pixel host integrator whose worker threads each call
// Scene::intersect / occluded one ray at a time (tracer.rs:428, :478; "Queries are &self and thread-safe",
// scene.rs:344). Every calling thread owns a persistent staging area — a pinned host record, a device record
// (ray, result, this thread's error word) and a non-blocking stream — created on its first call, so a query is
// one host-to-device copy, one launch and one device-to-host copy on the thread's own stream: no allocation, no
// device-wide synchronisation, nothing shared between threads. crt_thread_release frees the calling thread's area.
}  // extern "C"
namespace {
struct Staging {
  struct Rec { CrtRay ray; CrtRayHit hit; uint32_t occ; uint32_t err; };
  Rec *host = nullptr;   // pinned
  Rec *dev = nullptr;
  hipStream_t stream = nullptr;
  int device = -1;
  bool ok() const { return host && dev && stream; }
  void release() {
    if (stream) (void)hipStreamDestroy(stream);
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    host = nullptr; dev = nullptr; stream = nullptr; device = -1;
  }
  bool acquire() {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return false;
    if (ok() && device == cur) return true;
    release();
    if (!CRT_HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&host), sizeof(Rec), hipHostMallocDefault)) ||
        !CRT_HIP_OK(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(Rec))) ||
        !CRT_HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) ||
        // on the thread's OWN stream: a null-stream memset is asynchronous to the host and unordered against a
        // non-blocking stream, so it could land between the first query's ray upload and its kernel
        !CRT_HIP_OK(hipMemsetAsync(dev, 0, sizeof(Rec), stream))) {
      release();
      return false;
    }
    device = cur;
    return true;
  }
};
// Deliberately not freed by a thread_local destructor: at process exit that would run after the HIP runtime's own
// teardown. A thread that ends early calls crt_thread_release(); otherwise the area (≈ 200 bytes of HBM, one
// stream) lives as long as the process.
thread_local Staging t_staging;

template <bool ANY>
int query1(CrtScene *s, const CrtRay *ray, float t_min, float t_max, CrtRayHit *hit) {
  int rc = s->p->ensure_device();
  if (rc != CRT_OK) return rc;
  Staging &g = t_staging;
  if (!g.acquire()) return CRT_ERR_NO_DEVICE;
  g.host->ray = *ray;
  if (!CRT_HIP_OK(hipMemcpyAsync(&g.dev->ray, &g.host->ray, sizeof(CrtRay), hipMemcpyHostToDevice, g.stream)))
    return CRT_ERR_NO_DEVICE;
  rc = ANY ? launch_occluded_n(s->p->dev->view, &g.dev->ray, 1, t_min, t_max, &g.dev->occ, g.stream, nullptr, &g.dev->err)
           : launch_intersect_n(s->p->dev->view, &g.dev->ray, 1, t_min, t_max, &g.dev->hit, g.stream, nullptr, &g.dev->err);
  if (rc != CRT_OK) return rc;
  // result + error word in one copy (hit, occ, err are contiguous)
  if (!CRT_HIP_OK(hipMemcpyAsync(&g.host->hit, &g.dev->hit, sizeof(CrtRayHit) + 2 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                 g.stream)) ||
      !CRT_HIP_OK(hipStreamSynchronize(g.stream)))
    return CRT_ERR_NO_DEVICE;
  if (g.host->err) {  // this call's own word: another thread's or scene's overflow cannot show up here
    (void)hipMemsetAsync(&g.dev->err, 0, sizeof(uint32_t), g.stream);
    set_error_text("traversal error word 0x%x (1 = stack overflow, 2 = instance nesting)", g.host->err);
    return CRT_ERR_STACK;
  }
  if (ANY) return g.host->occ ? 1 : 0;
  *hit = g.host->hit;
  return hit->geom_id != CRT_INVALID_ID ? 1 : 0;
}
}  // namespace
extern "C" {

int crt_intersect1(CrtScene *s, const CrtRay *ray, float t_min, float t_max, CrtRayHit *hit) {
  if (!s || !ray || !hit) return CRT_ERR_BAD_ARG;
  return query1<false>(s, ray, t_min, t_max, hit);
}
int crt_occluded1(CrtScene *s, const CrtRay *ray, float t_min, float t_max) {
  if (!s || !ray) return CRT_ERR_BAD_ARG;
  return query1<true>(s, ray, t_min, t_max, nullptr);
}
void crt_thread_release(void) { t_staging.release(); }

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
