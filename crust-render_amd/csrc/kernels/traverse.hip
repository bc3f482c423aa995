// Batched Scene::intersect / Scene::occluded kernels (scene.rs:354-372) over the device image.
// Compile with -ffp-contract=off (see traverse.hip.h).
#include <cstdlib>

#include "traverse_pool.hip.h"

static_assert(crt::dev::kMaxLevels == (int)crt::kMaxInstanceLevels, "commit's nesting limit is the kernels' frame count");

namespace crt {

using namespace dev;

namespace {

// Rays are dealt to the workgroups in round-robin blocks of one workgroup's width: workgroup b takes blocks b, b + G,
// b + 2G ... of 256 consecutive rays, its k-th ray is ray ((k / 256) * G + b) * 256 + k % 256. Whatever order the caller's
// batch is in — pixel order, sorted, shuffled — every workgroup gets a uniform sample of it (rounds 1-2 dealt contiguous
// chunks: bounce rays of a frame in pixel order ran 7.8 Gray/s where the same rays dealt this way run 10.1,
// profiles/r03_coherence_probe.txt), while a wave still holds 64 consecutive rays. Inside a workgroup the waves pull
// their rays from an LDS counter as their lanes fall idle.
struct Chunk { size_t count; };
__device__ __forceinline__ Chunk my_chunk(size_t n) {
  const size_t blocks = (n + kBlock - 1) / kBlock, G = gridDim.x, b = blockIdx.x;
  if (b >= blocks) return Chunk{0};
  const size_t mine = (blocks - b + G - 1) / G;  // blocks b, b + G, ... < blocks
  size_t count = mine * kBlock;
  if ((blocks - 1) % G == b) count -= blocks * kBlock - n;  // the batch's last, ragged block is this workgroup's last
  return Chunk{count};
}
__device__ __forceinline__ size_t dealt_ray(uint32_t k) {
  return ((size_t)(k / kBlock) * gridDim.x + blockIdx.x) * kBlock + (k % kBlock);
}
__device__ __forceinline__ bool lds_take(bool want, uint32_t *next, uint32_t limit, uint32_t &idx) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0) return false;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(next, (uint32_t)__popcll(mask));
  base = __shfl(base, leader, 64);
  idx = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  return want && idx < limit;
}

// WIDE: the four-workgroups-per-CU split of the engine (traverse_pool.hip.h), launched for flat scenes.
template <bool STATS, bool WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? 4 : 3) void intersect_n_kernel(DevScene S, const CrtRay *__restrict__ rays, size_t n,
                                                            float t_min, float t_max, CrtRayHit *__restrict__ hits,
                                                            uint32_t *__restrict__ err_out, CrtTravStats *stats) {
  __shared__ __attribute__((aligned(16))) uint32_t engine_lds[WIDE ? kEngineLdsWide : kEngineLdsDwords];
  __shared__ uint32_t next;
  const Chunk ck = my_chunk(n);
  if (ck.count == 0) return;
  if (threadIdx.x == 0) next = 0;
  LaneStats st = {};
  uint32_t err = 0, done = 0;
  auto fetch = [&](bool want, RayIn &in) -> bool {
    uint32_t k;
    if (!lds_take(want, &next, (uint32_t)ck.count, k)) return false;
    const float4 *rp = reinterpret_cast<const float4 *>(rays + dealt_ray(k));
    const float4 o = rp[0], d = rp[1], tm = rp[2];
    in.ox = o.x; in.oy = o.y; in.oz = o.z; in.dx = d.x; in.dy = d.y; in.dz = d.z;
    in.time = tm.x; in.mask = __float_as_uint(tm.y); in.t_min = t_min; in.t_max = t_max; in.slot = k;
    return true;
  };
  auto emit = [&](uint32_t k, bool hit, const Hit &h, float dx, float dy, float dz) {
    const size_t i = dealt_ray(k);
    CrtRayHit out;
    if (hit) {  // scene.rs:355-365
      const bool front = dot3(dx, dy, dz, h.nx, h.ny, h.nz) < 0.0f;
      out.t = h.t;
      out.normal[0] = front ? h.nx : -h.nx;
      out.normal[1] = front ? h.ny : -h.ny;
      out.normal[2] = front ? h.nz : -h.nz;
      out.front_face = front ? 1u : 0u;
      out.u = h.u; out.v = h.v; out.geom_id = h.geom; out.prim_id = h.prim; out._pad = 0;
    } else {
      out.t = 0.0f; out.normal[0] = out.normal[1] = out.normal[2] = 0.0f; out.front_face = 0;
      out.u = out.v = 0.0f; out.geom_id = CRT_INVALID_ID; out.prim_id = CRT_INVALID_ID; out._pad = 0;
    }
    hits[i] = out;
    done++;
  };
  run_traversal<false, STATS, WIDE>(S, engine_lds, t_min, err, st, fetch, emit);
  if (err) atomicOr(err_out, err);
  if (STATS) flush_stats(st, stats, done);
}

template <bool STATS, bool WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? 4 : 3) void occluded_n_kernel(DevScene S, const CrtRay *__restrict__ rays, size_t n,
                                                           float t_min, float t_max, uint32_t *__restrict__ out,
                                                           uint32_t *__restrict__ err_out, CrtTravStats *stats) {
  __shared__ __attribute__((aligned(16))) uint32_t engine_lds[WIDE ? kEngineLdsWide : kEngineLdsDwords];
  __shared__ uint32_t next;
  const Chunk ck = my_chunk(n);
  if (ck.count == 0) return;
  if (threadIdx.x == 0) next = 0;
  LaneStats st = {};
  uint32_t err = 0, done = 0;
  auto fetch = [&](bool want, RayIn &in) -> bool {
    uint32_t k;
    if (!lds_take(want, &next, (uint32_t)ck.count, k)) return false;
    const float4 *rp = reinterpret_cast<const float4 *>(rays + dealt_ray(k));
    const float4 o = rp[0], d = rp[1], tm = rp[2];
    in.ox = o.x; in.oy = o.y; in.oz = o.z; in.dx = d.x; in.dy = d.y; in.dz = d.z;
    in.time = tm.x; in.mask = __float_as_uint(tm.y); in.t_min = t_min; in.t_max = t_max; in.slot = k;
    return true;
  };
  auto emit = [&](uint32_t k, bool occ, const Hit &, float, float, float) {
    out[dealt_ray(k)] = occ ? 1u : 0u;
    done++;
  };
  run_traversal<true, STATS, WIDE>(S, engine_lds, t_min, err, st, fetch, emit);
  if (err) atomicOr(err_out, err);
  if (STATS) flush_stats(st, stats, done);
}

// Small flat triangle scenes run the WIDE kernels (four workgroups resident per CU), the others the scene's own split
// on three: crt_internal.h, select_engine — the one place that decides, from the image, which engine instance runs it.
// The batched queries report u and v and keep every cold field (kColdAll).
int query_engine(const DevScene &s, EngineSelect &e) {
  const int rc = select_engine_env(s, e);
  if (rc != CRT_OK) return rc;
  if (!engine_accepts(e, s, (int)kColdAll)) {
    set_error_text("traversal launch refused: the selected engine instance (wide %d, direct %d) cannot decode this image (direct words %u)",
                   (int)e.wide, (int)e.direct, s.direct_leaves);
    return CRT_ERR_UNSUPPORTED;
  }
  return CRT_OK;
}
int grid_for(size_t n, bool wide) {
  static int cus = [] {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!CRT_HIP_OK(hipGetDeviceProperties(&prop, dev))) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  size_t need = (n + kBlock - 1) / kBlock;
  size_t cap = (size_t)cus * (wide ? 4 : 3);  // what stays resident per CU (LDS of the traversal engine): one chunk per workgroup
  return (int)(need < cap ? (need ? need : 1) : cap);
}

}  // namespace

int launch_intersect_n(const DevScene &s, const CrtRay *d_rays, size_t n, float t_min, float t_max, CrtRayHit *d_hits,
                       void *stream, CrtTravStats *d_stats, uint32_t *e) {
  if (n == 0) return CRT_OK;
  if (!e) return CRT_ERR_BAD_ARG;
  EngineSelect eng;
  if (const int rc = query_engine(s, eng)) return rc;
  const bool wide = eng.wide;
  const int grid = grid_for(n, wide);
#define CRT_LAUNCH(ST, W)                                                                                             \
  hipLaunchKernelGGL((intersect_n_kernel<ST, W>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, s, d_rays, n, t_min, \
                     t_max, d_hits, e, d_stats)
  if (d_stats) { if (wide) CRT_LAUNCH(true, true); else CRT_LAUNCH(true, false); }
  else { if (wide) CRT_LAUNCH(false, true); else CRT_LAUNCH(false, false); }
#undef CRT_LAUNCH
  return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : CRT_ERR_NO_DEVICE;
}

int launch_occluded_n(const DevScene &s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                      void *stream, CrtTravStats *d_stats, uint32_t *e) {
  if (n == 0) return CRT_OK;
  if (!e) return CRT_ERR_BAD_ARG;
  EngineSelect eng;
  if (const int rc = query_engine(s, eng)) return rc;
  const bool wide = eng.wide;
  const int grid = grid_for(n, wide);
#define CRT_LAUNCH(ST, W)                                                                                            \
  hipLaunchKernelGGL((occluded_n_kernel<ST, W>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, s, d_rays, n, t_min, \
                     t_max, d_out, e, d_stats)
  if (d_stats) { if (wide) CRT_LAUNCH(true, true); else CRT_LAUNCH(true, false); }
  else { if (wide) CRT_LAUNCH(false, true); else CRT_LAUNCH(false, false); }
#undef CRT_LAUNCH
  return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : CRT_ERR_NO_DEVICE;
}

}  // namespace crt
