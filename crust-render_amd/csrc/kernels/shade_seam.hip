// The shading seam as callable functions: `trait Material` (material.rs:26-116) and `trait Light` (light.rs:120-151)
// over batches of queries, for a host integrator that keeps its own trace_path on top of crt_intersect_n /
// crt_occluded_n ("the unmodified per-pixel integrator").
//
// Thin kernels over the device functions the wavefront integrator itself runs (shade.hip.h, general instance: every
// lobe, thin film, dispersion, thin walls, lights at infinity) — one query per lane, records read and written as
// 16-byte vectors. Nothing here is on the renderer's hot path; what matters is that it is the SAME arithmetic, so a
// function-level comparison with the oracle (tests/test_gpu_shading_seam.py) pins what the renders only show in sum.
// Compile with -ffp-contract=off (dmath.hip.h).
#include "shade.hip.h"
#include "../crt_internal.h"

namespace crt {

using namespace dev;

namespace {

constexpr int kSeamBlock = 256;

struct Query { V3 rd; uint32_t material; HitRec rec; V3 wi; float cos_o; Sampler dom; };
static_assert(sizeof(CrtShadeQuery) == 80 && sizeof(CrtScatterSample) == 48 && sizeof(CrtBsdfEval) == 32, "crt.h record sizes");
static_assert(sizeof(CrtLightQuery) == 48 && sizeof(CrtLightSample) == 48, "crt.h record sizes");

__device__ __forceinline__ Query load_query(const CrtShadeQuery *q) {
  const float4 *w = reinterpret_cast<const float4 *>(q);
  const float4 a = w[0], b = w[1], c = w[2], d = w[3];
  const uint4 e = reinterpret_cast<const uint4 *>(q)[4];
  Query o;
  o.rd = v3(a.x, a.y, a.z); o.material = __float_as_uint(a.w);
  o.rec.p = v3(b.x, b.y, b.z); o.rec.t = b.w;
  o.rec.normal = v3(c.x, c.y, c.z); o.rec.front_face = __float_as_uint(c.w) != 0;
  o.wi = v3(d.x, d.y, d.z); o.cos_o = d.w;
  o.dom = Sampler{e.x, e.y};
  return o;
}

// Material::scatter_importance (material.rs:40-45)
__global__ __launch_bounds__(kSeamBlock) void k_seam_scatter(const CrtMaterial *mats, uint32_t n_mats, const CrtShadeQuery *qs,
                                                             size_t n, CrtScatterSample *out) {
  __shared__ uint32_t sobol_tab[kSobolLdsWords];
  sobol_tables_init(sobol_tab);  // ends with a barrier
  const size_t i = (size_t)blockIdx.x * kSeamBlock + threadIdx.x;
  if (i >= n) return;
  const Query q = load_query(qs + i);
  Scatter sc;
  sc.origin = sc.dir = sc.value = splat(0.0f); sc.pdf = 0.0f; sc.delta = false; sc.medium = false;
  bool some = false;
  if (q.material < n_mats) {
    const CrtMaterial &m = mats[q.material];
    some = mat_scatter<false>(m, q.rd, q.rec, q.dom, sc, sobol_tab);
    // the ray carries the interior only if the material HAS one (openpbr.rs:1061-1066: `if let Some(medium)`)
    if (some && sc.medium) {
      DevMedium med;
      medium_from_material(m, med);
      sc.medium = med.present != 0;
    }
  }
  if (!some) { sc.origin = sc.dir = sc.value = splat(0.0f); sc.pdf = 0.0f; sc.delta = false; sc.medium = false; }
  float4 *w = reinterpret_cast<float4 *>(out + i);
  w[0] = make_float4(sc.origin.x, sc.origin.y, sc.origin.z, __uint_as_float(some ? 1u : 0u));
  w[1] = make_float4(sc.dir.x, sc.dir.y, sc.dir.z, sc.pdf);
  w[2] = make_float4(sc.value.x, sc.value.y, sc.value.z, __uint_as_float((sc.delta ? 1u : 0u) | (sc.medium ? 2u : 0u)));
}

// Material::eval (material.rs:56-74)
__global__ __launch_bounds__(kSeamBlock) void k_seam_eval(const CrtMaterial *mats, uint32_t n_mats, const CrtShadeQuery *qs, size_t n,
                                                          CrtBsdfEval *out) {
  const size_t i = (size_t)blockIdx.x * kSeamBlock + threadIdx.x;
  if (i >= n) return;
  const Query q = load_query(qs + i);
  V3 value = splat(0.0f);
  float pdf = 0.0f;
  bool some = false;
  if (q.material < n_mats) some = mat_eval<false>(mats[q.material], q.rd, q.rec, q.wi, value, pdf);
  if (!some) { value = splat(0.0f); pdf = 0.0f; }
  float4 *w = reinterpret_cast<float4 *>(out + i);
  w[0] = make_float4(value.x, value.y, value.z, pdf);
  w[1] = make_float4(__uint_as_float(some ? 1u : 0u), 0.0f, 0.0f, 0.0f);
}

// Material::emitted_directional (material.rs:112-115)
__global__ __launch_bounds__(kSeamBlock) void k_seam_emitted(const CrtMaterial *mats, uint32_t n_mats, const CrtShadeQuery *qs,
                                                             size_t n, float *rgb) {
  const size_t i = (size_t)blockIdx.x * kSeamBlock + threadIdx.x;
  if (i >= n) return;
  const Query q = load_query(qs + i);
  V3 e = splat(0.0f);
  if (q.material < n_mats) e = mat_emitted_directional<false>(mats[q.material], q.cos_o);
  rgb[3 * i] = e.x; rgb[3 * i + 1] = e.y; rgb[3 * i + 2] = e.z;
}

struct LQuery { V3 from; uint32_t light; float u, v; V3 point; };
__device__ __forceinline__ LQuery load_lquery(const CrtLightQuery *q) {
  const float4 *w = reinterpret_cast<const float4 *>(q);
  const float4 a = w[0], b = w[1], c = w[2];
  LQuery o;
  o.from = v3(a.x, a.y, a.z); o.light = __float_as_uint(a.w);
  o.u = b.x; o.v = b.y;
  o.point = v3(c.x, c.y, c.z);
  return o;
}
__device__ __forceinline__ void store_lsample(CrtLightSample *out, bool some, V3 dir, float dist, V3 rad, float pdf) {
  float4 *w = reinterpret_cast<float4 *>(out);
  if (!some) { dir = rad = splat(0.0f); dist = 0.0f; pdf = 0.0f; }
  w[0] = make_float4(dir.x, dir.y, dir.z, dist);
  w[1] = make_float4(rad.x, rad.y, rad.z, pdf);
  w[2] = make_float4(__uint_as_float(some ? 1u : 0u), 0.0f, 0.0f, 0.0f);
}

// mode 0: Light::sample_li (light.rs:126) | 1: Light::pdf_at_point (:132) | 2: Light::escaped (:141)
template <int MODE>
__global__ __launch_bounds__(kSeamBlock) void k_seam_light(const CrtLight *lights, uint32_t n_lights, const CrtLightQuery *qs, size_t n,
                                                           CrtLightSample *out, float *pdf_out) {
  const size_t i = (size_t)blockIdx.x * kSeamBlock + threadIdx.x;
  if (i >= n) return;
  const LQuery q = load_lquery(qs + i);
  const bool ok = q.light < n_lights;
  if (MODE == 0) {
    LightSample ls;
    ls.direction = ls.radiance = splat(0.0f); ls.distance = 0.0f; ls.pdf = 0.0f;
    const bool some = ok && light_sample_li<true>(lights[q.light], q.from, q.u, q.v, ls);
    store_lsample(out + i, some, ls.direction, ls.distance, ls.radiance, ls.pdf);
  } else if (MODE == 1) {
    float pdf = 0.0f;  // the trait's default for lights at infinity (light.rs:132-134)
    if (ok && lights[q.light].kind <= CRT_LIGHT_RECT) pdf = solid_angle_pdf(lights[q.light], q.from, q.point);
    pdf_out[i] = pdf;
  } else {
    V3 rad = splat(0.0f);
    float pdf = 0.0f;
    const bool some = ok && light_escaped(lights[q.light], q.point, rad, pdf);
    store_lsample(out + i, some, q.point, CRT_INF, rad, pdf);
  }
}

int seam_args(const void *table, size_t n_table, const void *queries, size_t n, const void *out) {
  if (n == 0) return 1;  // nothing to do
  if (!queries || !out || (n_table && !table) || n_table > 0xffffffffull) return CRT_ERR_BAD_ARG;
  if (!device_ok()) return CRT_ERR_NO_DEVICE;
  return CRT_OK;
}
unsigned seam_grid(size_t n) { return (unsigned)((n + kSeamBlock - 1) / kSeamBlock); }
int seam_done() { return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : CRT_ERR_NO_DEVICE; }

}  // namespace
}  // namespace crt

using namespace crt;

extern "C" {

int crt_material_scatter_n(const CrtMaterial *d_materials, size_t n_materials, const CrtShadeQuery *d_queries, size_t n,
                           CrtScatterSample *d_out, void *stream) {
  const int rc = seam_args(d_materials, n_materials, d_queries, n, d_out);
  if (rc != CRT_OK) return rc > 0 ? CRT_OK : rc;
  hipLaunchKernelGGL(k_seam_scatter, dim3(seam_grid(n)), dim3(kSeamBlock), 0, (hipStream_t)stream, d_materials, (uint32_t)n_materials,
                     d_queries, n, d_out);
  return seam_done();
}
int crt_material_eval_n(const CrtMaterial *d_materials, size_t n_materials, const CrtShadeQuery *d_queries, size_t n,
                        CrtBsdfEval *d_out, void *stream) {
  const int rc = seam_args(d_materials, n_materials, d_queries, n, d_out);
  if (rc != CRT_OK) return rc > 0 ? CRT_OK : rc;
  hipLaunchKernelGGL(k_seam_eval, dim3(seam_grid(n)), dim3(kSeamBlock), 0, (hipStream_t)stream, d_materials, (uint32_t)n_materials,
                     d_queries, n, d_out);
  return seam_done();
}
int crt_material_emitted_n(const CrtMaterial *d_materials, size_t n_materials, const CrtShadeQuery *d_queries, size_t n,
                           float *d_rgb, void *stream) {
  const int rc = seam_args(d_materials, n_materials, d_queries, n, d_rgb);
  if (rc != CRT_OK) return rc > 0 ? CRT_OK : rc;
  hipLaunchKernelGGL(k_seam_emitted, dim3(seam_grid(n)), dim3(kSeamBlock), 0, (hipStream_t)stream, d_materials, (uint32_t)n_materials,
                     d_queries, n, d_rgb);
  return seam_done();
}
int crt_light_sample_n(const CrtLight *d_lights, size_t n_lights, const CrtLightQuery *d_queries, size_t n,
                       CrtLightSample *d_out, void *stream) {
  const int rc = seam_args(d_lights, n_lights, d_queries, n, d_out);
  if (rc != CRT_OK) return rc > 0 ? CRT_OK : rc;
  hipLaunchKernelGGL((k_seam_light<0>), dim3(seam_grid(n)), dim3(kSeamBlock), 0, (hipStream_t)stream, d_lights, (uint32_t)n_lights,
                     d_queries, n, d_out, (float *)nullptr);
  return seam_done();
}
int crt_light_pdf_n(const CrtLight *d_lights, size_t n_lights, const CrtLightQuery *d_queries, size_t n, float *d_pdf,
                    void *stream) {
  const int rc = seam_args(d_lights, n_lights, d_queries, n, d_pdf);
  if (rc != CRT_OK) return rc > 0 ? CRT_OK : rc;
  hipLaunchKernelGGL((k_seam_light<1>), dim3(seam_grid(n)), dim3(kSeamBlock), 0, (hipStream_t)stream, d_lights, (uint32_t)n_lights,
                     d_queries, n, (CrtLightSample *)nullptr, d_pdf);
  return seam_done();
}
int crt_light_escaped_n(const CrtLight *d_lights, size_t n_lights, const CrtLightQuery *d_queries, size_t n,
                        CrtLightSample *d_out, void *stream) {
  const int rc = seam_args(d_lights, n_lights, d_queries, n, d_out);
  if (rc != CRT_OK) return rc > 0 ? CRT_OK : rc;
  hipLaunchKernelGGL((k_seam_light<2>), dim3(seam_grid(n)), dim3(kSeamBlock), 0, (hipStream_t)stream, d_lights, (uint32_t)n_lights,
                     d_queries, n, d_out, (float *)nullptr);
  return seam_done();
}

}  // extern "C"
