// CPU twin of kernels/shade_seam.hip (tests only): the SAME shading device functions (kernels/shade.hip.h, dmath.hip.h,
// qmc.hip.h) compiled as host C++ behind the HIP stand-in header profiles/host_shade/hip/hip_runtime.h, exported with
// the oracle drivers' signatures (oracle/ora_shade.c: ora_t_scatter_n ...) so tests/test_shading_seam_host.py can compare
// the two record by record — the function-level parity check of SURVEY §8 a26-a33 that runs without a GPU.
// Build: tests/test_shading_seam_host.py (g++ -O1 -ffp-contract=off -shared).
#include <cstddef>
#include <cstring>
#include "shade.hip.h"

using namespace crt;
using namespace crt::dev;

static uint32_t g_tab[kSobolLdsWords];
static bool g_tab_ready = false;
static const uint32_t *sobol_tab() {
  if (!g_tab_ready) { sobol_tables_init(g_tab); g_tab_ready = true; }
  return g_tab;
}
static HitRec rec_of(const CrtShadeQuery &q) {
  HitRec r;
  r.p = v3(q.p[0], q.p[1], q.p[2]); r.normal = v3(q.normal[0], q.normal[1], q.normal[2]); r.t = q.t;
  r.front_face = q.front_face != 0;
  return r;
}
static void put(float d[3], V3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; }

extern "C" {

void host_scatter_n(const CrtMaterial *mats, size_t n_mats, const CrtShadeQuery *qs, size_t n, CrtScatterSample *out) {
  const uint32_t *tab = sobol_tab();
  for (size_t i = 0; i < n; i++) {
    std::memset(&out[i], 0, sizeof out[i]);
    if (qs[i].material >= n_mats) continue;
    const CrtMaterial &m = mats[qs[i].material];
    Scatter sc;
    if (!mat_scatter<false>(m, v3(qs[i].ray_dir[0], qs[i].ray_dir[1], qs[i].ray_dir[2]), rec_of(qs[i]),
                            Sampler{qs[i].sampler_pattern, qs[i].sampler_index}, sc, tab)) continue;
    if (sc.medium) { DevMedium med; medium_from_material(m, med); sc.medium = med.present != 0; }  // as k_seam_scatter
    put(out[i].origin, sc.origin); put(out[i].dir, sc.dir); put(out[i].value, sc.value);
    out[i].some = 1; out[i].pdf = sc.pdf; out[i].flags = (sc.delta ? 1u : 0u) | (sc.medium ? 2u : 0u);
  }
}
void host_eval_n(const CrtMaterial *mats, size_t n_mats, const CrtShadeQuery *qs, size_t n, CrtBsdfEval *out) {
  for (size_t i = 0; i < n; i++) {
    std::memset(&out[i], 0, sizeof out[i]);
    if (qs[i].material >= n_mats) continue;
    V3 value; float pdf;
    if (!mat_eval<false>(mats[qs[i].material], v3(qs[i].ray_dir[0], qs[i].ray_dir[1], qs[i].ray_dir[2]), rec_of(qs[i]),
                         v3(qs[i].wi[0], qs[i].wi[1], qs[i].wi[2]), value, pdf)) continue;
    put(out[i].value, value); out[i].pdf = pdf; out[i].some = 1;
  }
}
void host_emitted_n(const CrtMaterial *mats, size_t n_mats, const CrtShadeQuery *qs, size_t n, float *rgb) {
  for (size_t i = 0; i < n; i++) {
    V3 e = splat(0.0f);
    if (qs[i].material < n_mats) e = mat_emitted_directional<false>(mats[qs[i].material], qs[i].cos_theta_o);
    put(rgb + 3 * i, e);
  }
}
void host_light_sample_n(const CrtLight *ls, size_t n_ls, const CrtLightQuery *qs, size_t n, CrtLightSample *out) {
  for (size_t i = 0; i < n; i++) {
    std::memset(&out[i], 0, sizeof out[i]);
    LightSample s;
    if (qs[i].light >= n_ls || !light_sample_li<true>(ls[qs[i].light], v3(qs[i].from[0], qs[i].from[1], qs[i].from[2]), qs[i].u, qs[i].v, s)) continue;
    put(out[i].direction, s.direction); out[i].distance = s.distance; put(out[i].radiance, s.radiance);
    out[i].pdf = s.pdf; out[i].some = 1;
  }
}
void host_light_pdf_n(const CrtLight *ls, size_t n_ls, const CrtLightQuery *qs, size_t n, float *pdf) {
  for (size_t i = 0; i < n; i++) {
    pdf[i] = 0.0f;
    if (qs[i].light < n_ls && ls[qs[i].light].kind <= CRT_LIGHT_RECT)
      pdf[i] = solid_angle_pdf(ls[qs[i].light], v3(qs[i].from[0], qs[i].from[1], qs[i].from[2]), v3(qs[i].point[0], qs[i].point[1], qs[i].point[2]));
  }
}
void host_light_escaped_n(const CrtLight *ls, size_t n_ls, const CrtLightQuery *qs, size_t n, CrtLightSample *out) {
  for (size_t i = 0; i < n; i++) {
    std::memset(&out[i], 0, sizeof out[i]);
    V3 rad; float pdf;
    const V3 d = v3(qs[i].point[0], qs[i].point[1], qs[i].point[2]);
    if (qs[i].light >= n_ls || !light_escaped(ls[qs[i].light], d, rad, pdf)) continue;
    put(out[i].direction, d); out[i].distance = CRT_INF; put(out[i].radiance, rad); out[i].pdf = pdf; out[i].some = 1;
  }
}

}  // extern "C"
