// The shading device functions of kernels/shade.hip.h — OpenPBR sample / eval (general instance: every lobe, thin film,
// dispersion, media), emission, every light kind's sampling incl. the lights at infinity, escaped-ray lookup, camera,
// filter, interior media, Henyey-Greenstein — compiled as host C++ behind profiles/host_shade/hip/hip_runtime.h and run
// on random inputs under MemorySanitizer (a use of an uninitialised value anywhere in a result aborts the run) and
// UndefinedBehaviorSanitizer. VERDICT r2, next-round item 1(b): "run the vertex step on the CPU under sanitizers".
//   bash profiles/host_shade/run.sh
#include <cstdio>
#include <cstdlib>
#include <random>
#include "shade.hip.h"
using namespace crt;
using namespace crt::dev;

static std::mt19937 rng(12345);
static float U() { return std::generate_canonical<float, 24>(rng); }
static float R(float a, float b) { return a + (b - a) * U(); }
static V3 rv(float a, float b) { return v3(R(a, b), R(a, b), R(a, b)); }

static CrtMaterial random_material(int k) {
  CrtMaterial m;
  std::memset(&m, 0, sizeof m);  // the host always uploads fully written records (ctypes arrays are zero-initialised)
  float *f = reinterpret_cast<float *>(&m);
  for (size_t i = 0; i < sizeof(CrtMaterial) / 4; i++) f[i] = U();  // every float field in [0, 1)
  m.kind = (k % 7 == 0) ? CRT_MAT_EMISSIVE : CRT_MAT_OPENPBR;
  // weights that exercise each arm in turn
  m.coat_weight = (k & 1) ? U() : 0.0f;
  m.fuzz_weight = (k & 2) ? U() : 0.0f;
  m.thin_film_weight = (k & 4) ? U() : 0.0f;
  m.transmission_weight = (k & 8) ? U() : 0.0f;
  m.subsurface_weight = (k & 16) ? U() : 0.0f;
  m.specular_ior = R(1.05f, 2.5f);
  m.coat_ior = R(1.1f, 2.0f);
  m.thin_film_ior = R(1.1f, 2.0f);
  m.thin_film_thickness = R(0.05f, 1.5f);
  m.transmission_depth = (k & 32) ? R(0.0f, 2.0f) : 0.0f;
  m.transmission_dispersion_scale = (k & 64) ? U() : 0.0f;
  m.transmission_dispersion_abbe_number = R(10.0f, 60.0f);
  m.thin_walled = (k & 128) ? 1u : 0u;
  return m;
}

int main() {
  static uint32_t tab[kSobolLdsWords];
  sobol_tables_init(tab);
  double sum = 0.0;
  unsigned long long n_scatter = 0, n_eval = 0, n_light = 0, n_esc = 0;
  for (int it = 0; it < 200000; it++) {
    const CrtMaterial m = random_material(it);
    HitRec rec;
    rec.normal = normalize(rv(-1.0f, 1.0f));
    rec.p = rv(-5.0f, 5.0f);
    rec.t = R(0.01f, 20.0f);
    rec.front_face = (it & 1) != 0;
    V3 rd = rv(-1.0f, 1.0f);
    if (dot(rd, rec.normal) > 0.0f) rd = rd * -1.0f;  // the hit record's normal faces the ray
    const Sampler dom = new_domain(sampler_new(it & 255, (it >> 8) & 255, 0, it), it % 13);
    // emission
    const V3 em = mat_emitted_directional<false>(m, fabs_(dot(normalize(rd), rec.normal)));
    sum += em.x + em.y + em.z;
    // interior medium of the material
    DevMedium med;
    medium_from_material(m, med);
    if (med.present) {
      const V3 tr = medium_transmittance(med, rec.t), ch = medium_chromatic(med, rec.t);
      const V3 hg = sample_henyey_greenstein(normalize(rd), med.g, U(), U());
      sum += tr.x + tr.y + tr.z + ch.x + ch.y + ch.z + hg.x + hg.y + hg.z + med.sigma_bar;
    }
    if (m.kind != CRT_MAT_OPENPBR) continue;
    // BSDF sample + eval toward a light direction
    Scatter sc;
    if (mat_scatter<false>(m, rd, rec, dom, sc, tab)) {
      n_scatter++;
      sum += sc.value.x + sc.value.y + sc.value.z + sc.pdf + sc.dir.x + sc.dir.y + sc.dir.z + sc.origin.x + sc.origin.y + sc.origin.z +
             (sc.delta ? 1.0 : 0.0) + (sc.medium ? 1.0 : 0.0);
    }
    const V3 wi = normalize(rv(-1.0f, 1.0f));
    V3 val; float pdf;
    if (mat_eval<false>(m, rd, rec, wi, val, pdf)) { n_eval++; sum += val.x + val.y + val.z + pdf; }
    // every light kind
    CrtLight l;
    std::memset(&l, 0, sizeof l);
    float *lf = reinterpret_cast<float *>(&l);
    for (size_t i = 0; i < sizeof(CrtLight) / 4; i++) lf[i] = R(-1.0f, 1.0f);
    l.kind = (uint32_t)(it % 4);  // sphere, rect, distant, dome
    l.geom_id = 3;
    LightSample ls;
    if (light_sample_li<true>(l, rec.p, U(), U(), ls)) {
      n_light++;
      sum += ls.direction.x + ls.direction.y + ls.direction.z + ls.distance * 1e-30 + ls.pdf * 1e-6 + ls.radiance.x + ls.radiance.y + ls.radiance.z;
      sum += solid_angle_pdf(l, rec.p, rec.p + ls.direction) * 1e-6;
    }
    V3 er; float ep;
    if (light_escaped(l, normalize(rd), er, ep)) { n_esc++; sum += er.x + er.y + er.z + ep * 1e-6; }
    // camera + filter
    CrtCamera cam;
    std::memset(&cam, 0, sizeof cam);
    float *cf = reinterpret_cast<float *>(&cam);
    for (size_t i = 0; i < sizeof(CrtCamera) / 4; i++) cf[i] = R(-1.0f, 1.0f);
    cam.lens_radius = (it & 3) ? 0.0f : 0.05f;
    V3 o, d;
    camera_get_ray(cam, filter_offset(it & 1, 1.0f, U()), filter_offset((it >> 1) & 1, 1.0f, U()), U(), U(), o, d);
    sum += o.x + o.y + o.z + d.x + d.y + d.z;
  }
  // every value above flowed into `sum`: printing it makes MemorySanitizer check the whole data flow
  if (sum != sum) std::printf("checksum is NaN (some inputs are degenerate by construction)\n");
  std::printf("scatter %llu eval %llu light samples %llu escaped %llu checksum %.17g\n", n_scatter, n_eval, n_light, n_esc, sum);
  return 0;
}
