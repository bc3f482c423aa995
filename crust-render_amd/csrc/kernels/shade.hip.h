// Per-hit shading on the device: the OpenPBR übershader, Emissive, area lights, camera and pixel filter.
//
// What it computes (operation order as written in the reference):
//   material/brdf.rs:10-421, material/openpbr.rs:317-1218, material/emissive.rs:25-38,
//   light.rs:27-81,:180-213, camera.rs:71-84, filter.rs:182-205, utils/src/common.rs:38-49,:128-174.
// The Material trait (material.rs:26-116) is closed over two implementations, so the device dispatches on
// CrtMaterial::kind instead of a vtable.
#pragma once

#include "dmath.hip.h"
#include "qmc.hip.h"
#include "../../../include/crt.h"

namespace crt {
namespace dev {

struct HitRec { V3 p, normal; float t; bool front_face; };  // hittable.rs:10-36
struct Scatter { V3 origin, dir, value; float pdf; bool delta; bool medium; };  // material.rs:8-22; medium: the ray enters the interior

__device__ __forceinline__ V3 ld3(const float c[3]) { return v3(c[0], c[1], c[2]); }

// ---- utils/src/common.rs ----
__device__ __forceinline__ float balance_heuristic(float a, float b) { return a / (a + b + 1e-6f); }
__device__ __forceinline__ float power_heuristic(float a, float b) {
  const float a2 = a * a, b2 = b * b;
  return a2 / (a2 + b2 + 1e-6f);
}
__device__ __forceinline__ V3 cosine_hemisphere(float u, float v) {
  const float z = sqrtf(1.0f - v);
  const float phi = 2.0f * CRT_PI * u;
  float s, c;
  sincos_det(phi, s, c);
  return v3(c * sqrtf(v), s * sqrtf(v), z);
}
__device__ __forceinline__ V3 concentric_disk(float u, float v) {
  const float sx = 2.0f * u - 1.0f, sy = 2.0f * v - 1.0f;
  if (sx == 0.0f && sy == 0.0f) return splat(0.0f);
  const float FRAC_PI_4 = 0.785398163397448309615660845819875721f, FRAC_PI_2 = 1.57079632679489661923132169163975144f;
  float r, theta;
  if (fabs_(sx) > fabs_(sy)) { r = sx; theta = FRAC_PI_4 * (sy / sx); }
  else { r = sy; theta = FRAC_PI_2 - FRAC_PI_4 * (sx / sy); }
  float s, c;
  sincos_det(theta, s, c);
  return v3(r * c, r * s, 0.0f);
}

// ---- material/brdf.rs ----
__device__ __forceinline__ V3 fresnel_schlick(float cos_theta, V3 f0) {
  return f0 + (splat(1.0f) - f0) * pow5_(1.0f - cos_theta);
}
__device__ __forceinline__ float fresnel_schlick_scalar(float cos_theta, float f0) {
  return f0 + (1.0f - f0) * pow5_(1.0f - cos_theta);
}
__device__ __forceinline__ float f0_from_ior(float ior) { const float r = (ior - 1.0f) / (ior + 1.0f); return r * r; }
__device__ __forceinline__ void roughness_to_alpha(float roughness, float anisotropy, float &ax, float &ay) {
  const float a = roughness * roughness;
  const float inv = 1.0f - rclamp(anisotropy, 0.0f, 1.0f);
  const float x = a * sqrtf(2.0f / (1.0f + inv * inv));
  const float y = inv * x;
  ax = rmax(x, 1e-4f);
  ay = rmax(y, 1e-4f);
}
__device__ __forceinline__ float ggx_d(float n_dot_h, float h_dot_t, float h_dot_b, float ax, float ay) {
  const float tx = h_dot_t / ax, ty = h_dot_b / ay;
  const float term = tx * tx + ty * ty + n_dot_h * n_dot_h;
  return 1.0f / (CRT_PI * ax * ay * term * term);
}
__device__ __forceinline__ float ggx_lambda(float v_dot_n, float v_dot_t, float v_dot_b, float ax, float ay) {
  const float vt = v_dot_t * ax, vb = v_dot_b * ay;
  const float a2 = vt * vt + vb * vb;
  const float n2 = rmax(v_dot_n * v_dot_n, 1e-8f);
  return (-1.0f + sqrtf(1.0f + a2 / n2)) * 0.5f;
}
__device__ __forceinline__ float ggx_g2(float vn, float vt, float vb, float ln, float lt, float lb, float ax, float ay) {
  const float lv = ggx_lambda(vn, vt, vb, ax, ay);
  const float ll = ggx_lambda(ln, lt, lb, ax, ay);
  return 1.0f / (1.0f + lv + ll);
}
__device__ __forceinline__ V3 sample_vndf(V3 v_local, float ax, float ay, float u1, float u2) {  // Heitz 2018
  const V3 vh = normalize(v3(ax * v_local.x, ay * v_local.y, v_local.z));
  const float lensq = vh.x * vh.x + vh.y * vh.y;
  const V3 t1 = lensq > 0.0f ? v3(-vh.y, vh.x, 0.0f) / sqrtf(lensq) : v3(1.0f, 0.0f, 0.0f);
  const V3 t2 = cross(vh, t1);
  const float r = sqrtf(u1);
  const float phi = 2.0f * CRT_PI * u2;
  float sp, cp;
  sincos_det(phi, sp, cp);
  const float t1c = r * cp;
  const float t2c_pre = r * sp;
  const float s = 0.5f * (1.0f + vh.z);
  const float t2c = (1.0f - s) * sqrtf(rmax(1.0f - t1c * t1c, 0.0f)) + s * t2c_pre;
  const V3 nh = t1 * t1c + t2 * t2c + vh * sqrtf(rmax(1.0f - t1c * t1c - t2c * t2c, 0.0f));
  return normalize(v3(ax * nh.x, ay * nh.y, rmax(nh.z, 0.0f)));
}
__device__ __forceinline__ float pdf_vndf(V3 v_local, V3 h_local, float ax, float ay) {
  const float n_dot_v = rmax(v_local.z, 1e-6f);
  const float n_dot_h = rmax(h_local.z, 1e-6f);
  const float d = ggx_d(n_dot_h, h_local.x, h_local.y, ax, ay);
  const float lambda_v = ggx_lambda(n_dot_v, v_local.x, v_local.y, ax, ay);
  const float g1 = 1.0f / (1.0f + lambda_v);
  return d * g1 / (4.0f * n_dot_v);
}
__device__ __forceinline__ float pdf_vndf_h(V3 v_local, V3 h_local, float ax, float ay) {
  const float n_dot_v = rmax(v_local.z, 1e-6f);
  const float v_dot_h = rmax(dot(v_local, h_local), 0.0f);
  const float d = ggx_d(rmax(h_local.z, 1e-6f), h_local.x, h_local.y, ax, ay);
  const float lambda_v = ggx_lambda(n_dot_v, v_local.x, v_local.y, ax, ay);
  const float g1 = 1.0f / (1.0f + lambda_v);
  return d * g1 * v_dot_h / n_dot_v;
}

#define CRT_EON_A (0.5f - 2.0f / (3.0f * CRT_PI))
#define CRT_EON_B (2.0f / 3.0f - 28.0f / (15.0f * CRT_PI))

__device__ __forceinline__ float eon_albedo_approx(float mu, float roughness) {
  const float mucomp = 1.0f - rclamp(mu, 0.0f, 1.0f);
  const float G1 = 0.057108529f, G2 = 0.49188187f, G3 = -0.33218144f, G4 = 0.071442995f;
  const float g_over_pi = mucomp * (G1 + mucomp * (G2 + mucomp * (G3 + mucomp * G4)));
  return (1.0f + roughness * g_over_pi) / (1.0f + CRT_EON_A * roughness);
}
__device__ __forceinline__ V3 eon_diffuse(V3 rho, float roughness, V3 v_local, V3 l_local) {
  rho = vclamp(rho, splat(0.0f), splat(1.0f));
  const float mu_i = v_local.z, mu_o = l_local.z;
  const float s = dot(v_local, l_local) - mu_i * mu_o;
  const float s_over_t = s > 0.0f ? s / rmax(rmax(mu_i, mu_o), 1e-6f) : s;
  const float af = 1.0f / (1.0f + CRT_EON_A * roughness);
  const V3 f_ss = rho * (af / CRT_PI) * (1.0f + roughness * s_over_t);
  const float e_o = eon_albedo_approx(mu_o, roughness);
  const float e_i = eon_albedo_approx(mu_i, roughness);
  const float avg_e = af * (1.0f + CRT_EON_B * roughness);
  const V3 rho_ms = (rho * rho) * avg_e / (splat(1.0f) - rho * (1.0f - avg_e));
  const float EPS = 1.0e-7f;
  const V3 f_ms = rho_ms * (1.0f / CRT_PI) * (rmax(1.0f - e_o, EPS) * rmax(1.0f - e_i, EPS) / rmax(1.0f - avg_e, EPS));
  return f_ss + f_ms;
}
__device__ __forceinline__ V3 fresnel_f82_tint(float cos_theta, V3 f0, V3 tint) {
  const float MU_BAR = 1.0f / 7.0f;
  const float mu = rclamp(cos_theta, 0.0f, 1.0f);
  const V3 one = splat(1.0f);
  const V3 fs_bar = f0 + (one - f0) * pow5_(1.0f - MU_BAR);
  const float denom = MU_BAR * pow6_(1.0f - MU_BAR);
  const V3 a = fs_bar * (one - tint) / denom;
  const V3 fs_mu = f0 + (one - f0) * pow5_(1.0f - mu);
  return vclamp(fs_mu - a * mu * pow6_(1.0f - mu), splat(0.0f), one);
}
__device__ __forceinline__ float fresnel_dielectric(float cos_i, float eta_i, float eta_t) {
  cos_i = rclamp(cos_i, 0.0f, 1.0f);
  const float sin2_t = (eta_i / eta_t) * (eta_i / eta_t) * (1.0f - cos_i * cos_i);
  if (sin2_t >= 1.0f) return 1.0f;
  const float cos_t = sqrtf(1.0f - sin2_t);
  const float r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
  const float r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
  return 0.5f * (r_par * r_par + r_perp * r_perp);
}
__device__ __forceinline__ void tangent_frame(V3 n, V3 &t, V3 &b) {  // Duff et al. 2017
  const float sign = n.z >= 0.0f ? 1.0f : -1.0f;
  const float a = -1.0f / (sign + n.z);
  const float bb = n.x * n.y * a;
  t = v3(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
  b = v3(bb, sign + n.y * n.y * a, -n.y);
}
__device__ __forceinline__ float sheen_charlie(float n_dot_v, float n_dot_l, float n_dot_h, float roughness) {
  const float alpha = rmax(roughness, 0.05f);
  const float inv_alpha = 1.0f / alpha;
  const float sin2 = rmax(1.0f - n_dot_h * n_dot_h, 0.0f);
  const float d = (2.0f + inv_alpha) * pow_det(sin2, inv_alpha * 0.5f) / (2.0f * CRT_PI);
  const float vis = 1.0f / (4.0f * rmax(n_dot_l + n_dot_v - n_dot_l * n_dot_v, 1e-4f));
  return d * vis;
}
__device__ __forceinline__ V3 coat_darkening_factor(V3 base_color, float coat_ior, float darkening) {
  const float f_avg = f0_from_ior(coat_ior) + (1.0f - f0_from_ior(coat_ior)) * 0.05f;
  const V3 one = splat(1.0f);
  const V3 dark = base_color / vmax(one - (one - base_color) * f_avg, splat(1e-4f));
  return one * (1.0f - darkening) + dark * darkening;
}
__device__ __forceinline__ float lambda_rgb(int i) { return i == 0 ? 615.0f : (i == 1 ? 545.0f : 465.0f); }
__device__ __forceinline__ float cauchy_ior(float n_d, float v_d, float lambda_nm) {
  const float C = 656.3f, D = 587.6f, F = 486.1f;
  const float b = (n_d - 1.0f) / (v_d * (1.0f / (F * F) - 1.0f / (C * C)));
  const float a = n_d - b / (D * D);
  return a + b / (lambda_nm * lambda_nm);
}
__device__ __forceinline__ float fresnel_amplitude(float eta_i, float eta_t, float cos_i, float cos_t) {
  const float rs = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
  const float rp = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
  return 0.5f * (rs + rp);
}
__device__ __forceinline__ float thin_film_lambda(float cos_theta_1, float eta_1, float eta_film, float eta_2,
                                                  float thickness_nm, float lambda_nm) {
  const float cos1 = rclamp(cos_theta_1, 0.0f, 1.0f);
  const float sin2_1 = 1.0f - cos1 * cos1;
  const float sin2_film = pow2_(eta_1 / eta_film) * sin2_1;
  if (sin2_film >= 1.0f) return 1.0f;
  const float cos_film = sqrtf(1.0f - sin2_film);
  const float sin2_base = pow2_(eta_film / eta_2) * sin2_film;
  if (sin2_base >= 1.0f) return 1.0f;
  const float cos_base = sqrtf(1.0f - sin2_base);
  const float r_a = fresnel_amplitude(eta_1, eta_film, cos1, cos_film);
  const float r_b = fresnel_amplitude(eta_film, eta_2, cos_film, cos_base);
  const float opd = 2.0f * eta_film * thickness_nm * cos_film;
  const float phi = 2.0f * CRT_PI * opd / lambda_nm;
  const float cos_phi = cos_det(phi);
  const float num = r_a * r_a + 2.0f * r_a * r_b * cos_phi + r_b * r_b;
  const float den = 1.0f + 2.0f * r_a * r_b * cos_phi + pow2_(r_a * r_b);
  return rclamp(num / rmax(den, 1e-8f), 0.0f, 1.0f);
}
__device__ __forceinline__ V3 thin_film_fresnel(float cos1, float eta1, float eta_film, float eta2, float thickness_nm) {
  return v3(thin_film_lambda(cos1, eta1, eta_film, eta2, thickness_nm, 615.0f),
            thin_film_lambda(cos1, eta1, eta_film, eta2, thickness_nm, 545.0f),
            thin_film_lambda(cos1, eta1, eta_film, eta2, thickness_nm, 465.0f));
}
__device__ __forceinline__ V3 thin_film_fresnel_metal(float cos1, float eta1, float eta_film, V3 f0, float thickness_nm) {
  float o[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float f0_c = rclamp(comp(f0, i), 0.0f, 0.9999f);
    const float sq = sqrtf(f0_c);
    const float eta_2 = (1.0f + sq) / (1.0f - sq);
    o[i] = thin_film_lambda(cos1, eta1, eta_film, eta_2, thickness_nm, lambda_rgb(i));
  }
  return v3(o[0], o[1], o[2]);
}

// ---- material/openpbr.rs ----
struct LobePmf { float p_diffuse, p_specular, p_coat, p_fuzz, p_transmission; };
__device__ __forceinline__ float luma(V3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }

__device__ __forceinline__ LobePmf lobe_pmf(const CrtMaterial &m) {  // openpbr.rs:317-378
  const float f0_diel = f0_from_ior(m.specular_ior);
  const float f0_coat = f0_from_ior(m.coat_ior);
  const float base_luma = rmax(luma(ld3(m.base_color)), 0.02f);
  const float spec_luma = rmax(luma(ld3(m.specular_color)), 0.02f);
  const float fuzz_luma = rmax(luma(ld3(m.fuzz_color)), 0.02f);
  const float w_metal = m.base_metalness * m.specular_weight * rmax(luma(ld3(m.base_color) * m.base_weight), 0.02f);
  const float w_diel_spec = (1.0f - m.base_metalness) * m.specular_weight * spec_luma * f0_diel;
  const float w_specular = rmax(w_metal + w_diel_spec, 1e-4f);
  const float w_diffuse = rmax((1.0f - m.base_metalness) * (1.0f - m.transmission_weight) * m.base_weight * base_luma *
                                   (1.0f - f0_diel), 1e-4f);
  const float w_coat = rmax(m.coat_weight * f0_coat, 1e-6f);
  const float w_fuzz = rmax(m.fuzz_weight * fuzz_luma, 1e-6f);
  const float trans_luma = rmax(luma(ld3(m.transmission_color)), 0.02f);
  const float w_transmission =
      m.transmission_weight > 0.0f ? rmax((1.0f - m.base_metalness) * m.transmission_weight * trans_luma, 1e-4f) : 0.0f;
  const float total = w_diffuse + w_specular + w_coat + w_fuzz + w_transmission;
  return LobePmf{w_diffuse / total, w_specular / total, w_coat / total, w_fuzz / total, w_transmission / total};
}
enum { LOBE_DIFFUSE, LOBE_SPECULAR, LOBE_COAT, LOBE_FUZZ, LOBE_TRANSMISSION };
__device__ __forceinline__ int lobe_pick(const LobePmf &p, float u) {  // openpbr.rs:380-398
  float acc = p.p_diffuse;
  if (u < acc) return LOBE_DIFFUSE;
  acc += p.p_specular;
  if (u < acc) return LOBE_SPECULAR;
  acc += p.p_coat;
  if (u < acc) return LOBE_COAT;
  acc += p.p_fuzz;
  if (u < acc) return LOBE_FUZZ;
  return LOBE_TRANSMISSION;
}

__device__ __forceinline__ V3 eval_diffuse(const CrtMaterial &m, V3 v_local, V3 l_local, float f_avg_diel) {
  if (l_local.z <= 0.0f || v_local.z <= 0.0f) return splat(0.0f);
  const float presence = m.base_weight * (1.0f - m.base_metalness) * (1.0f - m.transmission_weight);
  if (presence <= 0.0f) return splat(0.0f);
  const V3 diffuse_color = lerp(ld3(m.base_color), ld3(m.subsurface_color), m.subsurface_weight);
  const V3 rho = diffuse_color * presence;
  return eon_diffuse(rho, m.base_diffuse_roughness, v_local, l_local) * (1.0f - f_avg_diel);
}

// SIMPLE (here and below): no material of the scene has a coat, fuzz, thin film, transmission or subsurface
// (material class <= 1 for the whole table, pathtrace.hip). The arms those weights gate are then compiled out — they are
// never taken, so nothing changes but the kernel: the thin-film Airy sums, the coat passage's pow(), the sheen, rough
// transmission with dispersion are the largest bodies and the heaviest users of f64 constants in the vertex code.
template <bool SIMPLE>
__device__ __forceinline__ V3 eval_specular(const CrtMaterial &m, V3 v_local, V3 l_local, V3 h_local, float ax, float ay) {
  const float n_dot_v = rmax(v_local.z, 1e-4f);
  const float n_dot_l = rmax(l_local.z, 1e-4f);
  const float n_dot_h = rmax(h_local.z, 1e-4f);
  const float v_dot_h = rmax(dot(v_local, h_local), 1e-4f);
  const float d = ggx_d(n_dot_h, h_local.x, h_local.y, ax, ay);
  const float g = ggx_g2(n_dot_v, v_local.x, v_local.y, n_dot_l, l_local.x, l_local.y, ax, ay);
  const float f0_diel_scalar = f0_from_ior(m.specular_ior);
  const float outer_ior = m.coat_weight > 0.0f ? m.coat_ior : 1.0f;
  const float tf_thickness_nm = m.thin_film_thickness * 1000.0f;
  V3 diel_term = splat(0.0f);
  if (m.base_metalness < 1.0f) {
    const V3 f0_diel_base = ld3(m.specular_color) * f0_diel_scalar * m.specular_weight;
    V3 f_diel;
    if (!SIMPLE && m.thin_film_weight > 0.0f) {
      const V3 f_normal = fresnel_schlick(v_dot_h, f0_diel_base);
      const V3 f_iri = thin_film_fresnel(v_dot_h, outer_ior, m.thin_film_ior, m.specular_ior, tf_thickness_nm);
      f_diel = f_normal * (1.0f - m.thin_film_weight) + f_iri * m.thin_film_weight;
    } else {
      f_diel = fresnel_schlick(v_dot_h, f0_diel_base);
    }
    if (!SIMPLE && m.thin_walled && m.transmission_weight > 0.0f) {
      const float f_phys = fresnel_schlick_scalar(v_dot_h, f0_diel_scalar);
      const float boost = 2.0f / (1.0f + f_phys);
      f_diel = f_diel * (1.0f + (boost - 1.0f) * m.transmission_weight);
    }
    diel_term = f_diel * (1.0f - m.base_metalness);
  }
  V3 metal_term = splat(0.0f);
  if (m.base_metalness > 0.0f) {
    const V3 metal_f0 = ld3(m.base_color) * m.base_weight;
    const V3 f_metal_base = fresnel_f82_tint(v_dot_h, metal_f0, ld3(m.specular_color));
    V3 f_metal;
    if (!SIMPLE && m.thin_film_weight > 0.0f) {
      const V3 f_iri = thin_film_fresnel_metal(v_dot_h, outer_ior, m.thin_film_ior, metal_f0, tf_thickness_nm);
      f_metal = f_metal_base * (1.0f - m.thin_film_weight) + f_iri * m.thin_film_weight;
    } else {
      f_metal = f_metal_base;
    }
    f_metal = f_metal * m.specular_weight;
    metal_term = f_metal * m.base_metalness;
  }
  const float brdf = d * g / (4.0f * n_dot_v * n_dot_l);
  return (metal_term + diel_term) * brdf;
}

__device__ __forceinline__ V3 eval_coat(const CrtMaterial &m, V3 v_local, V3 l_local, V3 h_local, float ax, float ay) {
  const float n_dot_v = rmax(v_local.z, 1e-4f);
  const float n_dot_l = rmax(l_local.z, 1e-4f);
  const float n_dot_h = rmax(h_local.z, 1e-4f);
  const float v_dot_h = rmax(dot(v_local, h_local), 1e-4f);
  const float d = ggx_d(n_dot_h, h_local.x, h_local.y, ax, ay);
  const float g = ggx_g2(n_dot_v, v_local.x, v_local.y, n_dot_l, l_local.x, l_local.y, ax, ay);
  const float f = fresnel_schlick_scalar(v_dot_h, f0_from_ior(m.coat_ior));
  const float brdf = d * g / (4.0f * n_dot_v * n_dot_l);
  return splat(m.coat_weight * f * brdf);
}
__device__ __forceinline__ V3 coat_passage(const CrtMaterial &m, float cos_theta) {  // openpbr.rs:587-603
  const float cos_i = rclamp(cos_theta, 1e-4f, 1.0f);
  const float eta = rmax(m.coat_ior, 1e-4f);
  const float sin2_t = (1.0f - cos_i * cos_i) / (eta * eta);
  const float cos_t = sqrtf(rmax(1.0f - sin2_t, 0.0f));
  const float path_length = 1.0f / rmax(cos_t, 1e-3f);
  const V3 cc = vclamp(ld3(m.coat_color), splat(0.0f), splat(1.0f));
  const float e = 0.5f * path_length;
  const V3 one_passage = v3(pow_det(cc.x, e), pow_det(cc.y, e), pow_det(cc.z, e));
  const V3 absorb = lerp(splat(1.0f), one_passage, m.coat_weight);
  const float f_coat = fresnel_schlick_scalar(cos_i, f0_from_ior(m.coat_ior));
  return absorb * (1.0f - m.coat_weight * f_coat);
}
template <bool SIMPLE>
__device__ __forceinline__ V3 coat_attenuation(const CrtMaterial &m, float cos_v, float cos_l) {
  if (SIMPLE || m.coat_weight <= 0.0f) return splat(1.0f);
  const V3 dark = coat_darkening_factor(ld3(m.base_color), m.coat_ior, m.coat_darkening);
  return coat_passage(m, cos_v) * coat_passage(m, cos_l) * dark;
}
__device__ __forceinline__ V3 eval_fuzz(const CrtMaterial &m, V3 v_local, V3 l_local, V3 h_local) {
  const float n_dot_v = rmax(v_local.z, 1e-4f);
  const float n_dot_l = rmax(l_local.z, 1e-4f);
  const float n_dot_h = rmax(h_local.z, 0.0f);
  return ld3(m.fuzz_color) * m.fuzz_weight * sheen_charlie(n_dot_v, n_dot_l, n_dot_h, m.fuzz_roughness);
}

__device__ __forceinline__ bool transmission_is_continuous(const CrtMaterial &m) {
  return m.transmission_weight > 0.0f && !m.thin_walled;
}
__device__ __forceinline__ V3 dispersive_ior(float n_d, float abbe, float scale) {  // openpbr.rs:736-751
  if (scale <= 0.0f || n_d == 1.0f) return splat(n_d);
  const bool inverted = n_d < 1.0f;
  const float n_above_one = inverted ? 1.0f / n_d : n_d;
  const float v_d = rmax(rmax(abbe, 1.0f) / scale, 1.0f);
  float o[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float n = cauchy_ior(n_above_one, v_d, lambda_rgb(c));
    o[c] = inverted ? 1.0f / n : n;
  }
  return v3(o[0], o[1], o[2]);
}
__device__ __forceinline__ V3 transmission_iors(const CrtMaterial &m) {
  return dispersive_ior(m.specular_ior, m.transmission_dispersion_abbe_number, m.transmission_dispersion_scale);
}
__device__ __forceinline__ void transmission_alphas(const CrtMaterial &m, float &ax, float &ay) {
  roughness_to_alpha(rmax(m.specular_roughness, 0.01f), m.specular_roughness_anisotropy, ax, ay);
}
__device__ __forceinline__ void eval_transmission_channel(const CrtMaterial &m, V3 v_local, V3 l_local, bool entering,
                                                          float ior, float &btdf_o, float &pdf_o) {  // Walter 2007
  btdf_o = 0.0f; pdf_o = 0.0f;
  const float eta_i = entering ? 1.0f : ior, eta_t = entering ? ior : 1.0f;
  V3 h = -(v_local * eta_i + l_local * eta_t);
  if (len2(h) < 1e-12f) return;
  h = normalize(h);
  if (h.z < 0.0f) h = -h;
  const float v_dot_h = dot(v_local, h);
  const float l_dot_h = dot(l_local, h);
  if (v_dot_h <= 1e-6f || l_dot_h >= -1e-6f) return;
  float ax, ay;
  transmission_alphas(m, ax, ay);
  const float n_dot_v = rmax(v_local.z, 1e-6f);
  const float n_dot_l = rmax(-l_local.z, 1e-6f);
  const float d = ggx_d(rmax(h.z, 1e-6f), h.x, h.y, ax, ay);
  const float g = ggx_g2(n_dot_v, v_local.x, v_local.y, n_dot_l, l_local.x, l_local.y, ax, ay);
  const float f = fresnel_dielectric(v_dot_h, eta_i, eta_t);
  const float denom = eta_i * v_dot_h + eta_t * l_dot_h;
  const float denom2 = denom * denom;
  if (denom2 < 1e-10f) return;
  const float btdf = (v_dot_h * -l_dot_h) / (n_dot_v * n_dot_l) * (eta_t * eta_t * (1.0f - f) * d * g / denom2);
  const float p_h = pdf_vndf_h(v_local, h, ax, ay);
  const float jacobian = eta_t * eta_t * -l_dot_h / denom2;
  btdf_o = rmax(btdf, 0.0f);
  pdf_o = p_h * jacobian;
}
__device__ __forceinline__ void eval_transmission(const CrtMaterial &m, V3 v_local, V3 l_local, bool entering, V3 &value,
                                                  float &pdf) {  // openpbr.rs:924-949
  const V3 color = m.transmission_depth > 0.0f ? splat(1.0f) : ld3(m.transmission_color);
  const V3 tint = color * (m.transmission_weight * (1.0f - m.base_metalness));
  const V3 iors = transmission_iors(m);
  if (m.transmission_dispersion_scale <= 0.0f) {
    float btdf, p;
    eval_transmission_channel(m, v_local, l_local, entering, iors.y, btdf, p);
    value = tint * btdf; pdf = p;
    return;
  }
  float val[3];
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float btdf, p;
    eval_transmission_channel(m, v_local, l_local, entering, comp(iors, c), btdf, p);
    val[c] = btdf;
    acc += p / 3.0f;
  }
  value = tint * v3(val[0], val[1], val[2]);
  pdf = acc;
}

template <bool SIMPLE>
__device__ __forceinline__ V3 eval_all(const CrtMaterial &m, V3 v_local, V3 l_local, bool entering) {  // :629-683
  if (v_local.z <= 0.0f) return splat(0.0f);
  if (l_local.z <= 0.0f) {
    if (SIMPLE || !transmission_is_continuous(m)) return splat(0.0f);
    V3 val; float p;
    eval_transmission(m, v_local, l_local, entering, val, p);
    return val;
  }
  const V3 h_local = normalize(v_local + l_local);
  float ax, ay;
  roughness_to_alpha(m.specular_roughness, m.specular_roughness_anisotropy, ax, ay);
  const float f_avg_diel = f0_from_ior(m.specular_ior);
  const V3 diffuse = eval_diffuse(m, v_local, l_local, f_avg_diel);
  const V3 specular = eval_specular<SIMPLE>(m, v_local, l_local, h_local, ax, ay);
  V3 coat = splat(0.0f);
  if (!SIMPLE && m.coat_weight > 0.0f) {
    float axc, ayc;
    roughness_to_alpha(m.coat_roughness, m.coat_roughness_anisotropy, axc, ayc);
    coat = eval_coat(m, v_local, l_local, h_local, axc, ayc);
  }
  const V3 fuzz = (!SIMPLE && m.fuzz_weight > 0.0f) ? eval_fuzz(m, v_local, l_local, h_local) : splat(0.0f);
  const V3 coat_atten = coat_attenuation<SIMPLE>(m, v_local.z, l_local.z);
  const float base_atten = rclamp(1.0f - m.fuzz_weight, 0.0f, 1.0f);
  return fuzz + (coat + coat_atten * (diffuse + specular)) * base_atten;
}

template <bool SIMPLE>
__device__ __forceinline__ float pdf_all(const CrtMaterial &m, const LobePmf &pmf, V3 v_local, V3 l_local, bool entering) {
  if (v_local.z <= 0.0f) return 0.0f;  // openpbr.rs:689-722
  if (l_local.z <= 0.0f) {
    if (SIMPLE || !transmission_is_continuous(m)) return 0.0f;
    V3 val; float p;
    eval_transmission(m, v_local, l_local, entering, val, p);
    return pmf.p_transmission * p;
  }
  const V3 h_local = normalize(v_local + l_local);
  float ax, ay, axc, ayc;
  roughness_to_alpha(m.specular_roughness, m.specular_roughness_anisotropy, ax, ay);
  roughness_to_alpha(m.coat_roughness, m.coat_roughness_anisotropy, axc, ayc);
  const float pdf_cosine = rmax(l_local.z, 0.0f) / CRT_PI;
  const float pdf_specular = pdf_vndf(v_local, h_local, ax, ay);
  const float pdf_coat = pdf_vndf(v_local, h_local, axc, ayc);
  return pmf.p_diffuse * pdf_cosine + pmf.p_specular * pdf_specular + pmf.p_coat * pdf_coat + pmf.p_fuzz * pdf_cosine;
}

__device__ __forceinline__ bool sample_transmission_rough(const CrtMaterial &m, V3 v_local, bool entering,
                                                          float dispersion_u, float u1, float u2, V3 &l_out) {
  const V3 iors = transmission_iors(m);  // openpbr.rs:958-993
  float ior;
  if (m.transmission_dispersion_scale > 0.0f) {
    if (dispersion_u < 1.0f / 3.0f) ior = iors.x;
    else if (dispersion_u < 2.0f / 3.0f) ior = iors.y;
    else ior = iors.z;
  } else ior = iors.y;
  const float eta_i = entering ? 1.0f : ior, eta_t = entering ? ior : 1.0f;
  const float eta_rel = eta_i / eta_t;
  float ax, ay;
  transmission_alphas(m, ax, ay);
  const V3 h = sample_vndf(v_local, ax, ay, u1, u2);
  const float cos_i = dot(v_local, h);
  if (cos_i <= 1e-6f) return false;
  const float sin2_t = eta_rel * eta_rel * (1.0f - cos_i * cos_i);
  if (sin2_t >= 1.0f) return false;
  const float cos_t = sqrtf(1.0f - sin2_t);
  const V3 l = normalize((-v_local) * eta_rel + h * (eta_rel * cos_i - cos_t));
  if (l.z >= -1e-6f) return false;
  l_out = l;
  return true;
}
__device__ __forceinline__ void sample_transmission_thin(const CrtMaterial &m, V3 ray_dir, const HitRec &rec, V3 &dir_o,
                                                         V3 &throughput) {  // openpbr.rs:779-812
  const V3 dir = normalize(ray_dir);
  dir_o = dir;
  const float cos_i = rclamp(dot(-dir, rec.normal), 0.0f, 1.0f);
  const float eta = rmax(m.specular_ior, 1e-4f);
  const float sin2_t = (1.0f - cos_i * cos_i) / (eta * eta);
  if (sin2_t >= 1.0f) { throughput = splat(0.0f); return; }
  const float cos_t = sqrtf(1.0f - sin2_t);
  const float f = fresnel_dielectric(cos_i, 1.0f, eta);
  const float window_transmittance = (1.0f - f) / (1.0f + f);
  const float path_length = 1.0f / rmax(cos_t, 1e-4f);
  const V3 tc = vclamp(ld3(m.transmission_color), splat(0.0f), splat(1.0f));
  const V3 tint = v3(pow_det(tc.x, path_length), pow_det(tc.y, path_length), pow_det(tc.z, path_length));
  throughput = tint * (window_transmittance * m.transmission_weight);
}

struct Frame3 { V3 n, t, b; };
__device__ __forceinline__ Frame3 frame_new(V3 n) { Frame3 f; f.n = n; tangent_frame(n, f.t, f.b); return f; }
__device__ __forceinline__ V3 to_local(const Frame3 &f, V3 v) { return v3(dot(v, f.t), dot(v, f.b), dot(v, f.n)); }
__device__ __forceinline__ V3 to_world(const Frame3 &f, V3 l) { return f.t * l.x + f.b * l.y + f.n * l.z; }

// Material::scatter_importance (material.rs:40-45): OpenPBR::scatter_resolved (openpbr.rs:1026-1136);
// Emissive never scatters (emissive.rs:30-38).
template <bool SIMPLE>
__device__ bool mat_scatter(const CrtMaterial &m, V3 ray_dir, const HitRec &rec, Sampler dom, Scatter &out,
                            const uint32_t *sobol_tab) {
  if (m.kind == CRT_MAT_EMISSIVE) return false;
  out.medium = false;
  const Frame3 frame = frame_new(rec.normal);
  const V3 v_world = -normalize(ray_dir);
  const V3 v_local = to_local(frame, v_world);
  if (v_local.z <= 0.0f) return false;
  float s[4];
  draw_sample4(dom, s, sobol_tab);
  const LobePmf pmf = lobe_pmf(m);
  const int lobe = lobe_pick(pmf, s[0]);
  if (lobe == LOBE_TRANSMISSION) {
    // SIMPLE keeps this arm's thin form: lobe_pick falls through to LOBE_TRANSMISSION whenever u lands past the
    // rounded sum of the other four masses, even at transmission weight 0 (the reference does the same)
    if (!SIMPLE && transmission_is_continuous(m)) {
      V3 l_local;
      if (!sample_transmission_rough(m, v_local, rec.front_face, s[3], s[1], s[2], l_local)) return false;
      const V3 l_world = to_world(frame, l_local);
      const float pdf = rmax(pdf_all<SIMPLE>(m, pmf, v_local, l_local, rec.front_face), 1e-4f);
      const V3 brdf = eval_all<SIMPLE>(m, v_local, l_local, rec.front_face);
      out.origin = rec.p + l_world * 1e-4f;
      out.dir = l_world;
      out.value = brdf * fabs_(l_local.z);
      out.pdf = pdf;
      out.delta = false;
      out.medium = rec.front_face;  // openpbr.rs:1061-1066; the caller checks that the material has an interior at all
      return true;
    }
    V3 dir, throughput;
    sample_transmission_thin(m, ray_dir, rec, dir, throughput);
    const float p_select = rmax(pmf.p_transmission, 1e-4f);
    out.origin = rec.p; out.dir = dir; out.value = throughput / p_select; out.pdf = 1.0f; out.delta = true;
    return true;
  }
  V3 l_local;
  if (lobe == LOBE_DIFFUSE || lobe == LOBE_FUZZ) {
    l_local = cosine_hemisphere(s[1], s[2]);
  } else {
    float ax, ay;
    if (lobe == LOBE_SPECULAR) roughness_to_alpha(m.specular_roughness, m.specular_roughness_anisotropy, ax, ay);
    else roughness_to_alpha(m.coat_roughness, m.coat_roughness_anisotropy, ax, ay);
    const V3 h_local = sample_vndf(v_local, ax, ay, s[1], s[2]);
    const V3 l = h_local * (2.0f * dot(v_local, h_local)) - v_local;
    if (l.z <= 0.0f) return false;
    l_local = l;
  }
  const float pdf = rmax(pdf_all<SIMPLE>(m, pmf, v_local, l_local, rec.front_face), 1e-4f);
  const V3 brdf = eval_all<SIMPLE>(m, v_local, l_local, rec.front_face);
  const float n_dot_l = rmax(l_local.z, 0.0f);
  out.origin = rec.p;
  out.dir = to_world(frame, l_local);
  out.value = brdf * n_dot_l;
  out.pdf = pdf;
  out.delta = false;
  return true;
}

// Material::eval (material.rs:71-74): OpenPBR::eval_resolved (openpbr.rs:1138-1158); None for Emissive.
template <bool SIMPLE>
__device__ bool mat_eval(const CrtMaterial &m, V3 ray_dir, const HitRec &rec, V3 wi, V3 &value, float &pdf) {
  if (m.kind == CRT_MAT_EMISSIVE) return false;
  const Frame3 frame = frame_new(rec.normal);
  const V3 v_local = to_local(frame, -normalize(ray_dir));
  if (v_local.z <= 0.0f) return false;
  const V3 l_local = to_local(frame, normalize(wi));
  const LobePmf pmf = lobe_pmf(m);
  pdf = rmax(pdf_all<SIMPLE>(m, pmf, v_local, l_local, rec.front_face), 1e-4f);
  value = eval_all<SIMPLE>(m, v_local, l_local, rec.front_face) * fabs_(l_local.z);
  return true;
}

// Material::emitted_directional (material.rs:112-115; openpbr.rs:1211-1218).
template <bool SIMPLE>
__device__ __forceinline__ V3 mat_emitted_directional(const CrtMaterial &m, float cos_theta_o) {
  if (m.kind == CRT_MAT_EMISSIVE) return ld3(m.emission_color);
  const V3 uncoated = ld3(m.emission_color) * m.emission_luminance;
  if (SIMPLE || m.coat_weight <= 0.0f) return uncoated;
  const V3 dark = coat_darkening_factor(ld3(m.base_color), m.coat_ior, m.coat_darkening);
  return uncoated * coat_passage(m, cos_theta_o) * dark;
}

// ---- light.rs: area lights ----
struct LightSample { V3 direction; float distance; V3 radiance; float pdf; };

__device__ __forceinline__ float solid_angle_pdf(const CrtLight &l, V3 from, V3 light_point) {  // light.rs:180-187
  const V3 direction = light_point - from;
  const float d2 = len2(direction);
  const V3 dir_to_light = normalize(direction);
  V3 ln;
  float area;
  if (l.kind == CRT_LIGHT_SPHERE) {
    ln = normalize(light_point - ld3(l.center));
    area = 4.0f * CRT_PI * l.radius * l.radius;
  } else {
    ln = ld3(l.normal);
    area = length(cross(ld3(l.edge_u), ld3(l.edge_v)));
  }
  const float cosine = rmax(dot(ln, -dir_to_light), 0.0f);
  return d2 / (cosine * area + 1e-4f);
}
__device__ __forceinline__ V3 align_to_normal(V3 local, V3 normal) {  // common.rs:176-188
  const V3 up = fabs_(normal.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
  const V3 tangent = normalize(cross(normal, up));
  const V3 bitangent = cross(normal, tangent);
  return tangent * local.x + bitangent * local.y + normal * local.z;
}
// Light::escaped: what a ray leaving the scene along `direction` (unit) sees of a light at infinity
// (light.rs:141-146; DistantLight :268-282, :300-303; uniform DomeLight :340-355, :385-388).
__device__ __forceinline__ bool light_escaped(const CrtLight &l, V3 direction, V3 &radiance, float &pdf) {
  if (l.kind == CRT_LIGHT_DISTANT) {
    if (!(dot(direction, -ld3(l.normal)) >= l.radius)) return false;
    const float omega = rmax(l.center[0], 1e-12f);
    radiance = ld3(l.radiance) / omega;
    pdf = 1.0f / omega;
    return true;
  }
  if (l.kind == CRT_LIGHT_DOME) {
    radiance = ld3(l.radiance);
    pdf = 1.0f / (4.0f * CRT_PI);
    return true;
  }
  return false;
}
// INF: the light list may hold lights at infinity (compiled out of the kernels of scenes that have none).
template <bool INF>
__device__ __forceinline__ bool light_sample_li(const CrtLight &l, V3 from, float u, float v, LightSample &out) {
  if (INF && l.kind == CRT_LIGHT_DISTANT) {  // light.rs:285-298: uniform direction within the cone around -direction
    const float cos_theta = 1.0f - u * (1.0f - l.radius);
    const float sin_theta = sqrtf(rmax(1.0f - cos_theta * cos_theta, 0.0f));
    float sp, cp;
    sincos_det(2.0f * CRT_PI * v, sp, cp);
    const float omega = rmax(l.center[0], 1e-12f);
    out.direction = normalize(align_to_normal(v3(sin_theta * cp, sin_theta * sp, cos_theta), -ld3(l.normal)));
    out.distance = CRT_INF;
    out.radiance = ld3(l.radiance) / omega;
    out.pdf = 1.0f / omega;
    return true;
  }
  if (INF && l.kind == CRT_LIGHT_DOME) {  // light.rs:358-383 without a map: uniform over the sphere
    const float z = 1.0f - 2.0f * u;
    const float r = sqrtf(rmax(1.0f - z * z, 0.0f));
    float sp, cp;
    sincos_det((2.0f * CRT_PI) * v, sp, cp);
    out.direction = v3(r * cp, z, r * sp);
    out.distance = CRT_INF;
    out.radiance = ld3(l.radiance) * splat(1.0f);
    out.pdf = 1.0f / (4.0f * CRT_PI);
    return true;
  }
  V3 lp;  // light.rs:191-204
  if (l.kind == CRT_LIGHT_SPHERE) {
    const float theta = 2.0f * CRT_PI * u;
    const float phi = acos_det(1.0f - 2.0f * v);
    float sp, cp, st, ct;
    sincos_det(phi, sp, cp);
    sincos_det(theta, st, ct);
    lp = ld3(l.center) + v3(sp * ct, sp * st, cp) * l.radius;
  } else {
    lp = ld3(l.origin) + ld3(l.edge_u) * u + ld3(l.edge_v) * v;
  }
  const V3 to_light = lp - from;
  const float distance = length(to_light);
  if (distance < 1e-6f) return false;
  out.direction = to_light / distance;
  out.distance = distance;
  out.radiance = ld3(l.radiance);
  out.pdf = solid_angle_pdf(l, from, lp);
  return true;
}

// ---- camera.rs:71-84 ----
__device__ __forceinline__ void camera_get_ray(const CrtCamera &c, float s, float t, float lu, float lv, V3 &origin,
                                               V3 &dir) {
  V3 offset = splat(0.0f);
  if (c.lens_radius > 0.0f) {
    const V3 rd = concentric_disk(lu, lv) * c.lens_radius;
    offset = ld3(c.u) * rd.x + ld3(c.v) * rd.y;
  }
  origin = ld3(c.origin) + offset;
  dir = ld3(c.lower_left) + ld3(c.horizontal) * s + ld3(c.vertical) * t - ld3(c.origin) - offset;
}

// ---- filter.rs:182-205 (box, triangle: weight is exactly 1) ----
__device__ __forceinline__ float filter_offset(int kind, float r, float u) {
  if (kind == CRT_FILTER_BOX) return (0.5f - r) + (2.0f * r) * u;
  const float x = u < 0.5f ? r * (sqrtf(2.0f * u) - 1.0f) : r * (1.0f - sqrtf(2.0f * (1.0f - u)));
  return 0.5f + x;
}


// ---- carried interior medium (medium.rs:22-184, openpbr.rs:225-258) ----
struct DevMedium {  // 48 bytes; one per material, present == 0 where the interior neither absorbs nor scatters
  float sigma_a[3], sigma_s[3];
  float g;
  float sigma_bar;      // max(sigma_t_max, 1e-4): the free-flight majorant (tracer.rs:1161, :1258)
  uint32_t present;     // interior_medium() is Some
  uint32_t scattering;  // Medium::is_scattering (medium.rs:125-127)
  uint32_t id;          // compact 1-based id carried by the path state (0 = no medium)
  uint32_t pad;
};
__device__ __forceinline__ V3 medium_sigma_t(const DevMedium &m) { return ld3(m.sigma_a) + ld3(m.sigma_s); }
__device__ __forceinline__ V3 medium_transmittance(const DevMedium &m, float t) {  // medium.rs:117-120
  const V3 e = medium_sigma_t(m) * t;
  return v3(exp_det(-e.x), exp_det(-e.y), exp_det(-e.z));
}
// e^{(sigma_bar - sigma_t) t}: what a scattering medium still owes after the free-flight competition
__device__ __forceinline__ V3 medium_chromatic(const DevMedium &m, float t) {  // tracer.rs:1354-1358, :1274
  const V3 e = (splat(m.sigma_bar) - medium_sigma_t(m)) * t;
  return v3(exp_det(e.x), exp_det(e.y), exp_det(e.z));
}
__device__ __forceinline__ void medium_from_material(const CrtMaterial &m, DevMedium &out) {
  out.present = 0; out.scattering = 0; out.id = 0; out.pad = 0; out.g = 0.0f; out.sigma_bar = 1e-4f;
  for (int i = 0; i < 3; i++) { out.sigma_a[i] = 0.0f; out.sigma_s[i] = 0.0f; }
  if (m.kind == CRT_MAT_EMISSIVE) return;
  const float trans_frac = m.transmission_weight;
  const float sss_frac = (1.0f - m.transmission_weight) * m.subsurface_weight;
  const float total = trans_frac + sss_frac;
  if (total <= 0.0f) return;
  // Medium::from_transmission (medium.rs:40-66)
  V3 ta = splat(0.0f), ts = splat(0.0f);
  float tg = 0.0f;
  if (!(m.transmission_depth <= 1e-6f)) {
    const float depth = m.transmission_depth;
    const V3 t = vclamp(ld3(m.transmission_color), splat(1e-4f), splat(1.0f));
    const V3 extinction = v3(-log_det(t.x), -log_det(t.y), -log_det(t.z)) / depth;
    ts = vmax(ld3(m.transmission_scatter), splat(0.0f)) / depth;
    ta = extinction - ts;
    const float mn = smin(smin(ta.x, ta.y), ta.z);
    if (mn < 0.0f) ta = ta - splat(mn);
    tg = rclamp(m.transmission_scatter_anisotropy, -0.999f, 0.999f);
  }
  V3 sa = ta, ss = ts;
  float g = tg;
  if (sss_frac > 0.0f) {  // Medium::from_subsurface (medium.rs:77-97) + Medium::blend (:103-114)
    const V3 mfp = vmax(splat(m.subsurface_radius) * ld3(m.subsurface_radius_scale), splat(1e-3f));
    const V3 sigma_t = splat(1.0f) / mfp;
    const float sg = rclamp(m.subsurface_scatter_anisotropy, -0.999f, 0.999f);
    const V3 a = vclamp(ld3(m.subsurface_color), splat(0.0f), splat(1.0f));
    const V3 inner = (splat(9.59217f) + a * 41.6808f) + (a * 17.7126f) * a;
    const V3 sqrt_inner = v3(sqrtf(inner.x), sqrtf(inner.y), sqrtf(inner.z));
    const V3 s = (splat(4.09712f) + a * 4.20863f) - sqrt_inner;
    const V3 s2 = s * s;
    const V3 alpha_ss = vclamp((splat(1.0f) - s2) / (splat(1.0f) - s2 * sg), splat(0.0f), splat(1.0f));
    const V3 bs = sigma_t * alpha_ss;
    const V3 ba = sigma_t - bs;
    const float wa = trans_frac / total, wb = sss_frac / total;
    sa = ta * wa + ba * wb;
    ss = ts * wa + bs * wb;
    const float ca = (((ts.x + ts.y) + ts.z) / 3.0f) * wa;
    const float cb = (((bs.x + bs.y) + bs.z) / 3.0f) * wb;
    g = (ca + cb > 1e-8f) ? (tg * ca + sg * cb) / (ca + cb) : 0.0f;
  }
  const float sigma_t_max = max_elem(sa + ss);
  if (sigma_t_max <= 1e-6f) return;  // openpbr.rs:253-257: inert interiors carry no medium
  out.sigma_a[0] = sa.x; out.sigma_a[1] = sa.y; out.sigma_a[2] = sa.z;
  out.sigma_s[0] = ss.x; out.sigma_s[1] = ss.y; out.sigma_s[2] = ss.z;
  out.g = g;
  out.sigma_bar = rmax(sigma_t_max, 1e-4f);
  out.present = 1;
  out.scattering = max_elem(ss) > 1e-6f ? 1u : 0u;
}
__device__ __forceinline__ V3 sample_henyey_greenstein(V3 wi, float g, float u1, float u2) {  // medium.rs:158-184
  float cos_theta;
  if (fabs_(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u1;
  else {
    const float sq = (1.0f - g * g) / (1.0f - g + 2.0f * g * u1);
    cos_theta = (1.0f + g * g - sq * sq) / (2.0f * g);
  }
  cos_theta = rclamp(cos_theta, -1.0f, 1.0f);
  const float sin_theta = sqrtf(rmax(1.0f - cos_theta * cos_theta, 0.0f));
  const float phi = 2.0f * CRT_PI * u2;
  const V3 up = fabs_(wi.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
  const V3 t = normalize(cross(wi, up));
  const V3 b = cross(wi, t);
  float sp, cp;
  sincos_det(phi, sp, cp);
  return normalize((t * (sin_theta * cp) + b * (sin_theta * sp)) + wi * cos_theta);
}

}  // namespace dev
}  // namespace crt
