/* ORACLE — TEST INFRASTRUCTURE ONLY (see ora_math.h header).
 *
 * CPU restatement of material/{brdf,openpbr,emissive}.rs, light.rs (area lights), camera.rs,
 * filter.rs (box/triangle) and the utils warps. Scalar f32, operation order as written in the reference.
 */
#include "ora_shade.h"

/* ------------------------------------------------------------------ */
/* utils/src/common.rs                                                 */
/* ------------------------------------------------------------------ */
float ora_balance_heuristic(float a, float b) { return a / (a + b + 1e-6f); }
float ora_power_heuristic(float a, float b) {
  float a2 = a * a, b2 = b * b;
  return a2 / (a2 + b2 + 1e-6f);
}
v3 ora_cosine_hemisphere(float u, float v) { /* common.rs:128-135 */
  float z = sqrtf(1.0f - v);
  float phi = 2.0f * ORA_PI * u;
  float s, c;
  ora_sincosf(phi, &s, &c);
  return v3_new(c * sqrtf(v), s * sqrtf(v), z);
}
v3 ora_concentric_disk(float u, float v) { /* common.rs:158-174 */
  float sx = 2.0f * u - 1.0f, sy = 2.0f * v - 1.0f;
  if (sx == 0.0f && sy == 0.0f) return v3_splat(0.0f);
  const float FRAC_PI_4 = 0.785398163397448309615660845819875721f, FRAC_PI_2 = 1.57079632679489661923132169163975144f;
  float r, theta;
  if (ora_abs(sx) > ora_abs(sy)) { r = sx; theta = FRAC_PI_4 * (sy / sx); }
  else { r = sy; theta = FRAC_PI_2 - FRAC_PI_4 * (sx / sy); }
  float s, c;
  ora_sincosf(theta, &s, &c);
  return v3_new(r * c, r * s, 0.0f);
}

/* ------------------------------------------------------------------ */
/* material/brdf.rs                                                    */
/* ------------------------------------------------------------------ */
static v3 fresnel_schlick(float cos_theta, v3 f0) { /* brdf.rs:10-12 */
  return v3_add(f0, v3_scale(v3_sub(v3_splat(1.0f), f0), ora_pow5(1.0f - cos_theta)));
}
static float fresnel_schlick_scalar(float cos_theta, float f0) { /* brdf.rs:15-17 */
  return f0 + (1.0f - f0) * ora_pow5(1.0f - cos_theta);
}
float ora_f0_from_ior(float ior) { float r = (ior - 1.0f) / (ior + 1.0f); return r * r; } /* brdf.rs:28-31 */
void ora_roughness_to_alpha_aniso(float roughness, float anisotropy, float *ax, float *ay) { /* brdf.rs:39-45 */
  float a = roughness * roughness;
  float inv = 1.0f - ora_clamp(anisotropy, 0.0f, 1.0f);
  float x = a * sqrtf(2.0f / (1.0f + inv * inv));
  float y = inv * x;
  *ax = ora_max(x, 1e-4f);
  *ay = ora_max(y, 1e-4f);
}
static float ggx_d_aniso(float n_dot_h, float h_dot_t, float h_dot_b, float ax, float ay) { /* brdf.rs:49-54 */
  float tx = h_dot_t / ax, ty = h_dot_b / ay;
  float term = tx * tx + ty * ty + n_dot_h * n_dot_h;
  return 1.0f / (ORA_PI * ax * ay * term * term);
}
static float ggx_lambda_aniso(float v_dot_n, float v_dot_t, float v_dot_b, float ax, float ay) { /* brdf.rs:57-63 */
  float vt = v_dot_t * ax, vb = v_dot_b * ay;
  float a2 = vt * vt + vb * vb;
  float n2 = ora_max(v_dot_n * v_dot_n, 1e-8f);
  return (-1.0f + sqrtf(1.0f + a2 / n2)) * 0.5f;
}
static float ggx_g2_smith_aniso(float vn, float vt, float vb, float ln, float lt, float lb, float ax, float ay) {
  float lv = ggx_lambda_aniso(vn, vt, vb, ax, ay); /* brdf.rs:67-80 */
  float ll = ggx_lambda_aniso(ln, lt, lb, ax, ay);
  return 1.0f / (1.0f + lv + ll);
}
v3 ora_sample_vndf(v3 v_local, float ax, float ay, float u1, float u2) { /* brdf.rs:86-109 */
  v3 vh = v3_normalize(v3_new(ax * v_local.x, ay * v_local.y, v_local.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  v3 t1 = lensq > 0.0f ? v3_divs(v3_new(-vh.y, vh.x, 0.0f), sqrtf(lensq)) : v3_new(1.0f, 0.0f, 0.0f);
  v3 t2 = v3_cross(vh, t1);
  float r = sqrtf(u1);
  float phi = 2.0f * ORA_PI * u2;
  float sp, cp;
  ora_sincosf(phi, &sp, &cp);
  float t1c = r * cp;
  float t2c_pre = r * sp;
  float s = 0.5f * (1.0f + vh.z);
  float t2c = (1.0f - s) * sqrtf(ora_max(1.0f - t1c * t1c, 0.0f)) + s * t2c_pre;
  v3 nh = v3_add(v3_add(v3_scale(t1, t1c), v3_scale(t2, t2c)),
                 v3_scale(vh, sqrtf(ora_max(1.0f - t1c * t1c - t2c * t2c, 0.0f))));
  return v3_normalize(v3_new(ax * nh.x, ay * nh.y, ora_max(nh.z, 0.0f)));
}
static float pdf_vndf_ggx_aniso_local(v3 v_local, v3 h_local, float ax, float ay) { /* brdf.rs:113-123 */
  float n_dot_v = ora_max(v_local.z, 1e-6f);
  float n_dot_h = ora_max(h_local.z, 1e-6f);
  float d = ggx_d_aniso(n_dot_h, h_local.x, h_local.y, ax, ay);
  float lambda_v = ggx_lambda_aniso(n_dot_v, v_local.x, v_local.y, ax, ay);
  float g1 = 1.0f / (1.0f + lambda_v);
  return d * g1 / (4.0f * n_dot_v);
}
static float pdf_vndf_h_aniso_local(v3 v_local, v3 h_local, float ax, float ay) { /* brdf.rs:129-136 */
  float n_dot_v = ora_max(v_local.z, 1e-6f);
  float v_dot_h = ora_max(v3_dot(v_local, h_local), 0.0f);
  float d = ggx_d_aniso(ora_max(h_local.z, 1e-6f), h_local.x, h_local.y, ax, ay);
  float lambda_v = ggx_lambda_aniso(n_dot_v, v_local.x, v_local.y, ax, ay);
  float g1 = 1.0f / (1.0f + lambda_v);
  return d * g1 * v_dot_h / n_dot_v;
}

#define EON_A (0.5f - 2.0f / (3.0f * ORA_PI))         /* brdf.rs:149 */
#define EON_B (2.0f / 3.0f - 28.0f / (15.0f * ORA_PI)) /* brdf.rs:152 */

float ora_eon_albedo_exact(float mu, float roughness) { /* brdf.rs:157-164 */
  mu = ora_clamp(mu, 1e-4f, 1.0f);
  float af = 1.0f / (1.0f + EON_A * roughness);
  float bf = roughness * af;
  float si = sqrtf(ora_max(1.0f - mu * mu, 0.0f));
  float g = si * (ora_acosf(mu) - si * mu) + (2.0f / 3.0f) * ((si / mu) * (1.0f - si * si * si) - si);
  return af + (bf / ORA_PI) * g;
}
float ora_eon_albedo_approx(float mu, float roughness) { /* brdf.rs:168-176 */
  float mucomp = 1.0f - ora_clamp(mu, 0.0f, 1.0f);
  const float G1 = 0.057108529f, G2 = 0.49188187f, G3 = -0.33218144f, G4 = 0.071442995f;
  float g_over_pi = mucomp * (G1 + mucomp * (G2 + mucomp * (G3 + mucomp * G4)));
  return (1.0f + roughness * g_over_pi) / (1.0f + EON_A * roughness);
}
v3 ora_eon_diffuse(v3 rho, float roughness, v3 v_local, v3 l_local) { /* brdf.rs:182-209 */
  rho = v3_clamp(rho, v3_splat(0.0f), v3_splat(1.0f));
  float mu_i = v_local.z, mu_o = l_local.z;
  float s = v3_dot(v_local, l_local) - mu_i * mu_o;
  float s_over_t = s > 0.0f ? s / ora_max(ora_max(mu_i, mu_o), 1e-6f) : s;
  float af = 1.0f / (1.0f + EON_A * roughness);
  v3 f_ss = v3_scale(v3_scale(rho, af / ORA_PI), 1.0f + roughness * s_over_t);
  float e_o = ora_eon_albedo_approx(mu_o, roughness);
  float e_i = ora_eon_albedo_approx(mu_i, roughness);
  float avg_e = af * (1.0f + EON_B * roughness);
  v3 rho_ms = v3_div(v3_scale(v3_mul(rho, rho), avg_e), v3_sub(v3_splat(1.0f), v3_scale(rho, 1.0f - avg_e)));
  const float EPS = 1.0e-7f;
  v3 f_ms = v3_scale(v3_scale(rho_ms, 1.0f / ORA_PI),
                     ora_max(1.0f - e_o, EPS) * ora_max(1.0f - e_i, EPS) / ora_max(1.0f - avg_e, EPS));
  return v3_add(f_ss, f_ms);
}
v3 ora_fresnel_f82_tint(float cos_theta, v3 f0, v3 tint) { /* brdf.rs:217-224 */
  const float MU_BAR = 1.0f / 7.0f;
  float mu = ora_clamp(cos_theta, 0.0f, 1.0f);
  v3 one = v3_splat(1.0f);
  v3 fs_bar = v3_add(f0, v3_scale(v3_sub(one, f0), ora_pow5(1.0f - MU_BAR)));
  float denom = MU_BAR * ora_pow6(1.0f - MU_BAR);
  v3 a = v3_divs(v3_mul(fs_bar, v3_sub(one, tint)), denom);
  v3 fs_mu = v3_add(f0, v3_scale(v3_sub(one, f0), ora_pow5(1.0f - mu)));
  return v3_clamp(v3_sub(fs_mu, v3_scale(v3_scale(a, mu), ora_pow6(1.0f - mu))), v3_splat(0.0f), one);
}
float ora_fresnel_dielectric(float cos_i, float eta_i, float eta_t) { /* brdf.rs:230-240 */
  cos_i = ora_clamp(cos_i, 0.0f, 1.0f);
  float sin2_t = (eta_i / eta_t) * (eta_i / eta_t) * (1.0f - cos_i * cos_i);
  if (sin2_t >= 1.0f) return 1.0f;
  float cos_t = sqrtf(1.0f - sin2_t);
  float r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
  float r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
  return 0.5f * (r_par * r_par + r_perp * r_perp);
}
void ora_tangent_frame(v3 n, v3 *t, v3 *b) { /* brdf.rs:245-252 */
  float sign = n.z >= 0.0f ? 1.0f : -1.0f;
  float a = -1.0f / (sign + n.z);
  float bb = n.x * n.y * a;
  *t = v3_new(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
  *b = v3_new(bb, sign + n.y * n.y * a, -n.y);
}
static float sheen_charlie_d(float n_dot_h, float roughness) { /* brdf.rs:266-271 */
  float alpha = ora_max(roughness, 0.05f);
  float inv_alpha = 1.0f / alpha;
  float sin2 = ora_max(1.0f - n_dot_h * n_dot_h, 0.0f);
  return (2.0f + inv_alpha) * ora_powf(sin2, inv_alpha * 0.5f) / (2.0f * ORA_PI);
}
static float sheen_charlie_v(float n_dot_v, float n_dot_l) { /* brdf.rs:274-276 */
  return 1.0f / (4.0f * ora_max(n_dot_l + n_dot_v - n_dot_l * n_dot_v, 1e-4f));
}
static v3 coat_darkening_factor(v3 base_color, float coat_ior, float darkening) { /* brdf.rs:288-293 */
  float f_avg = ora_f0_from_ior(coat_ior) + (1.0f - ora_f0_from_ior(coat_ior)) * 0.05f;
  v3 one = v3_splat(1.0f);
  v3 dark = v3_div(base_color, v3_max(v3_sub(one, v3_scale(v3_sub(one, base_color), f_avg)), v3_splat(1e-4f)));
  return v3_add(v3_scale(one, 1.0f - darkening), v3_scale(dark, darkening));
}
static const float LAMBDA_RGB[3] = {615.0f, 545.0f, 465.0f}; /* brdf.rs:306 */
float ora_cauchy_ior(float n_d, float v_d, float lambda_nm) { /* brdf.rs:326-333 */
  const float C = 656.3f, D = 587.6f, F = 486.1f;
  float b = (n_d - 1.0f) / (v_d * (1.0f / (F * F) - 1.0f / (C * C)));
  float a = n_d - b / (D * D);
  return a + b / (lambda_nm * lambda_nm);
}
static float fresnel_amplitude(float eta_i, float eta_t, float cos_i, float cos_t) { /* brdf.rs:417-421 */
  float rs = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
  float rp = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
  return 0.5f * (rs + rp);
}
static float thin_film_lambda(float cos_theta_1, float eta_1, float eta_film, float eta_2, float thickness_nm,
                              float lambda_nm) { /* brdf.rs:337-373 */
  float cos1 = ora_clamp(cos_theta_1, 0.0f, 1.0f);
  float sin2_1 = 1.0f - cos1 * cos1;
  float sin2_film = ora_pow2(eta_1 / eta_film) * sin2_1;
  if (sin2_film >= 1.0f) return 1.0f;
  float cos_film = sqrtf(1.0f - sin2_film);
  float sin2_base = ora_pow2(eta_film / eta_2) * sin2_film;
  if (sin2_base >= 1.0f) return 1.0f;
  float cos_base = sqrtf(1.0f - sin2_base);
  float r_a = fresnel_amplitude(eta_1, eta_film, cos1, cos_film);
  float r_b = fresnel_amplitude(eta_film, eta_2, cos_film, cos_base);
  float opd = 2.0f * eta_film * thickness_nm * cos_film;
  float phi = 2.0f * ORA_PI * opd / lambda_nm;
  float cos_phi = ora_cosf(phi);
  float num = r_a * r_a + 2.0f * r_a * r_b * cos_phi + r_b * r_b;
  float den = 1.0f + 2.0f * r_a * r_b * cos_phi + ora_pow2(r_a * r_b);
  return ora_clamp(num / ora_max(den, 1e-8f), 0.0f, 1.0f);
}
v3 ora_thin_film_fresnel(float cos1, float eta1, float eta_film, float eta2, float thickness_nm) { /* brdf.rs:375-388 */
  v3 o;
  o.x = thin_film_lambda(cos1, eta1, eta_film, eta2, thickness_nm, LAMBDA_RGB[0]);
  o.y = thin_film_lambda(cos1, eta1, eta_film, eta2, thickness_nm, LAMBDA_RGB[1]);
  o.z = thin_film_lambda(cos1, eta1, eta_film, eta2, thickness_nm, LAMBDA_RGB[2]);
  return o;
}
static v3 thin_film_fresnel_metal(float cos1, float eta1, float eta_film, v3 f0, float thickness_nm) { /* brdf.rs:397-413 */
  v3 o = v3_splat(0.0f);
  for (int i = 0; i < 3; i++) {
    float f0_c = ora_clamp(v3_get(f0, i), 0.0f, 0.9999f);
    float sq = sqrtf(f0_c);
    float eta_2 = (1.0f + sq) / (1.0f - sq);
    v3_set(&o, i, thin_film_lambda(cos1, eta1, eta_film, eta_2, thickness_nm, LAMBDA_RGB[i]));
  }
  return o;
}

/* ------------------------------------------------------------------ */
/* material/openpbr.rs                                                 */
/* ------------------------------------------------------------------ */
static v3 c3(const float c[3]) { return v3_new(c[0], c[1], c[2]); }
static void set3(float d[3], float x, float y, float z) { d[0] = x; d[1] = y; d[2] = z; }

void ora_material_default(OraMaterial *m) { /* openpbr.rs:130-173 */
  memset(m, 0, sizeof *m);
  m->kind = ORA_MAT_OPENPBR;
  m->base_weight = 1.0f; set3(m->base_color, 0.8f, 0.8f, 0.8f);
  m->specular_weight = 1.0f; set3(m->specular_color, 1, 1, 1); m->specular_roughness = 0.3f; m->specular_ior = 1.5f;
  set3(m->transmission_color, 1, 1, 1); m->transmission_dispersion_abbe_number = 20.0f;
  set3(m->subsurface_color, 0.8f, 0.8f, 0.8f); m->subsurface_radius = 1.0f;
  set3(m->subsurface_radius_scale, 1.0f, 0.5f, 0.25f);
  set3(m->fuzz_color, 1, 1, 1); m->fuzz_roughness = 0.5f;
  set3(m->coat_color, 1, 1, 1); m->coat_ior = 1.6f; m->coat_darkening = 1.0f;
  m->thin_film_thickness = 0.5f; m->thin_film_ior = 1.4f;
  set3(m->emission_color, 1, 1, 1);
  m->geometry_opacity = 1.0f;
}
void ora_material_diffuse(OraMaterial *m, float r, float g, float b) { /* openpbr.rs:177-183 */
  ora_material_default(m);
  set3(m->base_color, r, g, b);
  m->specular_weight = 0.0f;
}
void ora_material_emissive(OraMaterial *m, float r, float g, float b) { /* emissive.rs:16-19 */
  ora_material_default(m);
  m->kind = ORA_MAT_EMISSIVE;
  set3(m->emission_color, r, g, b);
  m->emission_luminance = 1.0f;
}

typedef struct { float p_diffuse, p_specular, p_coat, p_fuzz, p_transmission; } LobePmf;
static float luma(v3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; } /* openpbr.rs:402-404 */

static LobePmf lobe_pmf(const OraMaterial *m) { /* openpbr.rs:317-378 */
  float f0_diel = ora_f0_from_ior(m->specular_ior);
  float f0_coat = ora_f0_from_ior(m->coat_ior);
  float base_luma = ora_max(luma(c3(m->base_color)), 0.02f);
  float spec_luma = ora_max(luma(c3(m->specular_color)), 0.02f);
  float fuzz_luma = ora_max(luma(c3(m->fuzz_color)), 0.02f);
  float w_metal = m->base_metalness * m->specular_weight * ora_max(luma(v3_scale(c3(m->base_color), m->base_weight)), 0.02f);
  float w_diel_spec = (1.0f - m->base_metalness) * m->specular_weight * spec_luma * f0_diel;
  float w_specular = ora_max(w_metal + w_diel_spec, 1e-4f);
  float w_diffuse = ora_max((1.0f - m->base_metalness) * (1.0f - m->transmission_weight) * m->base_weight * base_luma *
                                (1.0f - f0_diel), 1e-4f);
  float w_coat = ora_max(m->coat_weight * f0_coat, 1e-6f);
  float w_fuzz = ora_max(m->fuzz_weight * fuzz_luma, 1e-6f);
  float trans_luma = ora_max(luma(c3(m->transmission_color)), 0.02f);
  float w_transmission = m->transmission_weight > 0.0f
                             ? ora_max((1.0f - m->base_metalness) * m->transmission_weight * trans_luma, 1e-4f)
                             : 0.0f;
  float total = w_diffuse + w_specular + w_coat + w_fuzz + w_transmission;
  LobePmf p = {w_diffuse / total, w_specular / total, w_coat / total, w_fuzz / total, w_transmission / total};
  return p;
}
void ora_lobe_pmf(const OraMaterial *m, float out[5]) {
  LobePmf p = lobe_pmf(m);
  out[0] = p.p_diffuse; out[1] = p.p_specular; out[2] = p.p_coat; out[3] = p.p_fuzz; out[4] = p.p_transmission;
}
enum { LOBE_DIFFUSE, LOBE_SPECULAR, LOBE_COAT, LOBE_FUZZ, LOBE_TRANSMISSION };
static int lobe_pick(const LobePmf *p, float u) { /* openpbr.rs:380-398 */
  float acc = p->p_diffuse;
  if (u < acc) return LOBE_DIFFUSE;
  acc += p->p_specular;
  if (u < acc) return LOBE_SPECULAR;
  acc += p->p_coat;
  if (u < acc) return LOBE_COAT;
  acc += p->p_fuzz;
  if (u < acc) return LOBE_FUZZ;
  return LOBE_TRANSMISSION;
}

static v3 eval_diffuse(const OraMaterial *m, v3 v_local, v3 l_local, float f_avg_diel) { /* openpbr.rs:410-446 */
  if (l_local.z <= 0.0f || v_local.z <= 0.0f) return v3_splat(0.0f);
  float presence = m->base_weight * (1.0f - m->base_metalness) * (1.0f - m->transmission_weight);
  if (presence <= 0.0f) return v3_splat(0.0f);
  v3 diffuse_color = v3_lerp(c3(m->base_color), c3(m->subsurface_color), m->subsurface_weight);
  v3 rho = v3_scale(diffuse_color, presence);
  return v3_scale(ora_eon_diffuse(rho, m->base_diffuse_roughness, v_local, l_local), 1.0f - f_avg_diel);
}

static v3 eval_specular(const OraMaterial *m, v3 v_local, v3 l_local, v3 h_local, float ax, float ay) { /* openpbr.rs:448-551 */
  float n_dot_v = ora_max(v_local.z, 1e-4f);
  float n_dot_l = ora_max(l_local.z, 1e-4f);
  float n_dot_h = ora_max(h_local.z, 1e-4f);
  float v_dot_h = ora_max(v3_dot(v_local, h_local), 1e-4f);
  float d = ggx_d_aniso(n_dot_h, h_local.x, h_local.y, ax, ay);
  float g = ggx_g2_smith_aniso(n_dot_v, v_local.x, v_local.y, n_dot_l, l_local.x, l_local.y, ax, ay);
  float f0_diel_scalar = ora_f0_from_ior(m->specular_ior);
  float outer_ior = m->coat_weight > 0.0f ? m->coat_ior : 1.0f;
  float tf_thickness_nm = m->thin_film_thickness * 1000.0f;
  v3 diel_term = v3_splat(0.0f);
  if (m->base_metalness < 1.0f) {
    v3 f0_diel_base = v3_scale(v3_scale(c3(m->specular_color), f0_diel_scalar), m->specular_weight);
    v3 f_diel;
    if (m->thin_film_weight > 0.0f) {
      v3 f_normal = fresnel_schlick(v_dot_h, f0_diel_base);
      v3 f_iri = ora_thin_film_fresnel(v_dot_h, outer_ior, m->thin_film_ior, m->specular_ior, tf_thickness_nm);
      f_diel = v3_add(v3_scale(f_normal, 1.0f - m->thin_film_weight), v3_scale(f_iri, m->thin_film_weight));
    } else {
      f_diel = fresnel_schlick(v_dot_h, f0_diel_base);
    }
    if (m->thin_walled && m->transmission_weight > 0.0f) {
      float f_phys = fresnel_schlick_scalar(v_dot_h, f0_diel_scalar);
      float boost = 2.0f / (1.0f + f_phys);
      f_diel = v3_scale(f_diel, 1.0f + (boost - 1.0f) * m->transmission_weight);
    }
    diel_term = v3_scale(f_diel, 1.0f - m->base_metalness);
  }
  v3 metal_term = v3_splat(0.0f);
  if (m->base_metalness > 0.0f) {
    v3 metal_f0 = v3_scale(c3(m->base_color), m->base_weight);
    v3 f_metal_base = ora_fresnel_f82_tint(v_dot_h, metal_f0, c3(m->specular_color));
    v3 f_metal;
    if (m->thin_film_weight > 0.0f) {
      v3 f_iri = thin_film_fresnel_metal(v_dot_h, outer_ior, m->thin_film_ior, metal_f0, tf_thickness_nm);
      f_metal = v3_add(v3_scale(f_metal_base, 1.0f - m->thin_film_weight), v3_scale(f_iri, m->thin_film_weight));
    } else {
      f_metal = f_metal_base;
    }
    f_metal = v3_scale(f_metal, m->specular_weight);
    metal_term = v3_scale(f_metal, m->base_metalness);
  }
  float brdf = d * g / (4.0f * n_dot_v * n_dot_l);
  return v3_scale(v3_add(metal_term, diel_term), brdf);
}

static v3 eval_coat(const OraMaterial *m, v3 v_local, v3 l_local, v3 h_local, float ax, float ay) { /* openpbr.rs:553-576 */
  float n_dot_v = ora_max(v_local.z, 1e-4f);
  float n_dot_l = ora_max(l_local.z, 1e-4f);
  float n_dot_h = ora_max(h_local.z, 1e-4f);
  float v_dot_h = ora_max(v3_dot(v_local, h_local), 1e-4f);
  float d = ggx_d_aniso(n_dot_h, h_local.x, h_local.y, ax, ay);
  float g = ggx_g2_smith_aniso(n_dot_v, v_local.x, v_local.y, n_dot_l, l_local.x, l_local.y, ax, ay);
  float f = fresnel_schlick_scalar(v_dot_h, ora_f0_from_ior(m->coat_ior));
  float brdf = d * g / (4.0f * n_dot_v * n_dot_l);
  return v3_splat(m->coat_weight * f * brdf);
}

v3 ora_coat_passage(const OraMaterial *m, float cos_theta) { /* openpbr.rs:587-603 */
  float cos_i = ora_clamp(cos_theta, 1e-4f, 1.0f);
  float eta = ora_max(m->coat_ior, 1e-4f);
  float sin2_t = (1.0f - cos_i * cos_i) / (eta * eta);
  float cos_t = sqrtf(ora_max(1.0f - sin2_t, 0.0f));
  float path_length = 1.0f / ora_max(cos_t, 1e-3f);
  v3 cc = v3_clamp(c3(m->coat_color), v3_splat(0.0f), v3_splat(1.0f));
  float e = 0.5f * path_length;
  v3 one_passage = v3_new(ora_powf(cc.x, e), ora_powf(cc.y, e), ora_powf(cc.z, e));
  v3 absorb = v3_lerp(v3_splat(1.0f), one_passage, m->coat_weight);
  float f_coat = fresnel_schlick_scalar(cos_i, ora_f0_from_ior(m->coat_ior));
  return v3_scale(absorb, 1.0f - m->coat_weight * f_coat);
}
static v3 coat_attenuation(const OraMaterial *m, float cos_v, float cos_l) { /* openpbr.rs:614-620 */
  if (m->coat_weight <= 0.0f) return v3_splat(1.0f);
  v3 dark = coat_darkening_factor(c3(m->base_color), m->coat_ior, m->coat_darkening);
  return v3_mul(v3_mul(ora_coat_passage(m, cos_v), ora_coat_passage(m, cos_l)), dark);
}
static v3 eval_fuzz(const OraMaterial *m, v3 v_local, v3 l_local, v3 h_local) { /* openpbr.rs:622-627 */
  float n_dot_v = ora_max(v_local.z, 1e-4f);
  float n_dot_l = ora_max(l_local.z, 1e-4f);
  float n_dot_h = ora_max(h_local.z, 0.0f);
  float sc = sheen_charlie_d(n_dot_h, m->fuzz_roughness) * sheen_charlie_v(n_dot_v, n_dot_l); /* brdf.rs:279-281 */
  return v3_scale(v3_scale(c3(m->fuzz_color), m->fuzz_weight), sc);
}

static int transmission_is_continuous(const OraMaterial *m) { /* openpbr.rs:827-829 */
  return m->transmission_weight > 0.0f && !m->thin_walled;
}
v3 ora_dispersive_ior(float n_d, float abbe, float scale) { /* openpbr.rs:736-751 */
  if (scale <= 0.0f || n_d == 1.0f) return v3_splat(n_d);
  int inverted = n_d < 1.0f;
  float n_above_one = inverted ? 1.0f / n_d : n_d;
  float v_d = ora_max(ora_max(abbe, 1.0f) / scale, 1.0f);
  v3 o = v3_splat(0.0f);
  for (int c = 0; c < 3; c++) {
    float n = ora_cauchy_ior(n_above_one, v_d, LAMBDA_RGB[c]);
    v3_set(&o, c, inverted ? 1.0f / n : n);
  }
  return o;
}
static v3 transmission_iors(const OraMaterial *m) { /* openpbr.rs:833-839 */
  return ora_dispersive_ior(m->specular_ior, m->transmission_dispersion_abbe_number, m->transmission_dispersion_scale);
}
static void transmission_alphas(const OraMaterial *m, float *ax, float *ay) { /* openpbr.rs:850-855 */
  ora_roughness_to_alpha_aniso(ora_max(m->specular_roughness, 0.01f), m->specular_roughness_anisotropy, ax, ay);
}
static void eval_transmission_channel(const OraMaterial *m, v3 v_local, v3 l_local, int entering, float ior,
                                      float *btdf_o, float *pdf_o) { /* openpbr.rs:862-914 */
  *btdf_o = 0.0f; *pdf_o = 0.0f;
  float eta_i = entering ? 1.0f : ior, eta_t = entering ? ior : 1.0f; /* openpbr.rs:844-846 */
  v3 h = v3_neg(v3_add(v3_scale(v_local, eta_i), v3_scale(l_local, eta_t)));
  if (v3_len2(h) < 1e-12f) return;
  h = v3_normalize(h);
  if (h.z < 0.0f) h = v3_neg(h);
  float v_dot_h = v3_dot(v_local, h);
  float l_dot_h = v3_dot(l_local, h);
  if (v_dot_h <= 1e-6f || l_dot_h >= -1e-6f) return;
  float ax, ay;
  transmission_alphas(m, &ax, &ay);
  float n_dot_v = ora_max(v_local.z, 1e-6f);
  float n_dot_l = ora_max(-l_local.z, 1e-6f);
  float d = ggx_d_aniso(ora_max(h.z, 1e-6f), h.x, h.y, ax, ay);
  float g = ggx_g2_smith_aniso(n_dot_v, v_local.x, v_local.y, n_dot_l, l_local.x, l_local.y, ax, ay);
  float f = ora_fresnel_dielectric(v_dot_h, eta_i, eta_t);
  float denom = eta_i * v_dot_h + eta_t * l_dot_h;
  float denom2 = denom * denom;
  if (denom2 < 1e-10f) return;
  float btdf = (v_dot_h * -l_dot_h) / (n_dot_v * n_dot_l) * (eta_t * eta_t * (1.0f - f) * d * g / denom2);
  float p_h = pdf_vndf_h_aniso_local(v_local, h, ax, ay);
  float jacobian = eta_t * eta_t * -l_dot_h / denom2;
  *btdf_o = ora_max(btdf, 0.0f);
  *pdf_o = p_h * jacobian;
}
static void eval_transmission(const OraMaterial *m, v3 v_local, v3 l_local, int entering, v3 *value, float *pdf) {
  /* openpbr.rs:924-949 */
  v3 color = m->transmission_depth > 0.0f ? v3_splat(1.0f) : c3(m->transmission_color);
  v3 tint = v3_scale(color, m->transmission_weight * (1.0f - m->base_metalness));
  v3 iors = transmission_iors(m);
  if (m->transmission_dispersion_scale <= 0.0f) {
    float btdf, p;
    eval_transmission_channel(m, v_local, l_local, entering, iors.y, &btdf, &p);
    *value = v3_scale(tint, btdf); *pdf = p;
    return;
  }
  v3 val = v3_splat(0.0f);
  float acc = 0.0f;
  for (int c = 0; c < 3; c++) {
    float btdf, p;
    eval_transmission_channel(m, v_local, l_local, entering, v3_get(iors, c), &btdf, &p);
    v3_set(&val, c, btdf);
    acc += p / 3.0f;
  }
  *value = v3_mul(tint, val); *pdf = acc;
}

v3 ora_eval_all(const OraMaterial *m, v3 v_local, v3 l_local, int entering) { /* openpbr.rs:629-683 */
  if (v_local.z <= 0.0f) return v3_splat(0.0f);
  if (l_local.z <= 0.0f) {
    if (!transmission_is_continuous(m)) return v3_splat(0.0f);
    v3 val; float p;
    eval_transmission(m, v_local, l_local, entering, &val, &p);
    return val;
  }
  v3 h_local = v3_normalize(v3_add(v_local, l_local));
  float ax, ay;
  ora_roughness_to_alpha_aniso(m->specular_roughness, m->specular_roughness_anisotropy, &ax, &ay);
  float f_avg_diel = ora_f0_from_ior(m->specular_ior);
  v3 diffuse = eval_diffuse(m, v_local, l_local, f_avg_diel);
  v3 specular = eval_specular(m, v_local, l_local, h_local, ax, ay);
  v3 coat = v3_splat(0.0f);
  if (m->coat_weight > 0.0f) {
    float axc, ayc;
    ora_roughness_to_alpha_aniso(m->coat_roughness, m->coat_roughness_anisotropy, &axc, &ayc);
    coat = eval_coat(m, v_local, l_local, h_local, axc, ayc);
  }
  v3 fuzz = m->fuzz_weight > 0.0f ? eval_fuzz(m, v_local, l_local, h_local) : v3_splat(0.0f);
  v3 coat_atten = coat_attenuation(m, v_local.z, l_local.z);
  float base_atten = ora_clamp(1.0f - m->fuzz_weight, 0.0f, 1.0f);
  return v3_add(fuzz, v3_scale(v3_add(coat, v3_mul(coat_atten, v3_add(diffuse, specular))), base_atten));
}

static float pdf_all(const OraMaterial *m, const LobePmf *pmf, v3 v_local, v3 l_local, int entering) { /* openpbr.rs:689-722 */
  if (v_local.z <= 0.0f) return 0.0f;
  if (l_local.z <= 0.0f) {
    if (!transmission_is_continuous(m)) return 0.0f;
    v3 val; float p;
    eval_transmission(m, v_local, l_local, entering, &val, &p);
    return pmf->p_transmission * p;
  }
  v3 h_local = v3_normalize(v3_add(v_local, l_local));
  float ax, ay, axc, ayc;
  ora_roughness_to_alpha_aniso(m->specular_roughness, m->specular_roughness_anisotropy, &ax, &ay);
  ora_roughness_to_alpha_aniso(m->coat_roughness, m->coat_roughness_anisotropy, &axc, &ayc);
  float pdf_cosine = ora_max(l_local.z, 0.0f) / ORA_PI;
  float pdf_specular = pdf_vndf_ggx_aniso_local(v_local, h_local, ax, ay);
  float pdf_coat = pdf_vndf_ggx_aniso_local(v_local, h_local, axc, ayc);
  return pmf->p_diffuse * pdf_cosine + pmf->p_specular * pdf_specular + pmf->p_coat * pdf_coat + pmf->p_fuzz * pdf_cosine;
}
float ora_pdf_all(const OraMaterial *m, v3 v_local, v3 l_local, int entering) {
  LobePmf p = lobe_pmf(m);
  return pdf_all(m, &p, v_local, l_local, entering);
}

static int sample_transmission_rough(const OraMaterial *m, v3 v_local, int entering, float dispersion_u, float u1,
                                     float u2, v3 *l_out) { /* openpbr.rs:958-993 */
  v3 iors = transmission_iors(m);
  float ior;
  if (m->transmission_dispersion_scale > 0.0f) {
    if (dispersion_u < 1.0f / 3.0f) ior = iors.x;
    else if (dispersion_u < 2.0f / 3.0f) ior = iors.y;
    else ior = iors.z;
  } else ior = iors.y;
  float eta_i = entering ? 1.0f : ior, eta_t = entering ? ior : 1.0f;
  float eta_rel = eta_i / eta_t;
  float ax, ay;
  transmission_alphas(m, &ax, &ay);
  v3 h = ora_sample_vndf(v_local, ax, ay, u1, u2);
  float cos_i = v3_dot(v_local, h);
  if (cos_i <= 1e-6f) return 0;
  float sin2_t = eta_rel * eta_rel * (1.0f - cos_i * cos_i);
  if (sin2_t >= 1.0f) return 0;
  float cos_t = sqrtf(1.0f - sin2_t);
  v3 l = v3_normalize(v3_add(v3_scale(v3_neg(v_local), eta_rel), v3_scale(h, eta_rel * cos_i - cos_t)));
  if (l.z >= -1e-6f) return 0;
  *l_out = l;
  return 1;
}

/* openpbr.rs:779-812 */
static void sample_transmission_thin(const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec, v3 *dir_o,
                                     v3 *throughput) {
  v3 dir = v3_normalize(ray_dir);
  *dir_o = dir;
  float cos_i = ora_clamp(v3_dot(v3_neg(dir), rec->normal), 0.0f, 1.0f);
  float eta = ora_max(m->specular_ior, 1e-4f);
  float sin2_t = (1.0f - cos_i * cos_i) / (eta * eta);
  if (sin2_t >= 1.0f) { *throughput = v3_splat(0.0f); return; }
  float cos_t = sqrtf(1.0f - sin2_t);
  float f = ora_fresnel_dielectric(cos_i, 1.0f, eta);
  float window_transmittance = (1.0f - f) / (1.0f + f);
  float path_length = 1.0f / ora_max(cos_t, 1e-4f);
  v3 tc = v3_clamp(c3(m->transmission_color), v3_splat(0.0f), v3_splat(1.0f));
  v3 tint = v3_new(ora_powf(tc.x, path_length), ora_powf(tc.y, path_length), ora_powf(tc.z, path_length));
  *throughput = v3_scale(tint, window_transmittance * m->transmission_weight);
}

typedef struct { v3 n, t, b; } Frame; /* openpbr.rs:268-285 */
static Frame frame_new(v3 n) { Frame f; f.n = n; ora_tangent_frame(n, &f.t, &f.b); return f; }
static v3 frame_to_local(const Frame *f, v3 v) { return v3_new(v3_dot(v, f->t), v3_dot(v, f->b), v3_dot(v, f->n)); }
static v3 frame_to_world(const Frame *f, v3 l) {
  return v3_add(v3_add(v3_scale(f->t, l.x), v3_scale(f->b, l.y)), v3_scale(f->n, l.z));
}

int ora_mat_scatter(const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec, OraSampler dom, OraScatter *out) {
  if (m->kind == ORA_MAT_EMISSIVE) return 0; /* emissive.rs:30-38 */
  out->medium = 0;
  /* openpbr.rs:1026-1136 scatter_resolved */
  Frame frame = frame_new(rec->normal);
  v3 v_world = v3_neg(v3_normalize(ray_dir));
  v3 v_local = frame_to_local(&frame, v_world);
  if (v_local.z <= 0.0f) return 0;
  float s[4];
  ora_draw_sample4(dom, s);
  LobePmf pmf = lobe_pmf(m);
  int lobe = lobe_pick(&pmf, s[0]);
  if (lobe == LOBE_TRANSMISSION) {
    if (transmission_is_continuous(m)) {
      v3 l_local;
      if (!sample_transmission_rough(m, v_local, rec->front_face, s[3], s[1], s[2], &l_local)) return 0;
      v3 l_world = frame_to_world(&frame, l_local);
      float pdf = ora_max(pdf_all(m, &pmf, v_local, l_local, rec->front_face), 1e-4f);
      v3 brdf = ora_eval_all(m, v_local, l_local, rec->front_face);
      out->origin = v3_add(rec->p, v3_scale(l_world, 1e-4f));
      out->dir = l_world;
      out->value = v3_scale(brdf, ora_abs(l_local.z));
      out->pdf = pdf;
      out->delta = 0;
      OraMedium med; /* openpbr.rs:1061-1066: only a ray refracting INTO the front face carries the interior */
      out->medium = (rec->front_face && ora_interior_medium(m, &med)) ? 1 : 0;
      return 1;
    }
    v3 dir, throughput;
    sample_transmission_thin(m, ray_dir, rec, &dir, &throughput);
    float p_select = ora_max(pmf.p_transmission, 1e-4f);
    out->origin = rec->p; out->dir = dir;
    out->value = v3_divs(throughput, p_select);
    out->pdf = 1.0f; out->delta = 1;
    return 1;
  }
  v3 l_local;
  if (lobe == LOBE_DIFFUSE || lobe == LOBE_FUZZ) {
    l_local = ora_cosine_hemisphere(s[1], s[2]);
  } else {
    float ax, ay;
    if (lobe == LOBE_SPECULAR) ora_roughness_to_alpha_aniso(m->specular_roughness, m->specular_roughness_anisotropy, &ax, &ay);
    else ora_roughness_to_alpha_aniso(m->coat_roughness, m->coat_roughness_anisotropy, &ax, &ay);
    v3 h_local = ora_sample_vndf(v_local, ax, ay, s[1], s[2]);
    v3 l = v3_sub(v3_scale(h_local, 2.0f * v3_dot(v_local, h_local)), v_local);
    if (l.z <= 0.0f) return 0;
    l_local = l;
  }
  float pdf = ora_max(pdf_all(m, &pmf, v_local, l_local, rec->front_face), 1e-4f);
  v3 brdf = ora_eval_all(m, v_local, l_local, rec->front_face);
  float n_dot_l = ora_max(l_local.z, 0.0f);
  out->origin = rec->p;
  out->dir = frame_to_world(&frame, l_local);
  out->value = v3_scale(brdf, n_dot_l);
  out->pdf = pdf;
  out->delta = 0;
  return 1;
}

int ora_mat_eval(const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec, v3 wi, v3 *value, float *pdf) {
  if (m->kind == ORA_MAT_EMISSIVE) return 0; /* material.rs:71-74 default: None */
  /* openpbr.rs:1138-1158 eval_resolved */
  Frame frame = frame_new(rec->normal);
  v3 v_local = frame_to_local(&frame, v3_neg(v3_normalize(ray_dir)));
  if (v_local.z <= 0.0f) return 0;
  v3 l_local = frame_to_local(&frame, v3_normalize(wi));
  LobePmf pmf = lobe_pmf(m);
  *pdf = ora_max(pdf_all(m, &pmf, v_local, l_local, rec->front_face), 1e-4f);
  *value = v3_scale(ora_eval_all(m, v_local, l_local, rec->front_face), ora_abs(l_local.z));
  return 1;
}

v3 ora_mat_emitted(const OraMaterial *m) {
  if (m->kind == ORA_MAT_EMISSIVE) return c3(m->emission_color); /* emissive.rs:25-28 */
  return v3_scale(c3(m->emission_color), m->emission_luminance); /* openpbr.rs:1202-1204 */
}
v3 ora_mat_emitted_directional(const OraMaterial *m, float cos_theta_o) {
  if (m->kind == ORA_MAT_EMISSIVE) return c3(m->emission_color); /* material.rs:112-115 */
  v3 uncoated = v3_scale(c3(m->emission_color), m->emission_luminance); /* openpbr.rs:1211-1218 */
  if (m->coat_weight <= 0.0f) return uncoated;
  v3 dark = coat_darkening_factor(c3(m->base_color), m->coat_ior, m->coat_darkening);
  return v3_mul(v3_mul(uncoated, ora_coat_passage(m, cos_theta_o)), dark);
}

/* ------------------------------------------------------------------ */
/* medium.rs — carried interior medium                                 */
/* ------------------------------------------------------------------ */
OraMedium ora_medium_from_transmission(v3 tint, float depth, v3 scatter, float anisotropy) { /* medium.rs:40-66 */
  OraMedium m;
  if (depth <= 1e-6f) { m.sigma_a = v3_splat(0.0f); m.sigma_s = v3_splat(0.0f); m.g = 0.0f; return m; }
  v3 t = v3_clamp(tint, v3_splat(1e-4f), v3_splat(1.0f));
  v3 extinction = v3_divs(v3_new(-ora_logf(t.x), -ora_logf(t.y), -ora_logf(t.z)), depth);
  v3 sigma_s = v3_divs(v3_max(scatter, v3_splat(0.0f)), depth);
  v3 sigma_a = v3_sub(extinction, sigma_s);
  float mn = v3_min_elem(sigma_a);
  if (mn < 0.0f) sigma_a = v3_sub(sigma_a, v3_splat(mn));
  m.sigma_a = sigma_a; m.sigma_s = sigma_s; m.g = ora_clamp(anisotropy, -0.999f, 0.999f);
  return m;
}
OraMedium ora_medium_from_subsurface(v3 albedo, float radius, v3 radius_scale, float g) { /* medium.rs:77-97 */
  v3 mfp = v3_max(v3_mul(v3_splat(radius), radius_scale), v3_splat(1e-3f));
  v3 sigma_t = v3_div(v3_splat(1.0f), mfp);
  g = ora_clamp(g, -0.999f, 0.999f);
  v3 a = v3_clamp(albedo, v3_splat(0.0f), v3_splat(1.0f));
  /* inner = 9.59217 + 41.6808*a + 17.7126*a*a, left to right as glam evaluates it */
  v3 inner = v3_add(v3_add(v3_splat(9.59217f), v3_scale(a, 41.6808f)), v3_mul(v3_scale(a, 17.7126f), a));
  v3 sqrt_inner = v3_new(sqrtf(inner.x), sqrtf(inner.y), sqrtf(inner.z));
  v3 s = v3_sub(v3_add(v3_splat(4.09712f), v3_scale(a, 4.20863f)), sqrt_inner);
  v3 s2 = v3_mul(s, s);
  v3 alpha_ss = v3_clamp(v3_div(v3_sub(v3_splat(1.0f), s2), v3_sub(v3_splat(1.0f), v3_scale(s2, g))), v3_splat(0.0f),
                         v3_splat(1.0f));
  OraMedium m;
  m.sigma_s = v3_mul(sigma_t, alpha_ss);
  m.sigma_a = v3_sub(sigma_t, m.sigma_s);
  m.g = g;
  return m;
}
OraMedium ora_medium_blend(const OraMedium *a, float wa, const OraMedium *b, float wb) { /* medium.rs:103-114 */
  OraMedium m;
  m.sigma_a = v3_add(v3_scale(a->sigma_a, wa), v3_scale(b->sigma_a, wb));
  m.sigma_s = v3_add(v3_scale(a->sigma_s, wa), v3_scale(b->sigma_s, wb));
  float sa = (((a->sigma_s.x + a->sigma_s.y) + a->sigma_s.z) / 3.0f) * wa;
  float sb = (((b->sigma_s.x + b->sigma_s.y) + b->sigma_s.z) / 3.0f) * wb;
  m.g = (sa + sb > 1e-8f) ? (a->g * sa + b->g * sb) / (sa + sb) : 0.0f;
  return m;
}
v3 ora_medium_transmittance(const OraMedium *m, float t) { /* medium.rs:117-120 */
  v3 e = v3_scale(v3_add(m->sigma_a, m->sigma_s), t);
  return v3_new(ora_expf(-e.x), ora_expf(-e.y), ora_expf(-e.z));
}
int ora_medium_is_scattering(const OraMedium *m) { return v3_max_elem(m->sigma_s) > 1e-6f; }
float ora_medium_sigma_t_max(const OraMedium *m) { return v3_max_elem(v3_add(m->sigma_a, m->sigma_s)); }
v3 ora_medium_albedo(const OraMedium *m) {
  v3 denom = v3_max(v3_add(m->sigma_a, m->sigma_s), v3_splat(1e-6f));
  return v3_div(m->sigma_s, denom);
}
int ora_interior_medium(const OraMaterial *m, OraMedium *out) { /* openpbr.rs:225-258 */
  if (m->kind == ORA_MAT_EMISSIVE) return 0;
  float trans_frac = m->transmission_weight;
  float sss_frac = (1.0f - m->transmission_weight) * m->subsurface_weight;
  float total = trans_frac + sss_frac;
  if (total <= 0.0f) return 0;
  OraMedium trans_volume = ora_medium_from_transmission(c3(m->transmission_color), m->transmission_depth,
                                                        c3(m->transmission_scatter), m->transmission_scatter_anisotropy);
  OraMedium medium;
  if (sss_frac > 0.0f) {
    OraMedium sss_volume = ora_medium_from_subsurface(c3(m->subsurface_color), m->subsurface_radius,
                                                      c3(m->subsurface_radius_scale), m->subsurface_scatter_anisotropy);
    medium = ora_medium_blend(&trans_volume, trans_frac / total, &sss_volume, sss_frac / total);
  } else {
    medium = trans_volume;
  }
  if (ora_medium_sigma_t_max(&medium) <= 1e-6f) return 0;
  *out = medium;
  return 1;
}
float ora_hg_phase(float cos_theta, float g) { /* medium.rs:148-152 */
  float denom = ora_max(1.0f + g * g - 2.0f * g * cos_theta, 1e-6f);
  return (1.0f - g * g) / (4.0f * ORA_PI * denom * sqrtf(denom));
}
v3 ora_sample_henyey_greenstein(v3 wi, float g, float u1, float u2) { /* medium.rs:158-184 */
  float cos_theta;
  if (ora_abs(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u1;
  else {
    float sq = (1.0f - g * g) / (1.0f - g + 2.0f * g * u1);
    cos_theta = (1.0f + g * g - sq * sq) / (2.0f * g);
  }
  cos_theta = ora_clamp(cos_theta, -1.0f, 1.0f);
  float sin_theta = sqrtf(ora_max(1.0f - cos_theta * cos_theta, 0.0f));
  float phi = 2.0f * ORA_PI * u2;
  v3 up = ora_abs(wi.z) < 0.999f ? v3_new(0.0f, 0.0f, 1.0f) : v3_new(1.0f, 0.0f, 0.0f);
  v3 t = v3_normalize(v3_cross(wi, up));
  v3 b = v3_cross(wi, t);
  float sp, cp;
  ora_sincosf(phi, &sp, &cp);
  return v3_normalize(v3_add(v3_add(v3_scale(t, sin_theta * cp), v3_scale(b, sin_theta * sp)), v3_scale(wi, cos_theta)));
}

/* ------------------------------------------------------------------ */
/* light.rs — area lights                                              */
/* ------------------------------------------------------------------ */
static v3 light_sample_point(const OraLight *l, float u, float v) {
  if (l->kind == ORA_LIGHT_SPHERE) { /* light.rs:27-36 */
    float theta = 2.0f * ORA_PI * u;
    float phi = ora_acosf(1.0f - 2.0f * v);
    float sp, cp, st, ct;
    ora_sincosf(phi, &sp, &cp);
    ora_sincosf(theta, &st, &ct);
    v3 n = v3_new(sp * ct, sp * st, cp);
    return v3_add(c3(l->center), v3_scale(n, l->radius));
  }
  return v3_add(v3_add(c3(l->origin), v3_scale(c3(l->edge_u), u)), v3_scale(c3(l->edge_v), v)); /* light.rs:70-72 */
}
static v3 light_normal_at(const OraLight *l, v3 p) {
  if (l->kind == ORA_LIGHT_SPHERE) return v3_normalize(v3_sub(p, c3(l->center))); /* light.rs:38-40 */
  return c3(l->normal);
}
static float light_area(const OraLight *l) {
  if (l->kind == ORA_LIGHT_SPHERE) return 4.0f * ORA_PI * l->radius * l->radius; /* light.rs:42-44 */
  return v3_len(v3_cross(c3(l->edge_u), c3(l->edge_v)));                         /* light.rs:78-80 */
}
static float solid_angle_pdf(const OraLight *l, v3 from, v3 light_point) { /* light.rs:180-187 */
  v3 direction = v3_sub(light_point, from);
  float d2 = v3_len2(direction);
  v3 dir_to_light = v3_normalize(direction);
  v3 ln = light_normal_at(l, light_point);
  float cosine = ora_max(v3_dot(ln, v3_neg(dir_to_light)), 0.0f);
  return d2 / (cosine * light_area(l) + 1e-4f);
}
v3 ora_align_to_normal(v3 local, v3 normal) { /* common.rs:176-188 */
  v3 up = ora_abs(normal.z) < 0.999f ? v3_new(0.0f, 0.0f, 1.0f) : v3_new(1.0f, 0.0f, 0.0f);
  v3 tangent = v3_normalize(v3_cross(normal, up));
  v3 bitangent = v3_cross(normal, tangent);
  return v3_add(v3_add(v3_scale(tangent, local.x), v3_scale(bitangent, local.y)), v3_scale(normal, local.z));
}
void ora_light_distant(OraLight *l, v3 direction, v3 irradiance, float angle_deg) { /* light.rs:255-266 */
  memset(l, 0, sizeof *l);
  l->kind = ORA_LIGHT_DISTANT; l->geom_id = 0xFFFFFFFFu;
  float diameter = ora_clamp(angle_deg, 0.05f, 179.0f);
  float half_angle = 0.5f * (diameter * (ORA_PI / 180.0f));
  float cos_half = ora_cosf(half_angle);
  v3 d = v3_normalize(direction);
  set3(l->normal, d.x, d.y, d.z);
  set3(l->radiance, irradiance.x, irradiance.y, irradiance.z);
  l->radius = cos_half;
  l->center[0] = 2.0f * ORA_PI * (1.0f - cos_half);
}
int ora_light_escaped(const OraLight *l, v3 direction, v3 *radiance, float *pdf) {
  if (l->kind == ORA_LIGHT_DISTANT) { /* light.rs:268-282, :300-303 */
    if (!(v3_dot(direction, v3_neg(c3(l->normal))) >= l->radius)) return 0;
    float omega = ora_max(l->center[0], 1e-12f);
    *radiance = v3_divs(c3(l->radiance), omega);
    *pdf = 1.0f / omega;
    return 1;
  }
  if (l->kind == ORA_LIGHT_DOME) { /* light.rs:340-355, :385-388 (no map: uniform tint, uniform sphere pdf) */
    *radiance = c3(l->radiance);
    *pdf = 1.0f / (4.0f * ORA_PI);
    return 1;
  }
  return 0; /* light.rs:141-146: lights with geometry are found by hitting them */
}
int ora_light_sample_li(const OraLight *l, v3 from, float u, float v, OraLightSample *out) { /* light.rs:191-204 */
  if (l->kind == ORA_LIGHT_DISTANT) { /* light.rs:285-298: uniform direction within the cone around -direction */
    float cos_theta = 1.0f - u * (1.0f - l->radius);
    float sin_theta = sqrtf(ora_max(1.0f - cos_theta * cos_theta, 0.0f));
    float phi = 2.0f * ORA_PI * v;
    float sp, cp;
    ora_sincosf(phi, &sp, &cp);
    v3 local = v3_new(sin_theta * cp, sin_theta * sp, cos_theta);
    float omega = ora_max(l->center[0], 1e-12f);
    out->direction = v3_normalize(ora_align_to_normal(local, v3_neg(c3(l->normal))));
    out->distance = ORA_INF;
    out->radiance = v3_divs(c3(l->radiance), omega);
    out->pdf = 1.0f / omega;
    return 1;
  }
  if (l->kind == ORA_LIGHT_DOME) { /* light.rs:358-383, uniform over the sphere */
    float z = 1.0f - 2.0f * u;
    float r = sqrtf(ora_max(1.0f - z * z, 0.0f));
    float phi = (2.0f * ORA_PI) * v;
    float sp, cp;
    ora_sincosf(phi, &sp, &cp);
    out->direction = v3_new(r * cp, z, r * sp);
    out->distance = ORA_INF;
    out->radiance = v3_mul(c3(l->radiance), v3_splat(1.0f));
    out->pdf = 1.0f / (4.0f * ORA_PI);
    return 1;
  }
  v3 lp = light_sample_point(l, u, v);
  v3 to_light = v3_sub(lp, from);
  float distance = v3_len(to_light);
  if (distance < 1e-6f) return 0;
  out->direction = v3_divs(to_light, distance);
  out->distance = distance;
  out->radiance = c3(l->radiance);
  out->pdf = solid_angle_pdf(l, from, lp);
  return 1;
}
float ora_light_pdf_at_point(const OraLight *l, v3 from, v3 light_point) {
  if (l->kind >= ORA_LIGHT_DISTANT) return 0.0f; /* light.rs:132-134: no geometry to hit */
  return solid_angle_pdf(l, from, light_point);
}

/* ------------------------------------------------------------------ */
/* camera.rs                                                           */
/* ------------------------------------------------------------------ */
void ora_camera_new(OraCamera *c, v3 lookfrom, v3 lookat, v3 vup, float vfov, float aspect, float aperture,
                    float focus_dist) { /* camera.rs:27-63 */
  float theta = vfov * ORA_PI / 180.0f; /* common.rs:8-10 */
  float h = tanf(theta / 2.0f);         /* host-side setup only: libm */
  float viewport_height = 2.0f * h;
  float viewport_width = aspect * viewport_height;
  v3 w = v3_normalize(v3_sub(lookfrom, lookat));
  v3 u = v3_normalize(v3_cross(vup, w));
  v3 v = v3_cross(w, u);
  c->origin = lookfrom;
  c->horizontal = v3_scale(u, focus_dist * viewport_width);
  c->vertical = v3_scale(v, focus_dist * viewport_height);
  c->lower_left = v3_sub(v3_sub(v3_sub(lookfrom, v3_divs(c->horizontal, 2.0f)), v3_divs(c->vertical, 2.0f)),
                         v3_scale(w, focus_dist));
  c->u = u; c->v = v;
  c->lens_radius = aperture / 2.0f;
}
void ora_camera_get_ray(const OraCamera *c, float s, float t, float lu, float lv, float time, OraRay *out) {
  v3 offset = v3_splat(0.0f); /* camera.rs:71-84 */
  if (c->lens_radius > 0.0f) {
    v3 rd = v3_scale(ora_concentric_disk(lu, lv), c->lens_radius);
    offset = v3_add(v3_scale(c->u, rd.x), v3_scale(c->v, rd.y));
  }
  out->origin = v3_add(c->origin, offset);
  out->dir = v3_sub(v3_sub(v3_add(v3_add(c->lower_left, v3_scale(c->horizontal, s)), v3_scale(c->vertical, t)),
                           c->origin), offset);
  out->time = time;
  out->mask = ORA_MASK_CAMERA;
}

/* ------------------------------------------------------------------ */
/* filter.rs                                                           */
/* ------------------------------------------------------------------ */
void ora_filter_sample(int kind, float r, float u, float *offset, float *weight) { /* filter.rs:182-205 */
  *weight = 1.0f;
  if (kind == ORA_FILTER_BOX) { *offset = (0.5f - r) + (2.0f * r) * u; return; }
  float x = u < 0.5f ? r * (sqrtf(2.0f * u) - 1.0f) : r * (1.0f - sqrtf(2.0f * (1.0f - u)));
  *offset = 0.5f + x;
}

/* ------------------------------------------------------------------ */
/* Exported views of the file-local functions, for the known-answer    */
/* tests ported from openpbr.rs:1259-2240 (tests/test_oracle_shade.py) */
/* ------------------------------------------------------------------ */
float ora_t_sheen_charlie(float nv, float nl, float nh, float roughness) { /* brdf.rs:279-281 */
  return sheen_charlie_d(nh, roughness) * sheen_charlie_v(nv, nl);
}
v3 ora_t_coat_darkening_factor(v3 base_color, float coat_ior, float darkening) {
  return coat_darkening_factor(base_color, coat_ior, darkening);
}
v3 ora_t_coat_attenuation(const OraMaterial *m, float cos_v, float cos_l) { return coat_attenuation(m, cos_v, cos_l); }
v3 ora_t_eval_coat(const OraMaterial *m, v3 v, v3 l, v3 h, float ax, float ay) { return eval_coat(m, v, l, h, ax, ay); }
v3 ora_t_thin_film_fresnel_metal(float cos1, float eta1, float eta_film, v3 f0, float thickness_nm) {
  return thin_film_fresnel_metal(cos1, eta1, eta_film, f0, thickness_nm);
}
v3 ora_t_transmission_iors(const OraMaterial *m) { return transmission_iors(m); }
v3 ora_t_sample_transmission_thin(const OraMaterial *m, v3 ray_dir, const OraHitRecord *rec) {
  v3 d, thr;
  sample_transmission_thin(m, ray_dir, rec, &d, &thr);
  return thr;
}
v3 ora_t_light_sample_point(const OraLight *l, float u, float v) { return light_sample_point(l, u, v); }
v3 ora_t_light_normal_at(const OraLight *l, v3 p) { return light_normal_at(l, p); }
float ora_t_light_area(const OraLight *l) { return light_area(l); }
/* sampler (ora_qmc.h is header-only) */
uint32_t ora_t_sampler_new(int x, int y, int frame, int index) { return ora_sampler_new(x, y, frame, index).pattern; }
uint32_t ora_t_new_domain(uint32_t pattern, int key) {
  OraSampler s = {pattern, 0};
  return ora_new_domain(s, key).pattern;
}
void ora_t_draw4(uint32_t pattern, uint32_t index, float out[4]) {
  OraSampler s = {pattern, index};
  ora_draw_sample4(s, out);
}
float ora_t_rnd1(uint32_t pattern, uint32_t index) {
  OraSampler s = {pattern, index};
  return ora_draw_rnd1(s);
}

/* ------------------------------------------------------------------ */
/* Batched drivers over the trait-level functions above, one call per  */
/* record: the checker's side of tests/test_gpu_shading_seam.py and of */
/* tests/host_shade/compare_oracle.cpp. Record layouts = include/crt.h */
/* (CrtShadeQuery 80 B, CrtScatterSample 48 B, CrtBsdfEval 32 B,       */
/* CrtLightQuery 48 B, CrtLightSample 48 B), restated here.            */
/* ------------------------------------------------------------------ */
typedef struct {
  float ray_dir[3]; uint32_t material; float p[3]; float t; float normal[3]; uint32_t front_face;
  float wi[3]; float cos_theta_o; uint32_t sampler_pattern, sampler_index; uint32_t pad[2];
} OraShadeQuery;
typedef struct { float origin[3]; uint32_t some; float dir[3]; float pdf; float value[3]; uint32_t flags; } OraScatterOut;
typedef struct { float value[3]; float pdf; uint32_t some; uint32_t pad[3]; } OraEvalOut;
typedef struct { float from[3]; uint32_t light; float u, v; uint32_t pad[2]; float point[3]; uint32_t pad2; } OraLightQuery;
typedef struct { float direction[3]; float distance; float radiance[3]; float pdf; uint32_t some; uint32_t pad[3]; } OraLightOut;
_Static_assert(sizeof(OraShadeQuery) == 80 && sizeof(OraScatterOut) == 48 && sizeof(OraEvalOut) == 32, "crt.h layouts");
_Static_assert(sizeof(OraLightQuery) == 48 && sizeof(OraLightOut) == 48, "crt.h layouts");

static OraHitRecord query_rec(const OraShadeQuery *q) {
  OraHitRecord rec;
  rec.p = c3(q->p); rec.normal = c3(q->normal); rec.t = q->t; rec.front_face = q->front_face != 0;
  return rec;
}
static void put3(float dst[3], v3 a) { dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; }

void ora_t_scatter_n(const OraMaterial *mats, size_t n_mats, const OraShadeQuery *qs, size_t n, OraScatterOut *out) {
  for (size_t i = 0; i < n; i++) { /* Material::scatter_importance, material.rs:40-45 */
    memset(&out[i], 0, sizeof out[i]);
    if (qs[i].material >= n_mats) continue;
    OraHitRecord rec = query_rec(&qs[i]);
    OraSampler dom = {qs[i].sampler_pattern, qs[i].sampler_index};
    OraScatter sc;
    if (!ora_mat_scatter(&mats[qs[i].material], c3(qs[i].ray_dir), &rec, dom, &sc)) continue;
    put3(out[i].origin, sc.origin); put3(out[i].dir, sc.dir); put3(out[i].value, sc.value);
    out[i].some = 1; out[i].pdf = sc.pdf; out[i].flags = (sc.delta ? 1u : 0u) | (sc.medium ? 2u : 0u);
  }
}
void ora_t_eval_n(const OraMaterial *mats, size_t n_mats, const OraShadeQuery *qs, size_t n, OraEvalOut *out) {
  for (size_t i = 0; i < n; i++) { /* Material::eval, material.rs:56-74 */
    memset(&out[i], 0, sizeof out[i]);
    if (qs[i].material >= n_mats) continue;
    OraHitRecord rec = query_rec(&qs[i]);
    v3 value; float pdf;
    if (!ora_mat_eval(&mats[qs[i].material], c3(qs[i].ray_dir), &rec, c3(qs[i].wi), &value, &pdf)) continue;
    put3(out[i].value, value); out[i].pdf = pdf; out[i].some = 1;
  }
}
void ora_t_emitted_n(const OraMaterial *mats, size_t n_mats, const OraShadeQuery *qs, size_t n, float *rgb) {
  for (size_t i = 0; i < n; i++) { /* Material::emitted_directional, material.rs:112-115 */
    v3 e = v3_splat(0.0f);
    if (qs[i].material < n_mats) e = ora_mat_emitted_directional(&mats[qs[i].material], qs[i].cos_theta_o);
    put3(rgb + 3 * i, e);
  }
}
void ora_t_light_sample_n(const OraLight *ls, size_t n_ls, const OraLightQuery *qs, size_t n, OraLightOut *out) {
  for (size_t i = 0; i < n; i++) { /* Light::sample_li, light.rs:126 */
    memset(&out[i], 0, sizeof out[i]);
    OraLightSample s;
    if (qs[i].light >= n_ls || !ora_light_sample_li(&ls[qs[i].light], c3(qs[i].from), qs[i].u, qs[i].v, &s)) continue;
    put3(out[i].direction, s.direction); out[i].distance = s.distance; put3(out[i].radiance, s.radiance);
    out[i].pdf = s.pdf; out[i].some = 1;
  }
}
void ora_t_light_pdf_n(const OraLight *ls, size_t n_ls, const OraLightQuery *qs, size_t n, float *pdf) {
  for (size_t i = 0; i < n; i++) /* Light::pdf_at_point, light.rs:132 */
    pdf[i] = qs[i].light < n_ls ? ora_light_pdf_at_point(&ls[qs[i].light], c3(qs[i].from), c3(qs[i].point)) : 0.0f;
}
void ora_t_light_escaped_n(const OraLight *ls, size_t n_ls, const OraLightQuery *qs, size_t n, OraLightOut *out) {
  for (size_t i = 0; i < n; i++) { /* Light::escaped, light.rs:141 */
    memset(&out[i], 0, sizeof out[i]);
    v3 rad; float pdf;
    if (qs[i].light >= n_ls || !ora_light_escaped(&ls[qs[i].light], c3(qs[i].point), &rad, &pdf)) continue;
    put3(out[i].direction, c3(qs[i].point)); out[i].distance = INFINITY; put3(out[i].radiance, rad);
    out[i].pdf = pdf; out[i].some = 1;
  }
}
