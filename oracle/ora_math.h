/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the arithmetic crust-render's hot path relies
 * on. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link or call anything under oracle/. The product (crust-render_amd/)
 * never includes this file.
 *
 * Floating-point contract: every translation unit under oracle/
 * is compiled with -ffp-contract=off and no fast-math, so each +,-,*,/ and
 * sqrt is one IEEE-754 round-to-nearest operation in source order. The
 * transcendental functions (sin, cos, acos, exp, log, pow) are NOT taken from
 * libm: they are evaluated by fixed sequences of IEEE double operations
 * (msun-style kernels), so that an independent implementation that performs
 * the same sequence (the HIP kernels) reproduces them bit for bit. Accuracy
 * is < 1 ulp in f32 after the final rounding, i.e. the same contract as Rust's
 * f32::sin & co. that the reference calls (brdf.rs:101-102, light.rs:28-34,
 * common.rs:128-135).
 *
 * Vector helpers restate glam 0.33 Vec3A semantics on SSE2 (the reference's
 * build, docs/simd.md:4-5): component-wise IEEE ops, dot = (x*x'+y*y')+z*z',
 * normalize = v / sqrt(dot), min/max = SSE minps/maxps (second operand on
 * equality or NaN).
 */
#ifndef ORA_MATH_H
#define ORA_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORA_PI 3.14159265358979323846264338327950288f /* std::f32::consts::PI */
#define ORA_INF (__builtin_inff())

typedef struct { float x, y, z; } v3;

static inline uint32_t ora_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float ora_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t ora_d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double ora_u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* Rust f32::max / f32::min: the non-NaN operand; on ties the second. */
static inline float ora_max(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float ora_min(float a, float b) { return (a < b || b != b) ? a : b; }
/* Rust f32::clamp. */
static inline float ora_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float ora_abs(float x) { return ora_u2f(ora_f2u(x) & 0x7fffffffu); }
static inline float ora_copysign(float mag, float sgn) {
  return ora_u2f((ora_f2u(mag) & 0x7fffffffu) | (ora_f2u(sgn) & 0x80000000u));
}
/* SSE minps/maxps as glam's Vec3A::min/max use them. */
static inline float ora_sse_min(float a, float b) { return a < b ? a : b; }
static inline float ora_sse_max(float a, float b) { return a > b ? a : b; }

static inline v3 v3_new(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_splat(float s) { return v3_new(s, s, s); }
static inline v3 v3_add(v3 a, v3 b) { return v3_new(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_new(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_new(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_div(v3 a, v3 b) { return v3_new(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_new(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_divs(v3 a, float s) { return v3_new(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return v3_new(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float v3_len2(v3 a) { return v3_dot(a, a); }
static inline float v3_len(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) { return v3_divs(a, sqrtf(v3_dot(a, a))); }
static inline v3 v3_cross(v3 a, v3 b) {
  return v3_new(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 v3_min(v3 a, v3 b) { return v3_new(ora_sse_min(a.x, b.x), ora_sse_min(a.y, b.y), ora_sse_min(a.z, b.z)); }
static inline v3 v3_max(v3 a, v3 b) { return v3_new(ora_sse_max(a.x, b.x), ora_sse_max(a.y, b.y), ora_sse_max(a.z, b.z)); }
static inline v3 v3_clamp(v3 a, v3 lo, v3 hi) { return v3_min(v3_max(a, lo), hi); }
/* glam lerp: self * (1 - s) + rhs * s */
static inline v3 v3_lerp(v3 a, v3 b, float s) { return v3_add(v3_scale(a, 1.0f - s), v3_scale(b, s)); }
static inline float v3_max_elem(v3 a) { return ora_sse_max(ora_sse_max(a.x, a.y), a.z); }
static inline float v3_min_elem(v3 a) { return ora_sse_min(ora_sse_min(a.x, a.y), a.z); }
static inline float v3_get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline void v3_set(v3 *a, int i, float v) { if (i == 0) a->x = v; else if (i == 1) a->y = v; else a->z = v; }
static inline int v3_eq(v3 a, v3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

/* ---------------------------------------------------------------------
 * Deterministic transcendental functions (IEEE double op sequences).
 * ------------------------------------------------------------------- */

/* sin/cos kernels on |y| <= pi/4 (FreeBSD msun k_sinf.c / k_cosf.c forms). */
static inline double ora_ksin(double x) {
  const double S1 = -0.166666666416265235595, S2 = 0.0083333293858894631756,
               S3 = -0.000198393348360966317347, S4 = 0.0000027183114939898219064;
  double z = x * x;
  double w = z * z;
  double r = S3 + z * S4;
  double s = z * x;
  return (x + s * (S1 + z * S2)) + s * w * r;
}
static inline double ora_kcos(double x) {
  const double C0 = -0.499999997251031003120, C1 = 0.0416666233237390631894,
               C2 = -0.00138867637746099294692, C3 = 0.0000243904487962774090654;
  double z = x * x;
  double w = z * z;
  double r = C2 + z * C3;
  return ((1.0 + z * C0) + w * C1) + (w * z) * r;
}
/* Simultaneous sinf/cosf. Arguments on the hot path are in [0, 2*pi] (phi =
 * 2*pi*u) or small thin-film phases; the two-term pi/2 reduction is exact to
 * double precision for |x| < 1e5. */
static inline void ora_sincosf(float xf, float *s, float *c) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632673412561417e+00;
  const double PIO2_LO = 6.07710050650619224932e-11;
  double x = (double)xf;
  double fn = rint(x * INV_PIO2);
  double y = (x - fn * PIO2_HI) - fn * PIO2_LO;
  int n = (int)(long long)fn;
  double sy = ora_ksin(y), cy = ora_kcos(y);
  switch (n & 3) {
    case 0: *s = (float)sy; *c = (float)cy; break;
    case 1: *s = (float)cy; *c = (float)(-sy); break;
    case 2: *s = (float)(-sy); *c = (float)(-cy); break;
    default: *s = (float)(-cy); *c = (float)sy; break;
  }
}
static inline float ora_sinf(float x) { float s, c; ora_sincosf(x, &s, &c); return s; }
static inline float ora_cosf(float x) { float s, c; ora_sincosf(x, &s, &c); return c; }

/* acosf via the msun e_asin.c rational in double. */
static inline double ora_asin_r(double z) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  return p / q;
}
static inline float ora_acosf(float xf) {
  const double PIO2 = 1.57079632679489655800e+00, PI_D = 3.14159265358979311600e+00;
  double x = (double)xf;
  if (x != x) return xf;
  if (x >= 1.0) return 0.0f;
  if (x <= -1.0) return (float)PI_D;
  double ax = x < 0.0 ? -x : x;
  if (ax < 0.5) {
    double z = x * x;
    return (float)(PIO2 - (x + x * ora_asin_r(z)));
  }
  double z = (1.0 - ax) * 0.5;
  double s = sqrt(z);
  double t = 2.0 * (s + s * ora_asin_r(z)); /* acos(|x|) */
  return (float)(x < 0.0 ? PI_D - t : t);
}

/* exp in double: k = rint(x/ln2), degree-11 Taylor on the remainder. */
static inline double ora_exp_d(double x) {
  const double INV_LN2 = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  if (x != x) return x;
  if (x > 709.0) return (double)ORA_INF;
  if (x < -745.0) return 0.0;
  double fk = rint(x * INV_LN2);
  double r = (x - fk * LN2_HI) - fk * LN2_LO;
  double p = 1.0 / 39916800.0;
  p = 1.0 / 3628800.0 + r * p;
  p = 1.0 / 362880.0 + r * p;
  p = 1.0 / 40320.0 + r * p;
  p = 1.0 / 5040.0 + r * p;
  p = 1.0 / 720.0 + r * p;
  p = 1.0 / 120.0 + r * p;
  p = 1.0 / 24.0 + r * p;
  p = 1.0 / 6.0 + r * p;
  p = 0.5 + r * p;
  p = 1.0 + r * p;
  p = 1.0 + r * p;
  long long k = (long long)fk;
  /* 2^k in two steps so results in the f32 subnormal range stay exact. */
  long long k1 = k / 2, k2 = k - k1;
  double s1 = ora_u2d((uint64_t)(k1 + 1023) << 52);
  double s2 = ora_u2d((uint64_t)(k2 + 1023) << 52);
  return (p * s1) * s2;
}
/* log in double for finite x > 0: atanh series on m in [sqrt(1/2), sqrt(2)). */
static inline double ora_log_d(double x) {
  const double LN2 = 6.93147180559945286227e-01, SQRT2 = 1.41421356237309514547e+00;
  uint64_t b = ora_d2u(x);
  long long e = (long long)((b >> 52) & 0x7ff) - 1023;
  double m = ora_u2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > SQRT2) { m = m * 0.5; e += 1; }
  double s = (m - 1.0) / (m + 1.0);
  double z = s * s;
  double p = 1.0 / 19.0;
  p = 1.0 / 17.0 + z * p;
  p = 1.0 / 15.0 + z * p;
  p = 1.0 / 13.0 + z * p;
  p = 1.0 / 11.0 + z * p;
  p = 1.0 / 9.0 + z * p;
  p = 1.0 / 7.0 + z * p;
  p = 1.0 / 5.0 + z * p;
  p = 1.0 / 3.0 + z * p;
  p = 1.0 + z * p;
  return (double)e * LN2 + 2.0 * s * p;
}
static inline float ora_expf(float x) { return (float)ora_exp_d((double)x); }
/* ln(x) for x > 0 (f32 subnormals are normal doubles, so no special case);
 * ln(0) = -inf, ln(<0) = NaN like f32::ln. */
static inline float ora_logf(float x) {
  if (x != x) return x;
  if (x == 0.0f) return -ORA_INF;
  if (x < 0.0f) return ora_u2f(0x7fc00000u);
  if (x == ORA_INF) return x;
  return (float)ora_log_d((double)x);
}
/* powf for the domain the path uses: base in [0, inf), finite exponent. */
static inline float ora_powf(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (x != x || y != y) return ora_u2f(0x7fc00000u);
  if (x == 1.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : ORA_INF;
  if (x < 0.0f) return ora_u2f(0x7fc00000u);
  if (x == ORA_INF) return y > 0.0f ? ORA_INF : 0.0f;
  return (float)ora_exp_d((double)y * ora_log_d((double)x));
}
/* f32::powi(n) for the small constant exponents the path uses. */
static inline float ora_pow2(float x) { return x * x; }
static inline float ora_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
static inline float ora_pow6(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x2; }

#endif /* ORA_MATH_H */
