// Device math for the shading kernels: float3 helpers with glam Vec3A semantics and the transcendental
// functions as fixed sequences of IEEE-754 double operations.
//
// Why not __sinf/ocml: the parity contract of this build is bit-for-bit equality of the rendered image
// with the CPU oracle. IEEE +,-,*,/ and sqrt are correctly rounded on both sides (hipcc expands f32/f64
// division and sqrt to the correctly rounded sequences by default), so any function built only from those
// — in one fixed order, compiled with -ffp-contract=off — yields identical bits on gfx950 and x86-64.
// The kernels below follow the classic msun argument reductions and minimax/Taylor polynomials; after the
// final rounding to f32 they are within 1 ulp, the same contract as the f32::sin/cos/acos/exp/ln/powf the
// reference calls (brdf.rs:101-102, :270; light.rs:28-34; common.rs:128-135; openpbr.rs:598, :802).
// CDNA4 runs f64 FMA/ADD/MUL at half the f32 vector rate, so a sincos costs ~20 DP ops: cheap next to
// the BVH traversal that dominates a path vertex.
#pragma once

#include <hip/hip_runtime.h>

namespace crt {
namespace dev {

#define CRT_PI 3.14159265358979323846264338327950288f
#define CRT_INF (__builtin_inff())

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 splat(float s) { return V3{s, s, s}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float len2(V3 a) { return dot(a, a); }
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { return a / sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// f32::max / f32::min / f32::clamp (Rust): the non-NaN operand; the second operand on ties.
__device__ __forceinline__ float rmax(float a, float b) { return (a > b || b != b) ? a : b; }
__device__ __forceinline__ float rmin(float a, float b) { return (a < b || b != b) ? a : b; }
__device__ __forceinline__ float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ float fabs_(float x) { return __uint_as_float(__float_as_uint(x) & 0x7fffffffu); }
// glam Vec3A::min/max/clamp: SSE minps/maxps (second operand on ties or NaN).
__device__ __forceinline__ float smin(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float smax(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ V3 vmin(V3 a, V3 b) { return {smin(a.x, b.x), smin(a.y, b.y), smin(a.z, b.z)}; }
__device__ __forceinline__ V3 vmax(V3 a, V3 b) { return {smax(a.x, b.x), smax(a.y, b.y), smax(a.z, b.z)}; }
__device__ __forceinline__ V3 vclamp(V3 a, V3 lo, V3 hi) { return vmin(vmax(a, lo), hi); }
__device__ __forceinline__ V3 lerp(V3 a, V3 b, float s) { return a * (1.0f - s) + b * s; }  // glam lerp
__device__ __forceinline__ float max_elem(V3 a) { return smax(smax(a.x, a.y), a.z); }
__device__ __forceinline__ float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// ---- sin / cos ----
__device__ __forceinline__ double ksin(double x) {
  const double S1 = -0.166666666416265235595, S2 = 0.0083333293858894631756, S3 = -0.000198393348360966317347,
               S4 = 0.0000027183114939898219064;
  const double z = x * x;
  const double w = z * z;
  const double r = S3 + z * S4;
  const double s = z * x;
  return (x + s * (S1 + z * S2)) + s * w * r;
}
__device__ __forceinline__ double kcos(double x) {
  const double C0 = -0.499999997251031003120, C1 = 0.0416666233237390631894, C2 = -0.00138867637746099294692,
               C3 = 0.0000243904487962774090654;
  const double z = x * x;
  const double w = z * z;
  const double r = C2 + z * C3;
  return ((1.0 + z * C0) + w * C1) + (w * z) * r;
}
__device__ __forceinline__ void sincos_det(float xf, float &s, float &c) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632673412561417e+00;
  const double PIO2_LO = 6.07710050650619224932e-11;
  const double x = (double)xf;
  const double fn = rint(x * INV_PIO2);
  const double y = (x - fn * PIO2_HI) - fn * PIO2_LO;
  const int n = (int)(long long)fn;
  const double sy = ksin(y), cy = kcos(y);
  switch (n & 3) {
    case 0: s = (float)sy; c = (float)cy; break;
    case 1: s = (float)cy; c = (float)(-sy); break;
    case 2: s = (float)(-sy); c = (float)(-cy); break;
    default: s = (float)(-cy); c = (float)sy; break;
  }
}
__device__ __forceinline__ float cos_det(float x) { float s, c; sincos_det(x, s, c); return c; }

// ---- acos ----
__device__ __forceinline__ double asin_r(double z) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  return p / q;
}
__device__ __forceinline__ float acos_det(float xf) {
  const double PIO2 = 1.57079632679489655800e+00, PI_D = 3.14159265358979311600e+00;
  const double x = (double)xf;
  if (x != x) return xf;
  if (x >= 1.0) return 0.0f;
  if (x <= -1.0) return (float)PI_D;
  const double ax = x < 0.0 ? -x : x;
  if (ax < 0.5) {
    const double z = x * x;
    return (float)(PIO2 - (x + x * asin_r(z)));
  }
  const double z = (1.0 - ax) * 0.5;
  const double s = sqrt(z);
  const double t = 2.0 * (s + s * asin_r(z));
  return (float)(x < 0.0 ? PI_D - t : t);
}

// ---- exp / log / pow ----
__device__ __forceinline__ double exp_d(double x) {
  const double INV_LN2 = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  if (x != x) return x;
  if (x > 709.0) return (double)CRT_INF;
  if (x < -745.0) return 0.0;
  const double fk = rint(x * INV_LN2);
  const double r = (x - fk * LN2_HI) - fk * LN2_LO;
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
  const long long k = (long long)fk;
  const long long k1 = k / 2, k2 = k - k1;
  const double s1 = __longlong_as_double((long long)((unsigned long long)(k1 + 1023) << 52));
  const double s2 = __longlong_as_double((long long)((unsigned long long)(k2 + 1023) << 52));
  return (p * s1) * s2;
}
__device__ __forceinline__ double log_d(double x) {
  const double LN2 = 6.93147180559945286227e-01, SQRT2 = 1.41421356237309514547e+00;
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  long long e = (long long)((b >> 52) & 0x7ff) - 1023;
  double m = __longlong_as_double((long long)((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
  if (m > SQRT2) { m = m * 0.5; e += 1; }
  const double s = (m - 1.0) / (m + 1.0);
  const double z = s * s;
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
__device__ __forceinline__ float pow_det(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (x != x || y != y) return __uint_as_float(0x7fc00000u);
  if (x == 1.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : CRT_INF;
  if (x < 0.0f) return __uint_as_float(0x7fc00000u);
  if (x == CRT_INF) return y > 0.0f ? CRT_INF : 0.0f;
  return (float)exp_d((double)y * log_d((double)x));
}
__device__ __forceinline__ float exp_det(float x) { return (float)exp_d((double)x); }
// ln(0) = -inf, ln(<0) = NaN like f32::ln
__device__ __forceinline__ float log_det(float x) {
  if (x != x) return x;
  if (x == 0.0f) return -CRT_INF;
  if (x < 0.0f) return __uint_as_float(0x7fc00000u);
  if (x == CRT_INF) return x;
  return (float)log_d((double)x);
}
__device__ __forceinline__ float pow2_(float x) { return x * x; }
__device__ __forceinline__ float pow5_(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x4 * x; }
__device__ __forceinline__ float pow6_(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x4 * x2; }

}  // namespace dev
}  // namespace crt
