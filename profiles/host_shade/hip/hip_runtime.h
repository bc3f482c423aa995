// Host stand-in for <hip/hip_runtime.h>, for profiles/host_shade/shade_host.cpp ONLY: the few HIP names the shading
// headers (kernels/shade.hip.h, dmath.hip.h, qmc.hip.h) use, so that they compile as plain C++ and can run on the CPU
// under MemorySanitizer / UndefinedBehaviorSanitizer. Not part of the product, never on an include path of the library.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__
struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
static inline uint32_t __brev(uint32_t x) {
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
  x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
  return (x >> 16) | (x << 16);
}
// one "thread" fills the Sobol tables
struct Dim3 { unsigned x = 0, y = 0, z = 0; };
static const Dim3 threadIdx{0, 0, 0};
static const Dim3 blockDim{1, 1, 1};
static inline void __syncthreads() {}
