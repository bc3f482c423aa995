// On-device path sampler: Owen-scrambled Sobol (Burley 2020 hash-based Owen scrambling over the Joe-Kuo
// direction numbers of the first four dimensions) with a PCG-hashed domain tree.
//
// API shape = the reference's PathSampler call sites (tracer.rs:559-561, :1101, :1121, :1395, :1482;
// openpbr.rs:1042): new(x, y, frame, index), new_domain(key), draw_sample_f32::<4>, draw_rnd_f32::<1>.
// The reference's generator is the third-party crate openqmc-rs 0.1, whose source is not available
// (SURVEY §8c): sample VALUES are this build's own — "parity unpinned" against the reference, pinned
// bit-for-bit against the CPU oracle, which states the same algorithm independently.
#pragma once

#include <hip/hip_runtime.h>

namespace crt {
namespace dev {

struct Sampler { uint32_t pattern, index; };

__device__ const uint32_t kSobolDirs[4][32] = {
    {0x80000000u, 0x40000000u, 0x20000000u, 0x10000000u, 0x08000000u, 0x04000000u, 0x02000000u, 0x01000000u,
     0x00800000u, 0x00400000u, 0x00200000u, 0x00100000u, 0x00080000u, 0x00040000u, 0x00020000u, 0x00010000u,
     0x00008000u, 0x00004000u, 0x00002000u, 0x00001000u, 0x00000800u, 0x00000400u, 0x00000200u, 0x00000100u,
     0x00000080u, 0x00000040u, 0x00000020u, 0x00000010u, 0x00000008u, 0x00000004u, 0x00000002u, 0x00000001u},
    {0x80000000u, 0xc0000000u, 0xa0000000u, 0xf0000000u, 0x88000000u, 0xcc000000u, 0xaa000000u, 0xff000000u,
     0x80800000u, 0xc0c00000u, 0xa0a00000u, 0xf0f00000u, 0x88880000u, 0xcccc0000u, 0xaaaa0000u, 0xffff0000u,
     0x80008000u, 0xc000c000u, 0xa000a000u, 0xf000f000u, 0x88008800u, 0xcc00cc00u, 0xaa00aa00u, 0xff00ff00u,
     0x80808080u, 0xc0c0c0c0u, 0xa0a0a0a0u, 0xf0f0f0f0u, 0x88888888u, 0xccccccccu, 0xaaaaaaaau, 0xffffffffu},
    {0x80000000u, 0xc0000000u, 0x60000000u, 0x90000000u, 0xe8000000u, 0x5c000000u, 0x8e000000u, 0xc5000000u,
     0x68800000u, 0x9cc00000u, 0xee600000u, 0x55900000u, 0x80680000u, 0xc09c0000u, 0x60ee0000u, 0x90550000u,
     0xe8808000u, 0x5cc0c000u, 0x8e606000u, 0xc5909000u, 0x6868e800u, 0x9c9c5c00u, 0xeeee8e00u, 0x5555c500u,
     0x8000e880u, 0xc0005cc0u, 0x60008e60u, 0x9000c590u, 0xe8006868u, 0x5c009c9cu, 0x8e00eeeeu, 0xc5005555u},
    {0x80000000u, 0xc0000000u, 0x20000000u, 0x50000000u, 0xf8000000u, 0x74000000u, 0xa2000000u, 0x93000000u,
     0xd8800000u, 0x25400000u, 0x59e00000u, 0xe6d00000u, 0x78080000u, 0xb40c0000u, 0x82020000u, 0xc3050000u,
     0x208f8000u, 0x51474000u, 0xfbea2000u, 0x75d93000u, 0xa0858800u, 0x914e5400u, 0xdbe79e00u, 0x25db6d00u,
     0x58800080u, 0xe54000c0u, 0x79e00020u, 0xb6d00050u, 0x800800f8u, 0xc00c0074u, 0x200200a2u, 0x50050093u}};

__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
__device__ __forceinline__ uint32_t laine_karras(uint32_t x, uint32_t seed) {
  x ^= x * 0x3d20adeau;
  x += seed;
  x *= (seed >> 16) | 1u;
  x ^= x * 0x05526c56u;
  x ^= x * 0x53a22864u;
  return x;
}
__device__ __forceinline__ uint32_t owen(uint32_t x, uint32_t seed) {
  return __brev(laine_karras(__brev(x), seed));  // v_bfrev_b32
}
__device__ __forceinline__ float unit_f32(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ Sampler sampler_new(int x, int y, int frame, int index) {
  const uint32_t pixel = ((uint32_t)x & 0xffu) | (((uint32_t)y & 0xffu) << 8);
  return Sampler{pcg_hash(pcg_hash((uint32_t)frame) ^ pixel), (uint32_t)index};
}
__device__ __forceinline__ Sampler new_domain(Sampler s, int key) {
  return Sampler{pcg_hash(s.pattern + 0x9e3779b9u * ((uint32_t)key + 1u)), s.index};
}
// Byte-sliced Sobol matrices for dimensions 1..3, kept in LDS: entry [(d-1)*4 + k][v] is the XOR of the
// direction numbers 8k..8k+7 selected by the bits of byte v, so a 32-bit matrix-vector product over GF(2) is
// four table reads instead of a 32-step loop of dependent global loads. Same linear map, same bits.
constexpr int kSobolLdsWords = 3 * 4 * 256;

__device__ __forceinline__ void sobol_tables_init(uint32_t *tab /* LDS, kSobolLdsWords */) {
  for (int e = threadIdx.x; e < kSobolLdsWords; e += blockDim.x) {
    const int v = e & 255, k = (e >> 8) & 3, d = (e >> 10) + 1;
    uint32_t x = 0;
#pragma unroll
    for (int b = 0; b < 8; b++)
      if ((v >> b) & 1) x ^= kSobolDirs[d][8 * k + b];
    tab[e] = x;
  }
  __syncthreads();
}

__device__ __forceinline__ void draw_sample4(Sampler s, float out[4], const uint32_t *tab) {
  const uint32_t idx = owen(s.index, pcg_hash(s.pattern));
  const uint32_t b0 = idx & 255u, b1 = (idx >> 8) & 255u, b2 = (idx >> 16) & 255u, b3 = idx >> 24;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t x;
    if (d == 0) {
      x = __brev(idx);  // dimension 0's matrix is the identity on reversed bits
    } else {
      const uint32_t *t = tab + (d - 1) * 1024;
      x = t[b0] ^ t[256 + b1] ^ t[512 + b2] ^ t[768 + b3];
    }
    x = owen(x, pcg_hash(s.pattern + (uint32_t)d + 1u));
    out[d] = unit_f32(x);
  }
}
__device__ __forceinline__ float draw_rnd1(Sampler s) {
  return unit_f32(pcg_hash(s.pattern ^ (s.index * 0x9e3779b9u + 0x7f4a7c15u)));
}

}  // namespace dev
}  // namespace crt
