/* ORACLE — TEST INFRASTRUCTURE ONLY (see ora_math.h header).
 *
 * Path sampler: Owen-scrambled Sobol with a pcg-hashed domain tree.
 *
 * PARITY UNPINNED against the reference: the reference draws its numbers from
 * the third-party crate `openqmc-rs = "0.1"` (crates/crust-core/Cargo.toml:17,
 * alias PathSampler = openqmc::SobolSampler, crust-core/src/lib.rs:24), whose
 * source is not under /root/reference and cannot be fetched. This file
 * restates the *published algorithm family* OpenQMC's SobolSampler implements
 * — Burley 2020 "Practical Hash-based Owen Scrambling" (Laine-Karras
 * permutation on reversed bits, shuffled index, Joe-Kuo direction numbers for
 * the first four dimensions) with PCG-RXS-M-XS hashing for the domain tree —
 * and keeps the reference's call-site API: new(x, y, frame, index),
 * new_domain(key), draw_sample_f32::<N<=4>() (always one 4-D block,
 * tracer.rs:549-551), draw_rnd_f32::<1>(). Bit patterns of the samples are
 * therefore this build's own; the reference's tests touching the sampler
 * assert only relations that any uniform [0,1)^4 source satisfies
 * (openpbr.rs:1247-1253).
 */
#ifndef ORA_QMC_H
#define ORA_QMC_H
#include <stdint.h>

typedef struct { uint32_t pattern; uint32_t index; } OraSampler;

static const uint32_t ORA_SOBOL_DIRS[4][32] = {
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

/* PCG-RXS-M-XS 32 (O'Neill 2014; the hash form of Jarzynski & Olano 2020). */
static inline uint32_t ora_pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
static inline uint32_t ora_reverse_bits(uint32_t x) {
  x = (x << 16) | (x >> 16);
  x = ((x & 0x00ff00ffu) << 8) | ((x & 0xff00ff00u) >> 8);
  x = ((x & 0x0f0f0f0fu) << 4) | ((x & 0xf0f0f0f0u) >> 4);
  x = ((x & 0x33333333u) << 2) | ((x & 0xccccccccu) >> 2);
  x = ((x & 0x55555555u) << 1) | ((x & 0xaaaaaaaau) >> 1);
  return x;
}
/* Burley 2020, listing 2 (improved Laine-Karras constants). */
static inline uint32_t ora_laine_karras(uint32_t x, uint32_t seed) {
  x ^= x * 0x3d20adeau;
  x += seed;
  x *= (seed >> 16) | 1u;
  x ^= x * 0x05526c56u;
  x ^= x * 0x53a22864u;
  return x;
}
static inline uint32_t ora_owen(uint32_t x, uint32_t seed) {
  return ora_reverse_bits(ora_laine_karras(ora_reverse_bits(x), seed));
}
static inline float ora_u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/* PathSampler::new(x, y, frame, index) — call site tracer.rs:559-560. Pixels decorrelate within a
 * 256x256 tile (tracer.rs:540-543 adds the tile domain on top). */
static inline OraSampler ora_sampler_new(int x, int y, int frame, int index) {
  OraSampler s;
  uint32_t pixel = ((uint32_t)x & 0xffu) | (((uint32_t)y & 0xffu) << 8);
  s.pattern = ora_pcg_hash(ora_pcg_hash((uint32_t)frame) ^ pixel);
  s.index = (uint32_t)index;
  return s;
}
/* new_domain(key): pure function of the parent state (tracer.rs:553-555). */
static inline OraSampler ora_new_domain(OraSampler s, int key) {
  OraSampler r;
  r.pattern = ora_pcg_hash(s.pattern + 0x9e3779b9u * ((uint32_t)key + 1u));
  r.index = s.index;
  return r;
}
/* draw_sample_f32::<4>: one Owen-scrambled 4-D Sobol point at the sample index. */
static inline void ora_draw_sample4(OraSampler s, float out[4]) {
  uint32_t idx = ora_owen(s.index, ora_pcg_hash(s.pattern));
  for (int d = 0; d < 4; d++) {
    uint32_t x = 0, i = idx;
    for (int b = 0; i; b++, i >>= 1)
      if (i & 1u) x ^= ORA_SOBOL_DIRS[d][b];
    x = ora_owen(x, ora_pcg_hash(s.pattern + (uint32_t)d + 1u));
    out[d] = ora_u32_to_unit(x);
  }
}
/* draw_rnd_f32::<1>: PRNG side of the domain (tracer.rs:1482). */
static inline float ora_draw_rnd1(OraSampler s) {
  return ora_u32_to_unit(ora_pcg_hash(s.pattern ^ (s.index * 0x9e3779b9u + 0x7f4a7c15u)));
}

#endif
