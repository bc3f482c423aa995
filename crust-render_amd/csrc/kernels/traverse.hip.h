// BVH4 closest-hit / any-hit traversal for gfx950 (wave64): shared types and per-ray arithmetic. The scheduler that
// drives them is traverse_pool.hip.h.
//
// What it computes: crates/crust-rt/src/bvh.rs:441-509 (Bvh::hit), :585-611 (hit_any), :514-572
// (intersect_leaf), :617-653 (occlude_leaf), :790-808 (slab4), :662-668 (safe_inv3);
// triangle.rs:41-80 (RayShear), :276-348 (Tri4::intersect), :110-172 (f64 edge fallback);
// prim.rs:76-95, :133-161, :321-378 (triangle normal, sphere, instance); scene.rs:354-366.
//
// How it is laid out for CDNA4:
//  * one 128-byte node = one cache line, fetched by a lane as 8 x 16-byte loads; a 192-byte Tri4
//    packet as 9 x 16-byte plane loads addressed by the ray's permuted axes (no per-lane selects);
//  * the traversal stack lives in LDS, one plane per entry with a lane's slots at unit stride (a wave's push or
//    pop is conflict-free); entries past the LDS depth spill to per-lane private memory;
//  * leaf lanes are pushed on the same stack as inner lanes (tagged), after the inner lanes and in
//    far-to-near order, so they pop first and near-first: the sequence of leaf tests and node visits
//    is exactly the reference's "leaf lanes now, near first; inner lanes pushed far to near"
//    (bvh.rs:488-505), which is what makes exact-tie resolution (bvh.rs:537-544) identical;
//  * instances are followed without recursion: a per-lane frame per nesting level holds the ray and
//    the position in the leaf's primitive list; the instanced tree is traversed above a stack base.
//
// Float contract: this header must be compiled with -ffp-contract=off. Every expression below is
// written in the reference's operation order; the watertightness argument (triangle.rs:88-96) and the
// bit-for-bit parity with the CPU oracle both depend on it. Slab min/max use v_min/v_max: they differ
// from SSE minps/maxps only in the sign of a zero result, which no comparison downstream can see.
#pragma once

#include <hip/hip_runtime.h>

#include "../crt_internal.h"

namespace crt {
namespace dev {

constexpr int kBlock = 256;       // threads per workgroup (4 waves)
// Top of the tree staged in LDS: the first nodes (breadth-first numbering), bounds + child words only (112 of the
// node's 128 bytes: the traversal does not read `flags`). Two nodes' 16-byte reads collide on banks only when
// their indices differ by a multiple of 16, the same as with any padded stride.
#ifndef CRT_LDS_NODE_STRIDE
#define CRT_LDS_NODE_STRIDE 28
#endif
constexpr int kLdsNodeStride = CRT_LDS_NODE_STRIDE;  // dwords
constexpr int kMaxLevels = 8;     // instance nesting depth (usd_import.rs:60 MAX_INSTANCE_NESTING)
constexpr uint32_t kLeafTag = 0x80000000u;
constexpr uint32_t kInvalid = 0xFFFFFFFFu;

struct Hit {
  float t, u, v;
  float nx, ny, nz;  // geometric outward normal, not yet oriented (prim.rs:13-16)
  uint32_t geom, prim;
};

struct LaneStats {
  uint32_t queries[2], nodes[2], leaves[2], packets[2], prims[2];
  uint32_t accepted, descents;
  uint32_t ph_wave[8], ph_lane[8];  // CrtTravStats::phase_waves / phase_lanes
  unsigned long long ph_cyc[8];     // CrtTravStats::phase_cycles (lane 0 of each wave counts)
};
// Counts one execution of a phase: every live lane counts itself, the first live lane counts the wave.
#define CRT_PHASE(k)                                                        \
  if (STATS) {                                                              \
    const unsigned long long m_ = __ballot(1);                              \
    st.ph_lane[k]++;                                                        \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) st.ph_wave[k]++; \
  }

// One atomic per counter per wave: sum across the 64 lanes first.
__device__ __forceinline__ void flush_stats(const LaneStats &st, CrtTravStats *out, uint32_t rays) {
  auto wave_sum = [](uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
  };
  const bool lead = (threadIdx.x & 63) == 0;
  auto add = [&](uint64_t *dst, uint32_t v) {
    v = wave_sum(v);
    if (lead && v) atomicAdd((unsigned long long *)dst, (unsigned long long)v);
  };
  for (int k = 0; k < 2; k++) {
    add(&out->queries[k], st.queries[k]); add(&out->nodes[k], st.nodes[k]); add(&out->leaves[k], st.leaves[k]);
    add(&out->packets[k], st.packets[k]); add(&out->prims[k], st.prims[k]);
  }
  add(&out->accepted_hits, st.accepted); add(&out->instance_descents, st.descents); add(&out->rays, rays);
  for (int k = 0; k < 8; k++) { add(&out->phase_waves[k], st.ph_wave[k]); add(&out->phase_lanes[k], st.ph_lane[k]); }
  if (lead)
    for (int k = 0; k < 8; k++)
      if (st.ph_cyc[k]) atomicAdd((unsigned long long *)&out->phase_cycles[k], st.ph_cyc[k]);
}

__device__ __forceinline__ float absf(float x) { return __uint_as_float(__float_as_uint(x) & 0x7fffffffu); }
__device__ __forceinline__ float copysgn(float mag, float sgn) {
  return __uint_as_float((__float_as_uint(mag) & 0x7fffffffu) | (__float_as_uint(sgn) & 0x80000000u));
}
__device__ __forceinline__ float sel3(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}

struct RayCtx {
  float ox, oy, oz, dx, dy, dz;
  float ix, iy, iz;  // safe_inv3 (bvh.rs:662-668)
  int kx, ky, kz;    // Woop permutation (triangle.rs:47-58)
  float sx, sy, sz;  // shear (triangle.rs:61-63)
  float okx, oky, okz;
};

__device__ __forceinline__ void setup_ray(RayCtx &r, bool with_shear) {
  const float TINY = 1e-20f, HUGE_ = 1e20f;
  r.ix = absf(r.dx) < TINY ? copysgn(HUGE_, r.dx) : 1.0f / r.dx;
  r.iy = absf(r.dy) < TINY ? copysgn(HUGE_, r.dy) : 1.0f / r.dy;
  r.iz = absf(r.dz) < TINY ? copysgn(HUGE_, r.dz) : 1.0f / r.dz;
  if (with_shear) {
    const float ax = absf(r.dx), ay = absf(r.dy), az = absf(r.dz);
    int kz;
    if (ax > ay) kz = (ax > az) ? 0 : 2;
    else if (ay > az) kz = 1;
    else kz = 2;
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;
    const float dkz = sel3(r.dx, r.dy, r.dz, kz);
    if (dkz < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.sx = sel3(r.dx, r.dy, r.dz, kx) / dkz;
    r.sy = sel3(r.dx, r.dy, r.dz, ky) / dkz;
    r.sz = 1.0f / dkz;
    r.okx = sel3(r.ox, r.oy, r.oz, kx);
    r.oky = sel3(r.ox, r.oy, r.oz, ky);
    r.okz = sel3(r.ox, r.oy, r.oz, kz);
  }
}

// Scalar watertight test with the f64 re-evaluation of exact-zero edge functions
// (triangle.rs:110-172). Used for the packet's fallback lanes only.
__device__ __forceinline__ bool tri_scalar(const RayCtx &r, const float *v /*9: v0 v1 v2*/, float t_min, float t_max,
                                           float &to, float &uo, float &vo) {
  const float a0 = v[0] - r.ox, a1 = v[1] - r.oy, a2 = v[2] - r.oz;
  const float b0 = v[3] - r.ox, b1 = v[4] - r.oy, b2 = v[5] - r.oz;
  const float c0 = v[6] - r.ox, c1 = v[7] - r.oy, c2 = v[8] - r.oz;
  const float akx = sel3(a0, a1, a2, r.kx), aky = sel3(a0, a1, a2, r.ky), akz = sel3(a0, a1, a2, r.kz);
  const float bkx = sel3(b0, b1, b2, r.kx), bky = sel3(b0, b1, b2, r.ky), bkz = sel3(b0, b1, b2, r.kz);
  const float ckx = sel3(c0, c1, c2, r.kx), cky = sel3(c0, c1, c2, r.ky), ckz = sel3(c0, c1, c2, r.kz);
  const float ax = akx - r.sx * akz, ay = aky - r.sy * akz;
  const float bx = bkx - r.sx * bkz, by = bky - r.sy * bkz;
  const float cx = ckx - r.sx * ckz, cy = cky - r.sy * ckz;
  float e0 = bx * cy - by * cx;
  float e1 = cx * ay - cy * ax;
  float e2 = ax * by - ay * bx;
  if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
    e0 = (float)((double)bx * (double)cy - (double)by * (double)cx);
    e1 = (float)((double)cx * (double)ay - (double)cy * (double)ax);
    e2 = (float)((double)ax * (double)by - (double)ay * (double)bx);
  }
  if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
  const float det = e0 + e1 + e2;
  if (det == 0.0f) return false;
  const float az = r.sz * akz, bz = r.sz * bkz, cz = r.sz * ckz;
  const float t_scaled = e0 * az + e1 * bz + e2 * cz;
  if (det < 0.0f && (t_scaled > t_min * det || t_scaled < t_max * det)) return false;
  if (det > 0.0f && (t_scaled < t_min * det || t_scaled > t_max * det)) return false;
  const float inv_det = 1.0f / det;
  to = t_scaled * inv_det;
  uo = e1 * inv_det;
  vo = e2 * inv_det;
  return true;
}

// Geometric or interpolated normal of a triangle primitive (prim.rs:76-95). The degenerate-sliver
// rejection has already happened through Tri4::normal_ok / the explicit check at the call site.
__device__ __forceinline__ void tri_normal(const DevScene &S, uint32_t prim_abs, float u, float v, float &nx, float &ny,
                                           float &nz) {
  const DevPrim *p = &S.prims[prim_abs];
  const uint32_t slot = __float_as_uint(p->d[9]);
  float x, y, z;
  if (slot != kInvalid) {
    const float *n = S.normals + 9 * (size_t)slot;
    const float w = 1.0f - u - v;
    x = (n[0] * w + n[3] * u) + n[6] * v;
    y = (n[1] * w + n[4] * u) + n[7] * v;
    z = (n[2] * w + n[5] * u) + n[8] * v;
  } else {
    const float e1x = p->d[3] - p->d[0], e1y = p->d[4] - p->d[1], e1z = p->d[5] - p->d[2];
    const float e2x = p->d[6] - p->d[0], e2y = p->d[7] - p->d[1], e2z = p->d[8] - p->d[2];
    x = e1y * e2z - e2y * e1z;
    y = e1z * e2x - e2z * e1x;
    z = e1x * e2y - e2x * e1y;
  }
  const float len = sqrtf(dot3(x, y, z, x, y, z));
  nx = x / len; ny = y / len; nz = z / len;
}

// glam Affine3A at a shutter time (prim.rs:285-331): lerp of the two placements, inverted.
__device__ __forceinline__ void motion_w2l(const DevInstanceMotion &in, float time, float w2l[12]) {
  float m[12];
#pragma unroll
  for (int i = 0; i < 12; i++) m[i] = in.l2w[i] * (1.0f - time) + in.l2w_end[i] * time;
  // Mat3A::inverse: cross products over the determinant, transposed.
  const float t0x = m[4] * m[8] - m[7] * m[5], t0y = m[5] * m[6] - m[8] * m[3], t0z = m[3] * m[7] - m[6] * m[4];
  const float t1x = m[7] * m[2] - m[1] * m[8], t1y = m[8] * m[0] - m[2] * m[6], t1z = m[6] * m[1] - m[0] * m[7];
  const float t2x = m[1] * m[5] - m[4] * m[2], t2y = m[2] * m[3] - m[5] * m[0], t2z = m[0] * m[4] - m[3] * m[1];
  const float det = dot3(m[6], m[7], m[8], t2x, t2y, t2z);
  const float inv = 1.0f / det;
  // columns of the inverse = rows of (t0, t1, t2) * inv
  w2l[0] = t0x * inv; w2l[1] = t1x * inv; w2l[2] = t2x * inv;
  w2l[3] = t0y * inv; w2l[4] = t1y * inv; w2l[5] = t2y * inv;
  w2l[6] = t0z * inv; w2l[7] = t1z * inv; w2l[8] = t2z * inv;
  // translation = -(inverse * t)
  float rx = w2l[0] * m[9], ry = w2l[1] * m[9], rz = w2l[2] * m[9];
  rx = rx + w2l[3] * m[10]; ry = ry + w2l[4] * m[10]; rz = rz + w2l[5] * m[10];
  rx = rx + w2l[6] * m[11]; ry = ry + w2l[7] * m[11]; rz = rz + w2l[8] * m[11];
  w2l[9] = -rx; w2l[10] = -ry; w2l[11] = -rz;
}

// InstancePrim::hit's normal step (prim.rs:327, :358-364): n <- normalize(w2l.matrix3^T * n), w2l the cached one or
// the shutter-time one (prim.rs:285-331).
__device__ __forceinline__ void instance_normal(const DevScene &S, uint32_t inst, float time, float &nx, float &ny, float &nz) {
  const DevInstance &in = S.instances[inst];
  float nm[9];
  if ((in.flags & 2u) && time > 0.0f) {
    float w2l[12];
    motion_w2l(*reinterpret_cast<const DevInstanceMotion *>(S.normals + (in.flags >> 2)), time, w2l);
    nm[0] = w2l[0]; nm[1] = w2l[3]; nm[2] = w2l[6];
    nm[3] = w2l[1]; nm[4] = w2l[4]; nm[5] = w2l[7];
    nm[6] = w2l[2]; nm[7] = w2l[5]; nm[8] = w2l[8];
  } else {
    nm[0] = in.w2l[0]; nm[1] = in.w2l[3]; nm[2] = in.w2l[6];  // cached normal matrix = w2l.matrix3 transposed
    nm[3] = in.w2l[1]; nm[4] = in.w2l[4]; nm[5] = in.w2l[7];
    nm[6] = in.w2l[2]; nm[7] = in.w2l[5]; nm[8] = in.w2l[8];
  }
  float x = nm[0] * nx, y = nm[1] * nx, z = nm[2] * nx;
  x = x + nm[3] * ny; y = y + nm[4] * ny; z = z + nm[5] * ny;
  x = x + nm[6] * nz; y = y + nm[7] * nz; z = z + nm[8] * nz;
  const float len = sqrtf(dot3(x, y, z, x, y, z));
  nx = x / len; ny = y / len; nz = z / len;
}

// Cooperative copy of the scene window into LDS: the top of the tree (the first nodes, breadth-first) and, in what
// the nodes leave free of the window's `cap` node slots, the first Tri4 packets whole (192 B each: Woop vertices,
// primitive ids, masks) — on a small scene (veach_mis: 12 triangles) that is every node and every packet, so a ray
// never leaves LDS until it reports its hit. Returns the nodes staged, n_pk the packets. Ends with a barrier.
constexpr int kLdsPacketDwords = 48;
#ifndef CRT_LDS_NODE_MAX
#define CRT_LDS_NODE_MAX 1000000
#endif
__device__ __forceinline__ uint32_t stage_nodes(const DevScene &S, uint32_t *lds_nodes, int cap, uint32_t &n_pk) {
  // CRT_LDS_NODE_MAX: at most this many nodes are staged, the rest of the window goes to the packets (which the upload
  // orders hottest first, scene.cpp). Beyond the first two or three levels a node is visited by few rays, while the
  // packets of a room's walls are tested by most of them.
  uint32_t n = S.n_nodes < (uint32_t)cap ? S.n_nodes : (uint32_t)cap;
  if (S.n_packets > 0 && n > (uint32_t)CRT_LDS_NODE_MAX) n = (uint32_t)CRT_LDS_NODE_MAX;
  const uint32_t room = ((uint32_t)cap - n) * (uint32_t)kLdsNodeStride / (uint32_t)kLdsPacketDwords;
  n_pk = S.n_packets < room ? S.n_packets : room;
  for (uint32_t w = threadIdx.x; w < n * 8u; w += blockDim.x) {  // 7 of the 8 x 16 bytes of a node
    const uint32_t node = w >> 3, part = w & 7u;
    if (part == 7u) continue;
    const float4 v = reinterpret_cast<const float4 *>(S.nodes + node)[part];
    *reinterpret_cast<float4 *>(lds_nodes + (size_t)node * kLdsNodeStride + part * 4) = v;
  }
  float4 *pk_dst = reinterpret_cast<float4 *>(lds_nodes + (size_t)n * kLdsNodeStride);
  const float4 *pk_src = reinterpret_cast<const float4 *>(S.packets);
  for (uint32_t w = threadIdx.x; w < n_pk * 12u; w += blockDim.x) pk_dst[w] = pk_src[w];
  __syncthreads();
  return n;
}

struct Frame {
  float ox, oy, oz, dx, dy, dz;
  uint32_t cursor, cend, base, inst, geom, has_packets;
};

// One ray handed to a lane by the fetch callback of traverse_stream.
struct RayIn {
  float ox, oy, oz, dx, dy, dz, time, t_min, t_max;
  uint32_t mask;
  uint32_t slot;  // caller's tag, passed back to emit
};

}  // namespace dev
}  // namespace crt
