// BVH4 closest-hit / any-hit traversal for gfx950 (wave64), one ray per lane.
//
// What it computes: crates/crust-rt/src/bvh.rs:441-509 (Bvh::hit), :585-611 (hit_any), :514-572
// (intersect_leaf), :617-653 (occlude_leaf), :790-808 (slab4), :662-668 (safe_inv3);
// triangle.rs:41-80 (RayShear), :276-348 (Tri4::intersect), :110-172 (f64 edge fallback);
// prim.rs:76-95, :133-161, :321-378 (triangle normal, sphere, instance); scene.rs:354-366.
//
// How it is laid out for CDNA4:
//  * one 128-byte node = one cache line, fetched by a lane as 8 x 16-byte loads; a 192-byte Tri4
//    packet as 9 x 16-byte plane loads addressed by the ray's permuted axes (no per-lane selects);
//  * the traversal stack lives in LDS, entry k of thread t at lds[k * BLOCK + t] (bank = lane, so a
//    wave's push/pop is conflict-free); entries past the LDS depth spill to per-lane scratch;
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
constexpr int kStackLds = 16;     // stack entries per lane kept in LDS (16 KiB per workgroup)
constexpr int kStackSpill = 240;  // further entries per lane in scratch: 3 * MAX_DEPTH + 4 fits
// Top of the tree staged in LDS: the first kLdsNodes nodes (breadth-first numbering) as 144-byte records
// (128 + 16 pad: a 36-dword stride spreads the lanes' 16-byte reads over all banks). 36 KiB per workgroup.
constexpr int kLdsNodes = 256;
constexpr int kLdsNodeStride = 36;  // dwords
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
__device__ __forceinline__ void motion_w2l(const DevInstance &in, float time, float w2l[12]) {
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

// Cooperative copy of the top-of-tree window into LDS; returns the number of nodes staged. Ends with a barrier.
__device__ __forceinline__ uint32_t stage_nodes(const DevScene &S, uint32_t *lds_nodes, int cap = kLdsNodes) {
  const uint32_t n = S.n_nodes < (uint32_t)cap ? S.n_nodes : (uint32_t)cap;
  for (uint32_t w = threadIdx.x; w < n * 8u; w += blockDim.x) {  // 8 x 16 bytes per node
    const uint32_t node = w >> 3, part = w & 7u;
    const float4 v = reinterpret_cast<const float4 *>(S.nodes + node)[part];
    *reinterpret_cast<float4 *>(lds_nodes + (size_t)node * kLdsNodeStride + part * 4) = v;
  }
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

#ifndef CRT_REFILL
#define CRT_REFILL 32
#endif
constexpr int kRefillIdle = CRT_REFILL;  // refill a wave once this many of its 64 lanes have no ray

// Persistent-wave traversal: a wave keeps pulling rays until its source is dry. Whenever enough lanes have
// finished, they fetch new rays while the others keep their traversal state, so one long ray does not hold 63
// idle lanes hostage (incoherent secondary rays finish after very different numbers of steps).
//   fetch(want, ray) -> bool : called by the whole wave; lanes with want==true may receive a ray
//   emit(slot, hit?, Hit)    : called by a lane whose ray is finished (ANY: hit? means occluded)
template <bool ANY, bool STATS, class Fetch, class Emit>
__device__ void traverse_stream(const DevScene &S, uint32_t *lds /* &stack[tid] */,
                                const uint32_t *lds_nodes /* staged window */, uint32_t n_lds, uint32_t &err,
                                LaneStats &st, Fetch fetch, Emit emit) {
  float time = 0.0f, t_min = 0.0f, t_max = 0.0f;
  uint32_t rmask = 0, slot = 0;

  uint32_t spill[kStackSpill];
  Frame frames[kMaxLevels];
  // Logical stack = memory slots [0, sp-1) plus `top` in a register (valid when sp > 0): a push spills the
  // old top, a pop reloads the next one early, so the LDS latency of the reload hides behind the work on the
  // entry just popped.
  int sp = 0;
  uint32_t top = kInvalid;
  auto mem_store = [&](int slot, uint32_t x) {
    if (slot < kStackLds) lds[slot * kBlock] = x;
    else if (slot - kStackLds < kStackSpill) spill[slot - kStackLds] = x;
    else err |= 1u;
  };
  auto mem_load = [&](int slot) -> uint32_t {
    if (slot < kStackLds) return lds[slot * kBlock];
    if (slot - kStackLds < kStackSpill) return spill[slot - kStackLds];
    return kLeafTag | kInvalid;  // only after an overflow (err already set): harmless empty leaf tag
  };
  auto push = [&](uint32_t x) {
    if (sp > 0) mem_store(sp - 1, top);
    top = x;
    sp++;
  };
  auto pop = [&]() -> uint32_t {
    const uint32_t x = top;
    sp--;
    if (sp > 0) top = mem_load(sp - 1);
    return x;
  };

  RayCtx r = {};
  uint32_t cur_has_packets = S.has_packets;
  float closest = 0.0f;
  float bt = 0.0f, bu = 0.0f, bv = 0.0f, bnx = 0.0f, bny = 0.0f, bnz = 0.0f;
  uint32_t bgeom = kInvalid, bprim = kInvalid, bdefer = kInvalid;  // bdefer: triangle whose normal is pending
  uint32_t found = 0;  // bit L: level L holds a hit
  int level = 0;
  uint32_t base = 0, cursor = 0, cend = 0;
  auto begin = [&](const RayIn &in) {
    r.ox = in.ox; r.oy = in.oy; r.oz = in.oz; r.dx = in.dx; r.dy = in.dy; r.dz = in.dz;
    time = in.time; t_min = in.t_min; t_max = in.t_max; rmask = in.mask; slot = in.slot;
    cur_has_packets = S.has_packets;
    setup_ray(r, cur_has_packets != 0);
    closest = t_max;
    bgeom = kInvalid; bprim = kInvalid; bdefer = kInvalid;
    found = 0; level = 0; base = 0; cursor = 0; cend = 0;
    sp = 0; top = kInvalid;
    if (STATS) st.queries[0]++;
    if (S.root != kInvalid) push(S.root);  // an empty scene finishes at once (bvh.rs:442-444)
  };
  // A triangle hit keeps only its primitive index until its tree is finished; ids and the normal are
  // derived once, here (prim.rs:76-95).
  auto finalize = [&]() {
    if (bdefer != kInvalid) {
      const DevPrim *p = &S.prims[bdefer];
      bgeom = p->geom_id;
      bprim = p->prim_id;
      tri_normal(S, bdefer, bu, bv, bnx, bny, bnz);
      bdefer = kInvalid;
    }
  };

  // ---- inner node: 4-wide slab test (bvh.rs:790-808); pushes the hit lanes ----
  auto expand_node = [&](uint32_t e) {
    if (STATS) st.nodes[level > 0 ? 1 : 0]++;
    float4 mnx, mny, mnz, mxx, mxy, mxz;
    uint4 ch;
    uint32_t flags;
    if (e < n_lds) {  // top of the tree: LDS (ds_read_b128), no trip through the vector memory pipeline
      const float4 *nb = reinterpret_cast<const float4 *>(lds_nodes + (size_t)e * kLdsNodeStride);
      mnx = nb[0]; mny = nb[1]; mnz = nb[2]; mxx = nb[3]; mxy = nb[4]; mxz = nb[5];
      ch = *reinterpret_cast<const uint4 *>(nb + 6);
      flags = lds_nodes[(size_t)e * kLdsNodeStride + 28];
    } else {
      const WideNode *nd = &S.nodes[e];
      const float4 *nb = reinterpret_cast<const float4 *>(nd);
      mnx = nb[0]; mny = nb[1]; mnz = nb[2]; mxx = nb[3]; mxy = nb[4]; mxz = nb[5];
      ch = *reinterpret_cast<const uint4 *>(nd->child);
      flags = nd->flags;
    }
    const float lo_x[4] = {mnx.x, mnx.y, mnx.z, mnx.w}, lo_y[4] = {mny.x, mny.y, mny.z, mny.w},
                lo_z[4] = {mnz.x, mnz.y, mnz.z, mnz.w};
    const float hi_x[4] = {mxx.x, mxx.y, mxx.z, mxx.w}, hi_y[4] = {mxy.x, mxy.y, mxy.z, mxy.w},
                hi_z[4] = {mxz.x, mxz.y, mxz.z, mxz.w};
    const uint32_t child[4] = {ch.x, ch.y, ch.z, ch.w};
    const float bound = ANY ? t_max : closest;
    float key[4];
    uint32_t ent[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
      const float t0x = (lo_x[l] - r.ox) * r.ix, t1x = (hi_x[l] - r.ox) * r.ix;
      const float t0y = (lo_y[l] - r.oy) * r.iy, t1y = (hi_y[l] - r.oy) * r.iy;
      const float t0z = (lo_z[l] - r.oz) * r.iz, t1z = (hi_z[l] - r.oz) * r.iz;
      const float tn = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), t_min);
      const float tf = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)), bound);
      const bool on = (tn <= tf) && ((flags >> l) & 1u);
      key[l] = tn;
      ent[l] = on ? (child[l] | (((flags >> (4 + l)) & 1u) ? kLeafTag : 0u)) : kInvalid;  // kInvalid = lane off
    }
    if (ANY) {
#pragma unroll
      for (int l = 0; l < 4; l++)
        if (ent[l] != kInvalid) push(ent[l]);
      return;
    }
    // Stable insertion sort of the hit lanes by entry distance (bvh.rs:472-486). Lanes that are off
    // sort as +inf keys and are skipped at push time; relative order of the hit lanes is the reference's.
#pragma unroll
    for (int l = 0; l < 4; l++)
      if (ent[l] == kInvalid) key[l] = __builtin_inff();
    // Off lanes must not overtake hit lanes with an infinite key: give them a strictly-last rank by
    // sorting on (key, off) pairs — `off` breaks the tie.
    auto after = [&](int a, int b) {  // does slot a sort strictly after slot b?
      const bool offa = ent[a] == kInvalid, offb = ent[b] == kInvalid;
      return (key[a] > key[b]) || (key[a] == key[b] && offa && !offb);
    };
    auto swp = [&](int a, int b) {
      const float k = key[a]; key[a] = key[b]; key[b] = k;
      const uint32_t x = ent[a]; ent[a] = ent[b]; ent[b] = x;
    };
    if (after(0, 1)) swp(0, 1);
    if (after(1, 2)) { swp(1, 2); if (after(0, 1)) swp(0, 1); }
    if (after(2, 3)) { swp(2, 3); if (after(1, 2)) { swp(1, 2); if (after(0, 1)) swp(0, 1); } }
    // Inner lanes far to near, then leaf lanes far to near on top (bvh.rs:488-505, see header).
#pragma unroll
    for (int i = 3; i >= 0; i--)
      if (ent[i] != kInvalid && !(ent[i] & kLeafTag)) push(ent[i]);
#pragma unroll
    for (int i = 3; i >= 0; i--)
      if (ent[i] != kInvalid && (ent[i] & kLeafTag)) push(ent[i]);
  };

  // One scheduling step of the lane's ray: 0 = keep going, 1 = finished; ANY only: 2..5 = occluded, one code
  // per kind of occluder. The codes are deliberately distinct: with four identical `return 2` sites hipcc
  // (ROCm 7.2, gfx950) merged the exits and the occluded kernels reported 210 of 4096 missing rays of the
  // `mixed` test scene as occluded; distinct exit values keep the exits apart (tests/test_gpu_traverse.py
  // ::test_intersect_occluded_match_oracle_bitwise[mixed] is the regression test).
  auto step = [&]() -> int {
    // ---- continue a leaf's one-at-a-time primitives (bvh.rs:564-570 / :646-651) ----
    if (cursor < cend) {
      CRT_PHASE(4)
      const uint32_t pi = S.indices[cursor++];
      const DevPrim *p = &S.prims[pi];
      const uint4 hd = *reinterpret_cast<const uint4 *>(p);  // kind, geom_id, prim_id, mask
      if ((rmask & hd.w) == 0) return 0;                     // prim.rs:52-54
      if (hd.x == PRIM_SPHERE) {                             // prim.rs:133-161
        const float4 s = *reinterpret_cast<const float4 *>(p->d);
        const float ocx = r.ox - s.x, ocy = r.oy - s.y, ocz = r.oz - s.z;
        const float a = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
        const float half_b = dot3(ocx, ocy, ocz, r.dx, r.dy, r.dz);
        const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - s.w * s.w;
        const float disc = half_b * half_b - a * c;
        if (disc < 0.0f) return 0;
        const float sqrt_d = sqrtf(disc);
        float root = (-half_b - sqrt_d) / a;
        if (root <= t_min || root >= closest) {
          root = (-half_b + sqrt_d) / a;
          if (root <= t_min || root >= closest) return 0;
        }
        if (ANY) return 2;  // occluded by a sphere
        closest = root;
        bt = root; bu = 0.0f; bv = 0.0f;
        bnx = ((r.ox + root * r.dx) - s.x) / s.w;
        bny = ((r.oy + root * r.dy) - s.y) / s.w;
        bnz = ((r.oz + root * r.dz) - s.z) / s.w;
        bgeom = hd.y; bprim = 0u; bdefer = kInvalid;
        found |= 1u << level;
        if (STATS) st.accepted++;
      } else if (hd.x == PRIM_INSTANCE) {  // prim.rs:345-378
        if (level + 1 >= kMaxLevels) { err |= 2u; return 0; }
        const DevInstance *in = &S.instances[__float_as_uint(p->d[0])];
        float w2l[12];
        if (in->has_end && time > 0.0f) motion_w2l(*in, time, w2l);
        else {
#pragma unroll
          for (int i = 0; i < 12; i++) w2l[i] = in->w2l[i];
        }
        Frame &f = frames[level];
        f.ox = r.ox; f.oy = r.oy; f.oz = r.oz; f.dx = r.dx; f.dy = r.dy; f.dz = r.dz;
        f.cursor = cursor; f.cend = cend; f.base = base; f.inst = __float_as_uint(p->d[0]); f.geom = hd.y;
        f.has_packets = cur_has_packets;
        // transform_point3a / transform_vector3a: ((x_axis*v.x + y_axis*v.y) + z_axis*v.z) [+ translation]
        float px = w2l[0] * r.ox, py = w2l[1] * r.ox, pz = w2l[2] * r.ox;
        px = px + w2l[3] * r.oy; py = py + w2l[4] * r.oy; pz = pz + w2l[5] * r.oy;
        px = px + w2l[6] * r.oz; py = py + w2l[7] * r.oz; pz = pz + w2l[8] * r.oz;
        px = px + w2l[9]; py = py + w2l[10]; pz = pz + w2l[11];
        float qx = w2l[0] * r.dx, qy = w2l[1] * r.dx, qz = w2l[2] * r.dx;
        qx = qx + w2l[3] * r.dy; qy = qy + w2l[4] * r.dy; qz = qz + w2l[5] * r.dy;
        qx = qx + w2l[6] * r.dz; qy = qy + w2l[7] * r.dz; qz = qz + w2l[8] * r.dz;
        r.ox = px; r.oy = py; r.oz = pz; r.dx = qx; r.dy = qy; r.dz = qz;  // unnormalised: local t == world t
        level++;
        found &= ~(1u << level);
        base = (uint32_t)sp;
        cursor = cend = 0;
        cur_has_packets = in->has_packets;
        setup_ray(r, cur_has_packets != 0);
        if (STATS) { st.descents++; st.queries[1]++; }
        push(in->root);
      } else {
        // A triangle on the scalar list (the builder always packs triangles; kept for completeness).
        float t, u, v;
        RayCtx rr = r;
        if (!cur_has_packets) setup_ray(rr, true);
        if (!tri_scalar(rr, p->d, t_min, closest, t, u, v)) return 0;
        if (ANY) return 3;  // occluded by a scalar-list triangle
        const float e1x = p->d[3] - p->d[0], e1y = p->d[4] - p->d[1], e1z = p->d[5] - p->d[2];
        const float e2x = p->d[6] - p->d[0], e2y = p->d[7] - p->d[1], e2z = p->d[8] - p->d[2];
        const bool flat = (e1y * e2z - e2y * e1z) == 0.0f && (e1z * e2x - e2z * e1x) == 0.0f &&
                          (e1x * e2y - e2x * e1y) == 0.0f;
        if (flat && __float_as_uint(p->d[9]) == kInvalid) return 0;
        closest = t; bt = t; bu = u; bv = v; bdefer = pi;
        found |= 1u << level;
        if (STATS) st.accepted++;
      }
      return 0;
    }

    // ---- this tree is exhausted ----
    if ((uint32_t)sp == base) {
      if (level == 0) return 1;
      CRT_PHASE(5)
      const bool inner_found = (found >> level) & 1u;
      level--;
      const Frame &f = frames[level];
      if (inner_found) {  // prim.rs:358-364
        finalize();
        const DevInstance *in = &S.instances[f.inst];
        float nm[9];
        if (in->has_end && time > 0.0f) {
          float w2l[12];
          motion_w2l(*in, time, w2l);
          // normal matrix = w2l.matrix3 transposed (prim.rs:327)
          nm[0] = w2l[0]; nm[1] = w2l[3]; nm[2] = w2l[6];
          nm[3] = w2l[1]; nm[4] = w2l[4]; nm[5] = w2l[7];
          nm[6] = w2l[2]; nm[7] = w2l[5]; nm[8] = w2l[8];
        } else {
#pragma unroll
          for (int i = 0; i < 9; i++) nm[i] = in->nmat[i];
        }
        float x = nm[0] * bnx, y = nm[1] * bnx, z = nm[2] * bnx;
        x = x + nm[3] * bny; y = y + nm[4] * bny; z = z + nm[5] * bny;
        x = x + nm[6] * bnz; y = y + nm[7] * bnz; z = z + nm[8] * bnz;
        const float len = sqrtf(dot3(x, y, z, x, y, z));
        bnx = x / len; bny = y / len; bnz = z / len;
        bgeom = f.geom;  // the hit is attributed to the instance's geometry id; prim_id stays the inner one
        found |= 1u << level;
        if (STATS) st.accepted++;
      }
      r.ox = f.ox; r.oy = f.oy; r.oz = f.oz; r.dx = f.dx; r.dy = f.dy; r.dz = f.dz;
      cursor = f.cursor; cend = f.cend; base = f.base;
      cur_has_packets = f.has_packets;
      setup_ray(r, cur_has_packets != 0);
      return 0;
    }

    // ---- node phase ("while-while"): expand inner nodes until a leaf entry surfaces or this tree's stack
    // is empty. Keeping the lanes of a wave in the same phase is what keeps the SIMD lanes busy: a wave pays
    // max-over-lanes node steps, then max-over-lanes leaf steps, instead of (node + leaf) per step.
    uint32_t e = kInvalid;
    bool have_leaf = false;
    while ((uint32_t)sp > base) {
      e = pop();
      if (e & kLeafTag) { have_leaf = true; break; }
      CRT_PHASE(2)
      expand_node(e);
    }
    if (!have_leaf) return 0;  // exhausted: the check at the top of the loop returns or finishes
    const int sl = level > 0 ? 1 : 0;

    // ---- leaf: 4-wide packets first, then the scalar list (bvh.rs:514-572) ----
    {
      const uint32_t li = e & ~kLeafTag;
      if (li == (kInvalid & ~kLeafTag)) return 0;
      const Leaf lf = S.leaves[li];
      if (STATS) { st.leaves[sl]++; st.packets[sl] += lf.pkt_count; st.prims[sl] += lf.idx_count; }
      for (uint32_t k = 0; k < lf.pkt_count; k++) {
        const Tri4 *pk = &S.packets[lf.pkt_first + k];
        const uint4 meta = *reinterpret_cast<const uint4 *>(&pk->active);  // active, mask_and, mask_or, masks[0]
        uint32_t m;                                                         // triangle.rs:257-271
        if (rmask & meta.y) m = meta.x;
        else if ((rmask & meta.z) == 0) m = 0;
        else {
          m = 0;
#pragma unroll
          for (int l = 0; l < 4; l++)
            if ((meta.x & (1u << l)) && (pk->masks[l] & rmask)) m |= 1u << l;
        }
        if (m == 0) continue;
        CRT_PHASE(3)
        // 9 plane loads, addressed by the permuted axes: v[vertex][axis][0..3]
        const float4 *pl = reinterpret_cast<const float4 *>(&pk->v[0][0][0]);
        const float4 A_x = pl[0 + r.kx], A_y = pl[0 + r.ky], A_z = pl[0 + r.kz];
        const float4 B_x = pl[3 + r.kx], B_y = pl[3 + r.ky], B_z = pl[3 + r.kz];
        const float4 C_x = pl[6 + r.kx], C_y = pl[6 + r.ky], C_z = pl[6 + r.kz];
        const float vax[4] = {A_x.x, A_x.y, A_x.z, A_x.w}, vay[4] = {A_y.x, A_y.y, A_y.z, A_y.w},
                    vaz[4] = {A_z.x, A_z.y, A_z.z, A_z.w};
        const float vbx[4] = {B_x.x, B_x.y, B_x.z, B_x.w}, vby[4] = {B_y.x, B_y.y, B_y.z, B_y.w},
                    vbz[4] = {B_z.x, B_z.y, B_z.z, B_z.w};
        const float vcx[4] = {C_x.x, C_x.y, C_x.z, C_x.w}, vcy[4] = {C_y.x, C_y.y, C_y.z, C_y.w},
                    vcz[4] = {C_z.x, C_z.y, C_z.z, C_z.w};
        uint32_t fallback = 0, hits = 0;
        float ht[4], hu[4], hv[4];
        const float entry_closest = closest;  // every lane range-tests against the packet-entry bound
#pragma unroll
        for (int l = 0; l < 4; l++) {  // triangle.rs:284-347, one SIMD lane at a time
          const float akz = vaz[l] - r.okz, bkz = vbz[l] - r.okz, ckz = vcz[l] - r.okz;
          const float ax = (vax[l] - r.okx) - r.sx * akz, ay = (vay[l] - r.oky) - r.sy * akz;
          const float bx = (vbx[l] - r.okx) - r.sx * bkz, by = (vby[l] - r.oky) - r.sy * bkz;
          const float cx = (vcx[l] - r.okx) - r.sx * ckz, cy = (vcy[l] - r.oky) - r.sy * ckz;
          const float e0 = bx * cy - by * cx;
          const float e1 = cx * ay - cy * ax;
          const float e2 = ax * by - ay * bx;
          const bool zero = (e0 == 0.0f) | (e1 == 0.0f) | (e2 == 0.0f);
          const bool neg = (e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f);
          const bool pos = (e0 > 0.0f) | (e1 > 0.0f) | (e2 > 0.0f);
          const float det = e0 + e1 + e2;
          const float t_scaled = e0 * (r.sz * akz) + e1 * (r.sz * bkz) + e2 * (r.sz * ckz);
          const float abs_det = absf(det);
          const float ts = det < 0.0f ? -t_scaled : t_scaled;
          const bool in_range = (ts >= t_min * abs_det) & (ts <= entry_closest * abs_det);
          const bool lane_on = (m >> l) & 1u;
          if (lane_on && zero) fallback |= 1u << l;
          if (lane_on && !zero && !(neg && pos) && det != 0.0f && in_range) hits |= 1u << l;
          const float inv_det = 1.0f / det;
          ht[l] = t_scaled * inv_det;
          hu[l] = e1 * inv_det;
          hv[l] = e2 * inv_det;
        }
        if (ANY) {
          if (hits) return 4;  // occluded by a packet lane
        } else {
#pragma unroll
          for (int l = 0; l < 4; l++) {  // bvh.rs:533-550
            if (!((hits >> l) & 1u)) continue;
            if (ht[l] > closest) continue;            // strict: an exact tie goes to the later lane
            if (!((pk->normal_ok >> l) & 1u)) continue;  // prim.rs:81-83 degenerate sliver
            closest = ht[l]; bt = ht[l]; bu = hu[l]; bv = hv[l];
            bdefer = pk->prim[l];
            found |= 1u << level;
            if (STATS) st.accepted++;
          }
        }
        if (fallback) {  // bvh.rs:551-561 / :636-643 — lanes sitting exactly on an edge
#pragma unroll
          for (int l = 0; l < 4; l++) {
            if (!((fallback >> l) & 1u)) continue;
            const uint32_t pi = pk->prim[l];
            const DevPrim *p = &S.prims[pi];
            if ((rmask & p->mask) == 0) continue;
            CRT_PHASE(7)
            float t, u, v;
            if (!tri_scalar(r, p->d, t_min, closest, t, u, v)) continue;
            if (ANY) return 5;  // occluded by an on-edge (f64 fallback) lane
            if (!((pk->normal_ok >> l) & 1u)) continue;
            closest = t; bt = t; bu = u; bv = v; bdefer = pi;
            found |= 1u << level;
            if (STATS) st.accepted++;
          }
        }
      }
      cursor = lf.idx_first;
      cend = lf.idx_first + lf.idx_count;
      return 0;
    }

    return 0;
  };

  bool active = false;
  bool more = true;  // wave-uniform: the source may still hold rays
  for (;;) {
    const unsigned long long act = __ballot(active);
    if (more && (act == 0 || 64 - __popcll(act) >= kRefillIdle)) {
      RayIn in;
      if (fetch(!active, in)) {
        CRT_PHASE(1)
        begin(in);
        active = true;
      }
      if (__ballot(!active)) more = false;  // a lane asked and got nothing: the source is dry
    }
    if (!__ballot(active)) break;
    if (active) {
      CRT_PHASE(0)
      const int rc = step();
      if (rc) {
        CRT_PHASE(6)
        Hit hit;
        bool is_hit;
        if (ANY) {
          is_hit = rc >= 2;
        } else {
          is_hit = (found & 1u) != 0;
          if (is_hit) {
            finalize();
            hit.t = bt; hit.u = bu; hit.v = bv; hit.nx = bnx; hit.ny = bny; hit.nz = bnz; hit.geom = bgeom; hit.prim = bprim;
          }
        }
        emit(slot, is_hit, hit);
        active = false;
      }
    }
  }
}

}  // namespace dev
}  // namespace crt
