// BVH4 traversal, phase-scheduled: every lane of a wave owns ROWS rays whose state lives in LDS, and in each
// step the whole wave executes ONE kind of work — node expansion, packet test, scalar primitive, instance exit,
// emit, or fetch — for the lanes that have a ray waiting for exactly that.
//
// Why: with one ray per lane (the first engine of this repository, commit 51c24ac) incoherent rays sit in different phases of their traversal, and the
// wave pays for every phase with the few lanes that are in it: measured on cornellbox 1080p, 35 % of the lanes are
// live in a node expansion, 38 % in a packet test, 5 % in an instance entry or exit (CrtTravStats::phase_lanes).
// Replaying the oracle's per-ray work traces through both schedulers (profiles/simulate_scheduling.py) predicts
// 2.2x fewer wave instructions per bounce ray for two rays per lane. The per-ray sequence of operations — node
// order, leaf order, packet order, accept rule — is untouched, so results stay bit-identical to the reference
// restatement; only WHEN a ray advances changes.
//
// Layout: slot (row, lane) is only ever touched by that lane, so every LDS access of a wave is unit-stride
// (16 bytes per lane for the state groups, 4 for a stack entry) and conflict-free, and no gather list is needed.
// State per ray in LDS: five 16-byte groups + kPoolStack stack entries. Rarely used state (direction, time,
// pending normal, instance frames, deep stack) sits in per-lane private memory indexed by row.
#pragma once

#include "traverse.hip.h"

namespace crt {
namespace dev {

#ifndef CRT_ROOT_REJECT
#define CRT_ROOT_REJECT 1  // an instanced tree's root node is tested at entry; rays that touch no child never enter
#endif
#ifndef CRT_DEFER_NORMAL
#define CRT_DEFER_NORMAL 1  // a level-1 instanced hit's normal is taken to world space at emit, not at every exit
#endif
#ifndef CRT_LDS_ROOT
#define CRT_LDS_ROOT 1  // the root test at entry reads a root staged in the LDS window from LDS
#endif
#ifndef CRT_EMIT_REFILL
#define CRT_EMIT_REFILL 0  // measured: bench -10 % (39 spilled registers in the four-wave kernel), three-wave kernels +-0
#endif
#ifndef CRT_FETCH_MIN
#define CRT_FETCH_MIN 40
#endif
#ifndef CRT_NODE_SWAP
// 1: the node loop hands a lane over to its other ray when the current one leaves the phase (round 4). Measured, Mray/s
// off / on: cornellbox 10360 / 10258, stress 2557 / 2485, PointInstancedMedCity 2251 / 2198, the 7 M-triangle scene 3451 /
// 3296, veach_mis 12150 / 12264: the loop runs fuller and longer, and the phases that wait for it pay more than it saves.
#define CRT_NODE_SWAP 0
#endif
#ifndef CRT_STACK_RING
// 1: the LDS part of a ray's stack holds its TOP (a ring of pstack slots), not its bottom (round 4). Measured, Mray/s
// off / on: cornellbox 10273 / 10169, stress 2498 / 2438 (four-wave 2648 / 2546; six-entry split 2249 / 2111),
// PointInstancedMedCity 2224 / 2182, the 7 M-triangle scene 3384 / 3334: what a shallow LDS part costs is not the
// latency of its private-memory entries but the instructions of the path that handles them (CRT_SPILL_UNROLLED).
#define CRT_STACK_RING 0
#endif
#ifndef CRT_SPILL_UNROLLED
#define CRT_SPILL_UNROLLED 1  // a node step below the LDS part of the stack stores its entries with predicated writes, not in a loop
#endif
#ifndef CRT_RARE_MIN
#define CRT_RARE_MIN 24
#endif
// Stack entries per ray kept in LDS (deeper entries go to private memory) and nodes of the top of the tree staged in
// LDS are two sides of one LDS budget, and the best split depends on the scene: a single-level scene rarely goes
// deeper than six entries and profits from the node window; instance-heavy scenes stack the parent's entries under
// the instance's and run +6 % with ten entries and next to no window (profiles/README.md). The host picks the split
// per scene (DevScene::pool_stack); both fit the same arena.
constexpr int kPoolStack = CRT_POOL_STACK;       // flat scenes (the three CRT_POOL_STACK* defaults: crt_internal.h)
constexpr int kPoolStackDeep = CRT_POOL_STACK_DEEP;  // instance-heavy scenes
// Private part of the stack. sp lives in 8 bits of the ctl word, so LDS part + private part must not exceed 255 with
// EITHER split: a push at the limit then takes the err path (CRT_ERR_STACK) instead of wrapping into `base`.
constexpr int kPoolSpill = 255 - (kPoolStack > kPoolStackDeep ? kPoolStack : kPoolStackDeep);
static_assert(kPoolStack + kPoolSpill <= 255 && kPoolStackDeep + kPoolSpill <= 255, "sp must fit the ctl word's 8 bits");
constexpr int kFetchMin = CRT_FETCH_MIN;     // fetch new rays once this many lanes have a free slot
#ifndef CRT_EMIT_BIAS
#define CRT_EMIT_BIAS 16
#endif
#ifndef CRT_SCALAR_BIAS
#define CRT_SCALAR_BIAS 0
#endif
#ifndef CRT_EXIT_BIAS
#define CRT_EXIT_BIAS 0
#endif
#ifndef CRT_PACKET_BIAS
#define CRT_PACKET_BIAS 0
#endif
constexpr int kPacketBias = CRT_PACKET_BIAS;  // node steps are preferred until the packet group leads by more than this
constexpr int kEmitBias = CRT_EMIT_BIAS, kScalarBias = CRT_SCALAR_BIAS, kExitBias = CRT_EXIT_BIAS;
constexpr int kRareMin = CRT_RARE_MIN;       // run a rare phase (scalar prim / instance exit / emit) at this many lanes
#ifndef CRT_STICKY_MIN
#define CRT_STICKY_MIN 24
#endif
constexpr int kStickyMin = CRT_STICKY_MIN;   // keep repeating node / packet steps while this many lanes stay in the phase

// LDS dwords one wave needs for ROWS rays per lane.
// UMASK: the launch's rays all carry one ray mask (the renderer's kernels: camera / indirect / shadow), so the mask plane
// is not kept — 4 bytes per ray, one more stack entry in the same arena.
template <int ROWS>
constexpr int pool_lds_dwords(int stack, bool umask = false) { return (4 + 4 + 2 + (umask ? 4 : 5) + stack) * ROWS * 64; }

enum : uint32_t { PH_FREE = 0, PH_NODE = 1, PH_PACKET = 2, PH_SCALAR = 3, PH_EXIT = 4, PH_EMIT = 5 };

// ctl word: sp[0:8) base[8:16) level[16:19) has_packets[19] kz[20:22) swap[22] lo[24:32) — lo: how many of the stack's
// bottom entries have moved out of LDS (ring form of the stack, CRT_STACK_RING)
// aux word: found-in-level bits [0:8) (bit 0 doubles as "hit" / "occluded"), scalar entries left in the leaf [8:32)
__device__ __forceinline__ uint32_t ctl_pack(uint32_t sp, uint32_t base, uint32_t level, uint32_t hp, uint32_t kz,
                                             uint32_t swap, uint32_t lo) {
  return sp | (base << 8) | (level << 16) | (hp << 19) | (kz << 20) | (swap << 22) | (lo << 24);
}
// the control word with a new stack position
__device__ __forceinline__ uint32_t ctl_sp(uint32_t c, uint32_t sp, uint32_t lo) { return (c & 0x00ffff00u) | sp | (lo << 24); }

//   fetch(want, ray) -> bool : called by the whole wave; lanes with want==true may receive a ray
//   emit(slot, hit?, Hit)    : called by a lane whose ray is finished (ANY: hit? means occluded)
// t_min is uniform over the launch (every caller passes one value).
//
// LDS image of one wave (dwords, R = ROWS): [0, 4R*64) F0 = origin.xyz + closest (16 B per slot);
// [4R*64, 8R*64) F1 = inv_dir.xyz + shear.x; then 2R*64 F2 = shear.y, shear.z (8 B per slot); then five dword
// planes of R*64: ctl, cur, aux, cursor, mask; then the stack, kPoolStack planes. 60 + 4*kPoolStack bytes per ray.
// The hit found so far (u, v, primitive, geometry) and the caller's slot tag are written once or twice per ray and
// read at the end: they live in private memory with the other rarely used state.
// Lanes of the wave for which p holds, as a 32-bit scalar. The empty asm hides that the value came from a 64-bit
// population count: otherwise the comparisons below are carried out as 64-bit VECTOR compares (there is no scalar
// ordered compare for 64 bits) with a register-pair move each, in the scheduler's every round.
__device__ __forceinline__ int wave_count(bool p) {
  int n = (int)__popcll(__ballot(p));
  asm volatile("" : "+s"(n));
  return n;
}

// COLD: which of the rarely used per-ray fields this instantiation keeps at all (DevScene::cold, crt_internal.h) —
// kColdUV the hit's barycentrics (read only by smooth shading normals and by the batched query's output), kColdNormal
// the pending normal of a sphere or of a hit below the first instance level, kColdTime the shutter time. They are 4 + 8
// + 2 of the 28 registers of per-ray cold state a lane carries through every phase; a flat-shaded triangle scene with
// one level of instances (cornellbox, the stress scene, PointInstancedMedCity) needs none of them, and without them the
// four-wave kernel's 32 spilled registers are 0 (round 3: bench extend 87.8 -> 79.6 ms per step, MedCity +3.9 %).
template <bool ANY, bool STATS, int ROWS, bool DIRECT, bool LEAN, int COLD, bool UMASK, class Fetch, class Emit>
__device__ void traverse_pool(const DevScene &S, uint32_t *lds /* this wave's pool_lds_dwords<ROWS>() */, float t_min,
                              uint32_t umask /* UMASK: the ray mask of every ray of the launch */,
                              const uint32_t *lds_nodes /* staged top of the tree */, uint32_t n_lds,
                              uint32_t n_lds_pk /* packets staged behind the n_lds nodes */, uint32_t pstack,
                              uint32_t &err, LaneStats &st, Fetch fetch, Emit emit) {
  const int lane = threadIdx.x & 63;
  constexpr int RL = ROWS * 64;
  // NOPK (kNoPackets in the kernel's cold argument): the image holds no Tri4 packet — the packet phase only reads the
  // Leaf record and goes on to its scalar list, no tree ever needs the Woop shear constants
  constexpr bool NOPK = (COLD & (int)kNoPackets) != 0;
  float4 *f0 = reinterpret_cast<float4 *>(lds);
  float4 *f1 = reinterpret_cast<float4 *>(lds + 4 * RL);
  float2 *f2 = reinterpret_cast<float2 *>(lds + 8 * RL);
  uint32_t *w = lds + 10 * RL;  // dword planes
  enum { W_CTL = 0, W_CUR, W_AUX, W_CURSOR, W_MASK };
  // B_BINST: the level-1 instance the closest hit so far was found in, when its normal is still in that instance's
  // space (CRT_DEFER_NORMAL) — the transform to world space (prim.rs:358-364) is then applied once, at emit,
  // instead of at every instance exit that improved the hit; kInvalid otherwise.
  enum { B_SLOT = 0, B_BU, B_BV, B_BDEFER, B_BGEOM, B_BINST };
  uint32_t *stk = lds + (UMASK ? 14 : 15) * RL;
  auto at = [&](int row) { return row * 64 + lane; };
  auto W = [&](int plane, int row) -> uint32_t & { return w[plane * RL + row * 64 + lane]; };
  auto STK = [&](int e, int row) -> uint32_t & { return stk[e * RL + row * 64 + lane]; };

  // rarely touched per-ray state, private memory indexed by row
  float side_d[ROWS][4];   // direction in the current frame, shutter time
  float side_n[ROWS][4];   // pending outward normal of a sphere / instanced hit, prim id (bits)
  uint32_t best[ROWS][6];  // caller's slot tag; u, v (bits), pending triangle, geometry id of the closest hit so far, B_BINST
  Frame frames[ROWS][kMaxLevels];
  uint32_t spill[ROWS][kPoolSpill];

  // side_d / side_n / best live in registers. Indexed by a run-time row they would be read and written through
  // compare-and-select chains over ALL their elements (what the compiler makes of a dynamic index into a promoted
  // array: 30 instructions per store); with the row loop unrolled each access is one select per row.
  // Fields this instantiation does not keep read as zero and are never written (COLD): the scene cannot reach the
  // code that would use them (no sphere, no smooth normal, no moving instance, no second instance level).
  auto cold_dead = [&](auto &arr, int f) {
    if ((void *)&arr == (void *)&side_n) return !(COLD & kColdNormal);
    if ((void *)&arr == (void *)&side_d) return f == 3 && !(COLD & kColdTime);
    if ((void *)&arr == (void *)&best) return (f == B_BU || f == B_BV) && !(COLD & kColdUV);
    return false;
  };
  auto rd = [&](auto &arr, int row, int f) {
    auto r = arr[0][f];
    if (cold_dead(arr, f)) return (decltype(r))0;
#pragma unroll
    for (int k = 1; k < ROWS; k++) r = row == k ? arr[k][f] : r;
    return r;
  };
  auto wr = [&](auto &arr, int row, int f, auto val) {
    if (cold_dead(arr, f)) return;
#pragma unroll
    for (int k = 0; k < ROWS; k++) arr[k][f] = row == k ? (decltype(arr[k][f] + 0))val : arr[k][f];
  };
  // The stack: entries [lo, sp) live in LDS — the TOP of the stack, a ring of pstack slots (entry e in slot e mod
  // pstack) — entries [0, lo) in private memory. A push onto a full ring moves the ring's OLDEST entry out; a pop that
  // finds the ring empty takes the entry back from private memory. Pushes and pops of a ray working deep in a large tree
  // stay in LDS; private memory is touched only on net growth past pstack and on the way back down. (Rounds 1-3 kept the
  // BOTTOM pstack entries in LDS: every push and pop above them was a scratch access — the ten-entry split was worth
  // +10 % to the stress scene over the six-entry one for that reason alone.) While nothing has moved out (lo == 0)
  // slots are positions: the common case costs nothing. CRT_STACK_RING=0: the old form (A/B).
  const uint32_t ring_inv = (65536u + pstack - 1u) / pstack;  // e / pstack == (e * ring_inv) >> 16 for e < 256, pstack <= 16
  auto slot_of = [&](uint32_t e, uint32_t lo) -> int {
    if (!CRT_STACK_RING || lo == 0u) return (int)e;
    return (int)(e - ((e * ring_inv) >> 16) * pstack);
  };
  auto push = [&](int row, uint32_t &sp, uint32_t &lo, uint32_t x) {
    if (CRT_STACK_RING) {
      if (sp - lo == pstack) {  // the ring is full: its oldest entry moves out
        if (lo >= (uint32_t)kPoolSpill) { err |= 1u; return; }
        spill[row][lo] = STK(slot_of(lo, lo), row);
        lo++;
      }
      STK(slot_of(sp, lo), row) = x;
      sp++;
      return;
    }
    if (sp < pstack) STK((int)sp, row) = x;
    else if (sp - pstack < (uint32_t)kPoolSpill) spill[row][sp - pstack] = x;
    else { err |= 1u; return; }
    sp++;
  };
  auto pop = [&](int row, uint32_t &sp, uint32_t &lo) -> uint32_t {
    sp--;
    if (CRT_STACK_RING) {
      if (sp >= lo) return STK(slot_of(sp, lo), row);
      lo--;  // the ring is empty: sp == lo - 1, the entry is the last one that moved out
      return spill[row][sp];
    }
    return sp < pstack ? STK((int)sp, row) : spill[row][sp - pstack];
  };
  // What the ray does next: the rest of the leaf's scalar list, else the next stack entry, else leave the tree.
  // A leaf word (kLeafTag set, not the empty marker): the leaf to run packets of, or — direct form, scenes with
  // DevScene::direct_leaves — the scalar list itself (crt_internal.h). Counts what the packet step counts on entry.
  constexpr bool direct_leaves = DIRECT;  // the caller instantiates the engine per DevScene::direct_leaves
  auto leaf_word = [&](uint32_t e, uint32_t level, uint32_t &cur, uint32_t &cursor, uint32_t &rem) -> uint32_t {
    if (direct_leaves && (e & kDirectLeafTag)) {
      // entries: indices[cursor ..), or — kDirectInstTag — the instance slots themselves, carried in the cursor in the
      // list's own form (kIndexInstance | slot) so the scalar step needs no index fetch
      cursor = (CRT_DIRECT_INST && (e & kDirectInstTag)) ? (kIndexInstance | (e & kDirectIndexMask)) : (e & kDirectIndexMask);
      rem = (e >> 28) & 3u;
      if (STATS) { st.leaves[level > 0 ? 1 : 0]++; st.prims[level > 0 ? 1 : 0] += rem; }
      return PH_SCALAR;
    }
    cur = e & ~kLeafTag;
    cursor = 0;  // packet counter while in PH_PACKET
    return PH_PACKET;
  };
  auto advance = [&](int row, uint32_t &sp, uint32_t &lo, uint32_t base, uint32_t level, uint32_t &rem, uint32_t &cur,
                     uint32_t &cursor) -> uint32_t {
    if (rem > 0) return PH_SCALAR;
    for (;;) {
      if (sp == base) return level > 0 ? PH_EXIT : PH_EMIT;
      const uint32_t e = pop(row, sp, lo);
      if (e & kLeafTag) {
        if (e == kInvalid) continue;
        return leaf_word(e, level, cur, cursor, rem);
      }
      cur = e;
      return PH_NODE;
    }
  };
  auto unpack_k = [&](uint32_t c, int &kx, int &ky, int &kz) {
    kz = (int)((c >> 20) & 3u);
    kx = kz == 2 ? 0 : kz + 1;
    ky = kx == 2 ? 0 : kx + 1;
    if ((c >> 22) & 1u) { const int t = kx; kx = ky; ky = t; }
  };
  // (re)derive the per-tree ray constants from origin + direction and store them (closest is passed through)
  auto store_ray = [&](int row, RayCtx &r, bool with_shear, float closest, uint32_t &kz_out, uint32_t &swap_out) {
    r.kx = 0; r.ky = 1; r.kz = 2; r.sx = r.sy = r.sz = 0.0f;
    setup_ray(r, with_shear);
    kz_out = (uint32_t)r.kz;
    const int kx0 = r.kz == 2 ? 0 : r.kz + 1;
    swap_out = (r.kx != kx0) ? 1u : 0u;
    f0[at(row)] = make_float4(r.ox, r.oy, r.oz, closest);
    f1[at(row)] = make_float4(r.ix, r.iy, r.iz, r.sx);
    f2[at(row)] = make_float2(r.sy, r.sz);
  };

  // a fetched ray takes slot `row` of this lane: per-tree constants, control words, the cold state; returns its phase
  auto setup_new = [&](int row, const RayIn &in) -> uint32_t {
    RayCtx r;
    r.ox = in.ox; r.oy = in.oy; r.oz = in.oz; r.dx = in.dx; r.dy = in.dy; r.dz = in.dz;
    uint32_t kz, swap;
    store_ray(row, r, !NOPK && S.has_packets != 0, in.t_max, kz, swap);
    const bool empty = S.root == kInvalid;  // bvh.rs:442-444
    W(W_CTL, row) = ctl_pack(0, 0, 0, S.has_packets ? 1u : 0u, kz, swap, 0);
    W(W_CUR, row) = S.root;
    W(W_AUX, row) = 0; W(W_CURSOR, row) = 0; wr(best, row, B_SLOT, in.slot);
    if (!UMASK) W(W_MASK, row) = in.mask;
    wr(best, row, B_BU, 0); wr(best, row, B_BV, 0); wr(best, row, B_BDEFER, kInvalid); wr(best, row, B_BGEOM, kInvalid);
    wr(best, row, B_BINST, kInvalid);
    wr(side_d, row, 0, in.dx); wr(side_d, row, 1, in.dy); wr(side_d, row, 2, in.dz); wr(side_d, row, 3, in.time);
    if (STATS) st.queries[0]++;
    return empty ? PH_EMIT : PH_NODE;
  };

  uint32_t ph[ROWS];  // phase of this lane's slots: only this lane ever changes them, so they live in registers
#pragma unroll
  for (int row = 0; row < ROWS; row++) ph[row] = PH_FREE;
  bool more = true;  // wave-uniform: the source may still hold rays
  unsigned long long t_mark = STATS ? (unsigned long long)clock64() : 0ull;
  auto lap = [&](int k) {  // STATS: charge the cycles since the last mark to phase k
    if (STATS) {
      const unsigned long long now = (unsigned long long)clock64();
      st.ph_cyc[k] += now - t_mark;
      t_mark = now;
    }
  };

  for (;;) {
    uint32_t have = 0;  // bit q: this lane has a slot in phase q
#pragma unroll
    for (int row = 0; row < ROWS; row++) have |= 1u << ph[row];
    const int n_free = wave_count(have & (1u << PH_FREE));
    if (more && n_free >= kFetchMin) {
      // ---- fetch + setup ----
      lap(0);
      const bool want = (have & (1u << PH_FREE)) != 0;
      RayIn in;
      const bool got = fetch(want, in);
      if (got) {
        CRT_PHASE(1)
        int row = 0;
#pragma unroll
        for (int k = ROWS - 1; k >= 0; k--)
          if (ph[k] == PH_FREE) row = k;
        const uint32_t first = setup_new(row, in);
#pragma unroll
        for (int k = 0; k < ROWS; k++)
          if (k == row) ph[k] = first;
      }
      if (__ballot(want && !got)) more = false;  // a lane asked and got nothing: the source is dry
      lap(1);
      continue;
    }

    // ---- pick the phase to run: a rare one once enough lanes wait for it, else the busier of node / packet ----
    const int n_node = wave_count(have & (1u << PH_NODE));
    const int n_pkt = wave_count(have & (1u << PH_PACKET));
    const int n_sc = wave_count(have & (1u << PH_SCALAR));
    const int n_exit = wave_count(have & (1u << PH_EXIT));
    const int n_emit = wave_count(have & (1u << PH_EMIT));
    uint32_t q;
    {
      // emit is the cheapest rare phase and frees a slot; it triggers at kRareMin + kEmitBias waiting lanes
      int best_rare = n_sc - kScalarBias;
      uint32_t q_rare = PH_SCALAR;
      if (n_exit - kExitBias > best_rare) { best_rare = n_exit - kExitBias; q_rare = PH_EXIT; }
      if (n_emit - kEmitBias > best_rare) { best_rare = n_emit - kEmitBias; q_rare = PH_EMIT; }
      if (best_rare >= kRareMin) q = q_rare;
      else if (n_node + n_pkt > 0) q = (n_node > 0 && (n_pkt == 0 || n_node + kPacketBias >= n_pkt)) ? PH_NODE : PH_PACKET;
      else {  // nothing but rare work is left: the largest group, whatever its size
        best_rare = n_sc; q_rare = PH_SCALAR;
        if (n_exit > best_rare) { best_rare = n_exit; q_rare = PH_EXIT; }
        if (n_emit > best_rare) { best_rare = n_emit; q_rare = PH_EMIT; }
        if (best_rare > 0) q = q_rare;
        else break;  // every slot is free (and the source is dry, or the fetch above would have run)
      }
    }
    bool mine = (have & (1u << q)) != 0;
    int row = 0;
#pragma unroll
    for (int k = ROWS - 1; k >= 0; k--)
      if (ph[k] == q) row = k;
    uint32_t next = q;  // phase of slot `row` after this step (meaningful for lanes with `mine`)
    if (mine) { CRT_PHASE(0) }
    lap(0);

    if (q == PH_NODE) {
      // ================= node expansion (bvh.rs:455-505), repeated while most of the lanes stay in it =================
      // Every lane loads (its own slot of `row`: always addressable); lanes without a ray in this phase just
      // carry the values along, `act` keeps them out of every effect. No zero-initialisation, no masked loads.
      float4 g0 = f0[at(row)], g1 = f1[at(row)];
      uint32_t c = W(W_CTL, row);
      uint32_t cur = W(W_CUR, row), cursor = 0, rem = 0;
      uint32_t sp = c & 0xffu, lo = c >> 24;
      uint32_t base = (c >> 8) & 0xffu, level = (c >> 16) & 7u;
      bool act = mine;
      // the ray of slot `row` leaves the loop: its control words go back to LDS
      auto retire = [&]() {
        W(W_CTL, row) = ctl_sp(c, sp, lo);
        W(W_CUR, row) = cur;
        if (next == PH_PACKET) W(W_CURSOR, row) = 0;
        if (next == PH_SCALAR) {  // a direct leaf: the scalar list starts at once
          W(W_CURSOR, row) = cursor;
          W(W_AUX, row) = (W(W_AUX, row) & 0xffu) | (rem << 8);
        }
      };
      for (;;) {
        if (act) {
          CRT_PHASE(2)
          const float closest = g0.w;
          if (STATS) st.nodes[level > 0 ? 1 : 0]++;
          float4 mnx, mny, mnz, mxx, mxy, mxz;
          uint4 ch;  // device form: leaf children tagged, empty lanes kInvalid (scene.cpp)
          if (cur < n_lds) {  // top of the tree: LDS (ds_read_b128), no trip through the vector memory pipeline
            const float4 *nb = reinterpret_cast<const float4 *>(lds_nodes + (size_t)cur * kLdsNodeStride);
            mnx = nb[0]; mny = nb[1]; mnz = nb[2]; mxx = nb[3]; mxy = nb[4]; mxz = nb[5];
            ch = *reinterpret_cast<const uint4 *>(nb + 6);
            // Opaque to the optimiser: without it the two arms are merged into FLAT loads through a generic pointer,
            // which serve the LDS lanes more slowly than ds_read_b128 (MedCity +1.8 %, profiles/README.md).
            asm volatile("" : "+v"(mnx.x), "+v"(mxx.x), "+v"(ch.x));
          } else {
            const WideNode *nd = &S.nodes[cur];
            const float4 *nb = reinterpret_cast<const float4 *>(nd);
            mnx = nb[0]; mny = nb[1]; mnz = nb[2]; mxx = nb[3]; mxy = nb[4]; mxz = nb[5];
            ch = *reinterpret_cast<const uint4 *>(nd->child);
          }
          const float lo_x[4] = {mnx.x, mnx.y, mnx.z, mnx.w}, lo_y[4] = {mny.x, mny.y, mny.z, mny.w},
                      lo_z[4] = {mnz.x, mnz.y, mnz.z, mnz.w};
          const float hi_x[4] = {mxx.x, mxx.y, mxx.z, mxx.w}, hi_y[4] = {mxy.x, mxy.y, mxy.z, mxy.w},
                      hi_z[4] = {mxz.x, mxz.y, mxz.z, mxz.w};
          const uint32_t child[4] = {ch.x, ch.y, ch.z, ch.w};
          float key[4];
          uint32_t ent[4];
#pragma unroll
          for (int l = 0; l < 4; l++) {  // RaySlab::slab4, bvh.rs:790-808
            const float t0x = (lo_x[l] - g0.x) * g1.x, t1x = (hi_x[l] - g0.x) * g1.x;
            const float t0y = (lo_y[l] - g0.y) * g1.y, t1y = (hi_y[l] - g0.y) * g1.y;
            const float t0z = (lo_z[l] - g0.z) * g1.z, t1z = (hi_z[l] - g0.z) * g1.z;
            const float tn = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), t_min);
            const float tf = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)), closest);
            key[l] = tn;
            ent[l] = (tn <= tf) ? child[l] : kInvalid;  // kInvalid = lane off (an empty lane's word already is)
          }
          // Stack image after this node, bottom to top. Ordered traversal: inner lanes far to near, then leaf
          // lanes far to near on top, so they pop first and near-first — the reference's "leaf lanes now, near
          // first; inner lanes pushed far to near" (bvh.rs:488-505) after its stable insertion sort of the hit
          // lanes by entry distance (bvh.rs:472-486: equal keys keep lane order). Any-hit: leaf lanes first, in lane
          // order, then the inner lanes from the highest down (bvh.rs:596-606). Every entry's final position follows from the six pairwise "is nearer"
          // relations, so the entries are stored with independent predicated writes — no sorting network — and
          // the top one, the entry the ray visits next, stays in a register.
          uint32_t on[4], lf4[4], pos[4];
#pragma unroll
          for (int i = 0; i < 4; i++) { on[i] = ent[i] != kInvalid ? 1u : 0u; lf4[i] = (on[i] && (ent[i] & kLeafTag)) ? 1u : 0u; }
          uint32_t n_tot;
          if (ANY) {
            // hit_any (bvh.rs:596-606) tests a node's leaf lanes AT ONCE, in lane order, and pushes its inner lanes in
            // lane order (the last one pops first): inner lanes at the bottom in lane order, leaf lanes above them in
            // REVERSE lane order, so lane 0's leaf is the next entry, then lane 1's ..., then the highest inner lane.
            // (Rounds 1-3 stacked all hit lanes in lane order — the highest lane first, leaf or not: same answers, but
            // 32 % more node visits on the stress scene's shadow rays than the reference makes; the any-hit counters
            // now equal the oracle's, tests/test_gpu_stress.py.)
            const uint32_t in0 = on[0] & ~lf4[0], in1 = on[1] & ~lf4[1], in2 = on[2] & ~lf4[2], in3 = on[3] & ~lf4[3];
            const uint32_t n_in = in0 + in1 + in2 + in3;
            pos[0] = lf4[0] ? n_in + lf4[1] + lf4[2] + lf4[3] : 0u;
            pos[1] = lf4[1] ? n_in + lf4[2] + lf4[3] : in0;
            pos[2] = lf4[2] ? n_in + lf4[3] : in0 + in1;
            pos[3] = lf4[3] ? n_in : in0 + in1 + in2;
            n_tot = on[0] + on[1] + on[2] + on[3];
          } else {
            // For every pair i < j: x = 1 iff lane j ends up below lane i. Same kind: the farther one is below,
            // and lane i is the nearer one iff key_i <= key_j (stable: equal keys keep lane order, bvh.rs:472-486).
            // Different kinds: the inner lane is below the leaf lane. Exactly one of the two is below the other, so
            // pos_i += on_j & x and pos_j += on_i & !x; no branches, no per-kind sums.
#pragma unroll
            for (int i = 0; i < 4; i++) pos[i] = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
              for (int j = i + 1; j < 4; j++) {
                const uint32_t a = key[i] <= key[j] ? 1u : 0u;
                const uint32_t t = lf4[i] ^ lf4[j];
                const uint32_t x = (t & lf4[i]) | (~t & a);  // v_bfi_b32
                pos[i] += on[j] & x;
                pos[j] += on[i] & (x ^ 1u);
              }
            }
            n_tot = on[0] + on[1] + on[2] + on[3];
          }
          if (n_tot == 0) {
            next = advance(row, sp, lo, base, level, rem, cur, cursor);
          } else {
            uint32_t top_e = 0;
            if (lo == 0u && sp + n_tot - 1 <= pstack) {  // the stored entries fit the LDS part of the stack, nothing has moved out
#pragma unroll
              for (int i = 0; i < 4; i++) {
                if (on[i]) {
                  if (pos[i] == n_tot - 1) top_e = ent[i];
                  else STK((int)(sp + pos[i]), row) = ent[i];
                }
              }
              sp += n_tot - 1;
            } else if (!CRT_STACK_RING && CRT_SPILL_UNROLLED) {
              // deep stack: the same independent predicated writes, each to LDS or to the private part by its position
              // (rounds 1-3 ran a loop over the entries here — a select chain and a branchy push per entry, for every node
              // step of a ray working below the LDS part of its stack)
#pragma unroll
              for (int i = 0; i < 4; i++) {
                if (on[i]) {
                  if (pos[i] == n_tot - 1) top_e = ent[i];
                  else {
                    const uint32_t at = sp + pos[i];
                    if (at < pstack) STK((int)at, row) = ent[i];
                    else if (at - pstack < (uint32_t)kPoolSpill) spill[row][at - pstack] = ent[i];
                    else err |= 1u;
                  }
                }
              }
              sp += n_tot - 1;
            } else {  // one entry at a time through push()
              for (uint32_t k = 0; k < n_tot; k++) {
                uint32_t e = 0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                  if (on[i] && pos[i] == k) e = ent[i];
                if (k == n_tot - 1) top_e = e;
                else push(row, sp, lo, e);
              }
            }
            if (top_e & kLeafTag) {
              if (top_e == kInvalid) next = advance(row, sp, lo, base, level, rem, cur, cursor);
              else next = leaf_word(top_e, level, cur, cursor, rem);
            } else {
              cur = top_e;
              next = PH_NODE;
            }
          }
          act = next == PH_NODE;
        }
        // SWAP: a lane whose ray has just left the node phase hands the lane over to its OTHER ray if that one waits
        // for a node step — inside the loop, for the price of four LDS stores and four loads, instead of idling until
        // the loop ends and a whole scheduling round re-picks the rows. The loop stays fuller and runs longer; a ray's
        // own sequence of steps is untouched (results and counters unchanged).
        if (CRT_NODE_SWAP && ROWS == 2) {
          if (mine && !act) {
            retire();
#pragma unroll
            for (int k = 0; k < ROWS; k++)
              if (k == row) ph[k] = next;
            const uint32_t ph_other = row == 0 ? ph[ROWS - 1] : ph[0];
            if (ph_other == PH_NODE) {
              row ^= 1;
              g0 = f0[at(row)]; g1 = f1[at(row)];
              c = W(W_CTL, row);
              cur = W(W_CUR, row); cursor = 0; rem = 0;
              sp = c & 0xffu; lo = c >> 24; base = (c >> 8) & 0xffu; level = (c >> 16) & 7u;
              next = PH_NODE;
              act = true;
            } else {
              mine = false;  // nothing of this lane is in flight in this loop any more
            }
          }
        }
        if (wave_count(act) < kStickyMin) break;  // wave-uniform
      }
      if (mine) retire();
    } else if (q == PH_PACKET) {
      // ============ Tri4 packets of the current leaf (bvh.rs:514-562, triangle.rs:276-348), one per turn ============
      const float4 g0 = f0[at(row)], g1 = f1[at(row)];  // unconditional, as in the node phase
      const float2 g2 = f2[at(row)];
      const uint32_t c = W(W_CTL, row), rmask = UMASK ? umask : W(W_MASK, row);
      uint32_t cur = W(W_CUR, row), k = W(W_CURSOR, row), aux = W(W_AUX, row);
      uint32_t sp = c & 0xffu, lo = c >> 24;
      const uint32_t base = (c >> 8) & 0xffu, level = (c >> 16) & 7u;
      int kx, ky, kz;
      unpack_k(c, kx, ky, kz);
      float closest = g0.w;
      float bu = 0.0f, bv = 0.0f;
      uint32_t bdefer = kInvalid;
      bool accepted = false, occluded = false;
      uint32_t cursor = 0, rem = 0;
      bool act = mine;
      for (;;) {
        if (act) {
          // The leaf record whole, in one 16-byte load: read field by field the compiler fetches pkt_first in a second,
          // dependent round trip (it is used only when the leaf has a packet left).
          Leaf lf;
          {
            uint4 lw = *reinterpret_cast<const uint4 *>(&S.leaves[cur]);
            asm volatile("" : "+v"(lw.x), "+v"(lw.y), "+v"(lw.z), "+v"(lw.w));
            lf.pkt_first = lw.x; lf.pkt_count = lw.y; lf.idx_first = lw.z; lf.idx_count = lw.w;
          }
          const int sl = level > 0 ? 1 : 0;
          if (STATS && k == 0) { st.leaves[sl]++; st.packets[sl] += lf.pkt_count; st.prims[sl] += lf.idx_count; }
          if (!NOPK && k < lf.pkt_count) {
            const uint32_t pki = lf.pkt_first + k;
            // A packet of the staged window is read from LDS (ds_read_b128), the rest from L2; two arms, as for the
            // nodes: merged they become FLAT loads through a generic pointer.
            const bool pk_lds = pki < n_lds_pk;
            const Tri4 *pk = &S.packets[pki];
            const Tri4 *pk_l = reinterpret_cast<const Tri4 *>(lds_nodes + (size_t)n_lds * kLdsNodeStride) + (pk_lds ? pki : 0u);
            uint4 prim4;
            uint32_t normal_ok, m;
            float4 A_x, A_y, A_z, B_x, B_y, B_z, C_x, C_y, C_z;
            auto load_vertices = [&](const Tri4 *q) {
              const float4 *pl = reinterpret_cast<const float4 *>(q);
              A_x = pl[0 + kx]; A_y = pl[0 + ky]; A_z = pl[0 + kz];
              B_x = pl[3 + kx]; B_y = pl[3 + ky]; B_z = pl[3 + kz];
              C_x = pl[6 + kx]; C_y = pl[6 + ky]; C_z = pl[6 + kz];
            };
            if (!LEAN) {
              // The packet whole — nine vertex planes in the ray's Woop order, primitive ids, masks, normal-ok bits: twelve
              // 16-byte loads requested together and waited for once. (Fetched as needed — masks, then vertices behind
              // the mask test, then ids — a packet step is three dependent round trips; the mask test rejects next to
              // nothing.) Costs 12 more live registers than the staged form below.
              uint4 meta, mtail;  // active, mask_and, mask_or, masks[0] | masks[1..3], normal_ok
              auto load_packet = [&](const Tri4 *q) {
                load_vertices(q);
                const uint4 *pw = reinterpret_cast<const uint4 *>(q);
                prim4 = pw[9]; meta = pw[10]; mtail = pw[11];
              };
#define CRT_PACKET_ARRIVED                                                                                          \
  asm volatile("" : "+v"(A_x.x), "+v"(A_y.x), "+v"(A_z.x), "+v"(B_x.x), "+v"(B_y.x), "+v"(B_z.x), "+v"(C_x.x), \
               "+v"(C_y.x), "+v"(C_z.x), "+v"(prim4.x), "+v"(meta.x), "+v"(mtail.x))
              if (pk_lds) { load_packet(pk_l); CRT_PACKET_ARRIVED; }
              else { load_packet(pk); CRT_PACKET_ARRIVED; }
#undef CRT_PACKET_ARRIVED
              normal_ok = mtail.w;
              if (rmask & meta.y) m = meta.x;                                    // triangle.rs:257-271
              else if ((rmask & meta.z) == 0) m = 0;
              else {
                const uint32_t lane_mask[4] = {meta.w, mtail.x, mtail.y, mtail.z};
                m = 0;
#pragma unroll
                for (int l = 0; l < 4; l++)
                  if ((meta.x & (1u << l)) && (lane_mask[l] & rmask)) m |= 1u << l;
              }
            } else {
              // Staged form (the 128-register kernels): masks first, vertices and ids behind the mask test.
              uint4 meta;  // active, mask_and, mask_or, masks[0]
              if (pk_lds) { meta = *reinterpret_cast<const uint4 *>(&pk_l->active); asm volatile("" : "+v"(meta.x)); }
              else meta = *reinterpret_cast<const uint4 *>(&pk->active);
              if (rmask & meta.y) m = meta.x;                                    // triangle.rs:257-271
              else if ((rmask & meta.z) == 0) m = 0;
              else {
                m = 0;
#pragma unroll
                for (int l = 0; l < 4; l++)
                  if ((meta.x & (1u << l)) && ((pk_lds ? pk_l->masks[l] : pk->masks[l]) & rmask)) m |= 1u << l;
              }
              if (m != 0) {
                // primitive ids and the normal-ok bits ride along with the vertex loads: fetched at the point of use
                // they would be up to four more dependent round trips per packet, one per accepted lane
                auto load_rest = [&](const Tri4 *q) {
                  prim4 = *reinterpret_cast<const uint4 *>(q->prim);
                  normal_ok = q->normal_ok;
                  load_vertices(q);
                };
                if (pk_lds) { load_rest(pk_l); asm volatile("" : "+v"(A_x.x), "+v"(B_x.x), "+v"(C_x.x), "+v"(prim4.x)); }
                else load_rest(pk);
              }
            }
            if (m != 0) {
              CRT_PHASE(3)
              RayCtx r;
              r.ox = g0.x; r.oy = g0.y; r.oz = g0.z;
              r.kx = kx; r.ky = ky; r.kz = kz; r.sx = g1.w; r.sy = g2.x; r.sz = g2.y;
              r.okx = sel3(g0.x, g0.y, g0.z, kx); r.oky = sel3(g0.x, g0.y, g0.z, ky); r.okz = sel3(g0.x, g0.y, g0.z, kz);
              const uint32_t prim_of[4] = {prim4.x, prim4.y, prim4.z, prim4.w};
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
                if (hits) occluded = true;
              } else {
#pragma unroll
                for (int l = 0; l < 4; l++) {  // bvh.rs:533-550
                  if (!((hits >> l) & 1u)) continue;
                  if (ht[l] > closest) continue;               // strict: an exact tie goes to the later lane
                  if (!((normal_ok >> l) & 1u)) continue;  // prim.rs:81-83 degenerate sliver
                  closest = ht[l]; bu = hu[l]; bv = hv[l];
                  bdefer = prim_of[l];
                  accepted = true;
                  if (STATS) st.accepted++;
                }
              }
              if (fallback && !occluded) {  // bvh.rs:551-561 / :636-643 — lanes sitting exactly on an edge
#pragma unroll
                for (int l = 0; l < 4; l++) {
                  if (!((fallback >> l) & 1u)) continue;
                  const uint32_t pi = prim_of[l];
                  const DevPrim *p = &S.prims[pi];
                  if ((rmask & p->mask) == 0) continue;
                  CRT_PHASE(7)
                  float t, u, v;
                  if (!tri_scalar(r, p->d, t_min, closest, t, u, v)) continue;
                  if (ANY) { occluded = true; break; }
                  if (!((normal_ok >> l) & 1u)) continue;
                  closest = t; bu = u; bv = v; bdefer = pi;
                  accepted = true;
                  if (STATS) st.accepted++;
                }
              }
            }
            k++;
          }
          if (occluded) {
            next = PH_EMIT;
            act = false;
          } else if (k < lf.pkt_count) {
            next = PH_PACKET;  // act stays true: the leaf has another packet
          } else {
            cursor = lf.idx_first;
            rem = lf.idx_count;
            next = PH_SCALAR;  // resolved below, once, after the run
            act = false;
          }
        }
        if (wave_count(act) < kStickyMin) break;  // wave-uniform
      }
      if (mine) {
        if (accepted) {
          f0[at(row)].w = closest;
          wr(best, row, B_BU, __float_as_uint(bu)); wr(best, row, B_BV, __float_as_uint(bv)); wr(best, row, B_BDEFER, bdefer);
          wr(best, row, B_BINST, kInvalid);
          aux |= 1u << level;
        }
        if (occluded) aux |= 1u;
        if (next == PH_PACKET) {
          cursor = k;
        } else if (next == PH_SCALAR) {  // the leaf's packets are done: scalar list, or on to the next entry
          next = advance(row, sp, lo, base, level, rem, cur, cursor);
          aux = (aux & 0xffu) | (rem << 8);
        }
        W(W_CTL, row) = ctl_sp(c, sp, lo);
        W(W_CUR, row) = cur;
        W(W_CURSOR, row) = cursor;
        W(W_AUX, row) = aux;
      }
    } else if (q == PH_SCALAR) {
      // ================= one primitive of the leaf's scalar list (bvh.rs:564-570 / :646-651) =================
      if (mine) {
        CRT_PHASE(4)
        const float4 g0 = f0[at(row)];
        uint32_t c = W(W_CTL, row);
        uint32_t sp = c & 0xffu, lo = c >> 24;
        uint32_t base = (c >> 8) & 0xffu, level = (c >> 16) & 7u;
        const uint32_t hp = (c >> 19) & 1u;
        uint32_t cur = W(W_CUR, row);
        uint32_t cursor = W(W_CURSOR, row);
        uint32_t aux = W(W_AUX, row);
        const uint32_t rmask = UMASK ? umask : W(W_MASK, row);
        uint32_t rem = aux >> 8;  // 24 bits: a leaf's scalar list is not limited to 255 entries (make_leaf fallbacks)
        float closest = g0.w;
        const float dx = rd(side_d, row, 0), dy = rd(side_d, row, 1), dz = rd(side_d, row, 2), time = rd(side_d, row, 3);
        bool occluded = false;
        const uint32_t pi = (CRT_DIRECT_INST && direct_leaves && (cursor & kIndexInstance)) ? cursor : S.indices[cursor];
        cursor++;
        rem--;
        if (pi & kIndexInstance) {  // prim.rs:345-378 — the list entry IS the instance record's slot
          const uint32_t inst = pi & ~kIndexInstance;
          const DevInstance *in = &S.instances[inst];
          const uint4 ih = *reinterpret_cast<const uint4 *>(&in->root);  // root, flags, geom_id, mask
          if ((rmask & ih.w) != 0) {  // prim.rs:52-54
            if (level + 1 >= (uint32_t)kMaxLevels) err |= 2u;
            else {
              float w2l[12];
              if ((ih.y & 2u) && time > 0.0f) motion_w2l(*reinterpret_cast<const DevInstanceMotion *>(S.normals + (ih.y >> 2)), time, w2l);
              else {
#pragma unroll
                for (int i = 0; i < 12; i++) w2l[i] = in->w2l[i];
              }
              // transform_point3a / transform_vector3a: ((x_axis*v.x + y_axis*v.y) + z_axis*v.z) [+ translation]
              float px = w2l[0] * g0.x, py = w2l[1] * g0.x, pz = w2l[2] * g0.x;
              px = px + w2l[3] * g0.y; py = py + w2l[4] * g0.y; pz = pz + w2l[5] * g0.y;
              px = px + w2l[6] * g0.z; py = py + w2l[7] * g0.z; pz = pz + w2l[8] * g0.z;
              px = px + w2l[9]; py = py + w2l[10]; pz = pz + w2l[11];
              float qx = w2l[0] * dx, qy = w2l[1] * dx, qz = w2l[2] * dx;
              qx = qx + w2l[3] * dy; qy = qy + w2l[4] * dy; qz = qz + w2l[5] * dy;
              qx = qx + w2l[6] * dz; qy = qy + w2l[7] * dz; qz = qz + w2l[8] * dz;
              RayCtx r;
              r.ox = px; r.oy = py; r.oz = pz; r.dx = qx; r.dy = qy; r.dz = qz;  // unnormalised: local t == world t
              // The instanced tree's root node is tested HERE, with the local ray still in registers: the placement's
              // world box is the box of a transformed box (prim.rs:298-319), so many rays that enter it touch none of
              // the root's children. Such a visit ends where the reference's ends — Bvh::hit expands the root, finds
              // no lane, returns None (bvh.rs:455-470) — but without saving a frame, rewriting the ray's LDS state,
              // a node step and an exit step (each a scheduling round at a third of the lanes) for nothing.
              bool enter = true;
              if (CRT_ROOT_REJECT && DIRECT) {  // the instance-heavy engine only: on cornellbox (two placements) the test costs 1.5 % and rejects little
                float4 mnx, mny, mnz, mxx, mxy, mxz;
                uint4 ch;
                if (CRT_LDS_ROOT && ih.x < n_lds) {  // the instanced trees' roots are numbered into the LDS window (scene.cpp)
                  const float4 *nb = reinterpret_cast<const float4 *>(lds_nodes + (size_t)ih.x * kLdsNodeStride);
                  mnx = nb[0]; mny = nb[1]; mnz = nb[2]; mxx = nb[3]; mxy = nb[4]; mxz = nb[5];
                  ch = *reinterpret_cast<const uint4 *>(nb + 6);
                  asm volatile("" : "+v"(mnx.x), "+v"(mxx.x), "+v"(ch.x));  // two arms, as in the node phase
                } else {
                  const float4 *nb = reinterpret_cast<const float4 *>(&S.nodes[ih.x]);
                  mnx = nb[0]; mny = nb[1]; mnz = nb[2]; mxx = nb[3]; mxy = nb[4]; mxz = nb[5];
                  ch = *reinterpret_cast<const uint4 *>(nb + 6);
                }
                const float TINY = 1e-20f, HUGE_ = 1e20f;  // safe_inv3 (bvh.rs:662-668), as setup_ray
                const float ix = absf(qx) < TINY ? copysgn(HUGE_, qx) : 1.0f / qx;
                const float iy = absf(qy) < TINY ? copysgn(HUGE_, qy) : 1.0f / qy;
                const float iz = absf(qz) < TINY ? copysgn(HUGE_, qz) : 1.0f / qz;
                const float lo_x[4] = {mnx.x, mnx.y, mnx.z, mnx.w}, lo_y[4] = {mny.x, mny.y, mny.z, mny.w},
                            lo_z[4] = {mnz.x, mnz.y, mnz.z, mnz.w};
                const float hi_x[4] = {mxx.x, mxx.y, mxx.z, mxx.w}, hi_y[4] = {mxy.x, mxy.y, mxy.z, mxy.w},
                            hi_z[4] = {mxz.x, mxz.y, mxz.z, mxz.w};
                const uint32_t child[4] = {ch.x, ch.y, ch.z, ch.w};
                enter = false;
#pragma unroll
                for (int l = 0; l < 4; l++) {  // RaySlab::slab4, bvh.rs:790-808
                  const float t0x = (lo_x[l] - px) * ix, t1x = (hi_x[l] - px) * ix;
                  const float t0y = (lo_y[l] - py) * iy, t1y = (hi_y[l] - py) * iy;
                  const float t0z = (lo_z[l] - pz) * iz, t1z = (hi_z[l] - pz) * iz;
                  const float tn = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), t_min);
                  const float tf = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)), closest);
                  enter |= (tn <= tf) && child[l] != kInvalid;
                }
              }
              if (STATS) { st.descents++; st.queries[1]++; }
              if (!enter) {
                if (STATS) st.nodes[1]++;  // the root visit the reference makes
              } else {
                Frame f;
                f.ox = g0.x; f.oy = g0.y; f.oz = g0.z; f.dx = dx; f.dy = dy; f.dz = dz;
                f.cursor = cursor; f.cend = rem; f.base = base; f.inst = inst; f.geom = ih.z; f.has_packets = hp;
                frames[row][level] = f;
                wr(side_d, row, 0, qx); wr(side_d, row, 1, qy); wr(side_d, row, 2, qz);
                level++;
                aux &= ~(1u << level);
                base = sp;
                cursor = 0;
                rem = 0;
                uint32_t kz, swap;
                store_ray(row, r, !NOPK && (ih.y & 1u) != 0, closest, kz, swap);
                c = ctl_pack(sp, base, level, ih.y & 1u, kz, swap, 0);
                push(row, sp, lo, ih.x);
              }
            }
          }
        } else {
        const DevPrim *p = &S.prims[pi];
        const uint4 hd = *reinterpret_cast<const uint4 *>(p);  // kind, geom_id, prim_id, mask
        if ((rmask & hd.w) != 0) {  // prim.rs:52-54
          if (hd.x == PRIM_SPHERE) {  // prim.rs:133-161
            const float4 s = *reinterpret_cast<const float4 *>(p->d);
            const float ocx = g0.x - s.x, ocy = g0.y - s.y, ocz = g0.z - s.z;
            const float a = dot3(dx, dy, dz, dx, dy, dz);
            const float half_b = dot3(ocx, ocy, ocz, dx, dy, dz);
            const float cc = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - s.w * s.w;
            const float disc = half_b * half_b - a * cc;
            if (!(disc < 0.0f)) {
              const float sqrt_d = sqrtf(disc);
              float root = (-half_b - sqrt_d) / a;
              bool ok = true;
              if (root <= t_min || root >= closest) {
                root = (-half_b + sqrt_d) / a;
                if (root <= t_min || root >= closest) ok = false;
              }
              if (ok) {
                if (ANY) occluded = true;
                else {
                  closest = root;
                  f0[at(row)].w = root;
                  wr(side_n, row, 0, ((g0.x + root * dx) - s.x) / s.w);
                  wr(side_n, row, 1, ((g0.y + root * dy) - s.y) / s.w);
                  wr(side_n, row, 2, ((g0.z + root * dz) - s.z) / s.w);
                  wr(side_n, row, 3, __uint_as_float(0u));  // prim_id 0
                  wr(best, row, B_BU, 0); wr(best, row, B_BV, 0); wr(best, row, B_BDEFER, kInvalid); wr(best, row, B_BGEOM, hd.y);
                  wr(best, row, B_BINST, kInvalid);
                  aux |= 1u << level;
                  if (STATS) st.accepted++;
                }
              }
            }
          } else {
            // A triangle on the scalar list (the builder always packs triangles; kept for completeness).
            RayCtx rr;
            rr.ox = g0.x; rr.oy = g0.y; rr.oz = g0.z; rr.dx = dx; rr.dy = dy; rr.dz = dz;
            setup_ray(rr, true);
            float t, u, v;
            if (tri_scalar(rr, p->d, t_min, closest, t, u, v)) {
              if (ANY) occluded = true;
              else {
                const float e1x = p->d[3] - p->d[0], e1y = p->d[4] - p->d[1], e1z = p->d[5] - p->d[2];
                const float e2x = p->d[6] - p->d[0], e2y = p->d[7] - p->d[1], e2z = p->d[8] - p->d[2];
                const bool flat = (e1y * e2z - e2y * e1z) == 0.0f && (e1z * e2x - e2z * e1x) == 0.0f &&
                                  (e1x * e2y - e2x * e1y) == 0.0f;
                if (!(flat && __float_as_uint(p->d[9]) == kInvalid)) {
                  closest = t;
                  f0[at(row)].w = t;
                  wr(best, row, B_BU, __float_as_uint(u)); wr(best, row, B_BV, __float_as_uint(v)); wr(best, row, B_BDEFER, pi);
                  wr(best, row, B_BINST, kInvalid);
                  aux |= 1u << level;
                  if (STATS) st.accepted++;
                }
              }
            }
          }
        }
        }
        if (occluded) { aux |= 1u; next = PH_EMIT; }
        else next = advance(row, sp, lo, base, level, rem, cur, cursor);
        aux = (aux & 0xffu) | (rem << 8);
        W(W_CTL, row) = (c & 0x00ff0000u) | sp | (base << 8) | (lo << 24);
        W(W_CUR, row) = cur;
        W(W_CURSOR, row) = cursor;
        W(W_AUX, row) = aux;
      }
    } else if (q == PH_EXIT) {
      // ============ the instanced tree is exhausted: back to the parent frame (prim.rs:358-364) ============
      if (mine) {
        CRT_PHASE(5)
        const uint32_t c = W(W_CTL, row);
        uint32_t sp = c & 0xffu, lo = c >> 24;
        uint32_t level = (c >> 16) & 7u;
        uint32_t aux = W(W_AUX, row);
        uint32_t cur = W(W_CUR, row);
        const float closest = f0[at(row)].w;
        const float time = rd(side_d, row, 3);
        const bool inner_found = (aux >> level) & 1u;
        level--;
        const Frame f = frames[row][level];
        // The instance's normal transform, shared by the exit (nested levels) and emit (deferred) sites: the hit's
        // normal and primitive id in the instanced scene's space -> normalize(w2l.matrix3^T * n) (prim.rs:327, :360).
        if (inner_found && CRT_DEFER_NORMAL && level == 0) {
          // back at the top level: keep the normal where it is (pending triangle or side_n) and remember the instance
          wr(best, row, B_BINST, f.inst); wr(best, row, B_BGEOM, f.geom);
          aux |= 1u;
          if (STATS) st.accepted++;
        } else if (inner_found) {
          float bnx, bny, bnz;
          uint32_t bprim;
          const uint32_t bdefer = rd(best, row, B_BDEFER);
          if (bdefer != kInvalid) {
            tri_normal(S, bdefer, __uint_as_float(rd(best, row, B_BU)), __uint_as_float(rd(best, row, B_BV)), bnx, bny, bnz);
            bprim = S.prims[bdefer].prim_id;
          } else {
            bnx = rd(side_n, row, 0); bny = rd(side_n, row, 1); bnz = rd(side_n, row, 2);
            bprim = __float_as_uint(rd(side_n, row, 3));
          }
          instance_normal(S, f.inst, time, bnx, bny, bnz);
          wr(side_n, row, 0, bnx); wr(side_n, row, 1, bny); wr(side_n, row, 2, bnz);
          wr(side_n, row, 3, __uint_as_float(bprim));
          // the hit is attributed to the instance's geometry id; prim_id stays the inner one
          wr(best, row, B_BDEFER, kInvalid); wr(best, row, B_BGEOM, f.geom);
          wr(best, row, B_BINST, kInvalid);
          aux |= 1u << level;
          if (STATS) st.accepted++;
        }
        RayCtx r;
        r.ox = f.ox; r.oy = f.oy; r.oz = f.oz; r.dx = f.dx; r.dy = f.dy; r.dz = f.dz;
        wr(side_d, row, 0, f.dx); wr(side_d, row, 1, f.dy); wr(side_d, row, 2, f.dz);
        uint32_t kz, swap;
        store_ray(row, r, !NOPK && f.has_packets != 0, closest, kz, swap);
        uint32_t cursor = f.cursor;
        uint32_t rem = f.cend;
        next = advance(row, sp, lo, f.base, level, rem, cur, cursor);
        W(W_CTL, row) = ctl_pack(sp, f.base, level, f.has_packets ? 1u : 0u, kz, swap, lo);
        W(W_CUR, row) = cur;
        W(W_CURSOR, row) = cursor;
        W(W_AUX, row) = (aux & 0xffu) | (rem << 8);
      }
    } else {
      // ================= emit: the ray is finished =================
      if (mine) {
        CRT_PHASE(6)
        const uint32_t aux = W(W_AUX, row);
        const uint32_t slot = rd(best, row, B_SLOT);
        const bool is_hit = (aux & 1u) != 0;
        Hit hit;
        hit.t = 0.0f; hit.u = 0.0f; hit.v = 0.0f; hit.nx = hit.ny = hit.nz = 0.0f; hit.geom = kInvalid; hit.prim = kInvalid;
        if (!ANY && is_hit) {
          hit.t = f0[at(row)].w;
          hit.u = __uint_as_float(rd(best, row, B_BU)); hit.v = __uint_as_float(rd(best, row, B_BV));
          const uint32_t bdefer = rd(best, row, B_BDEFER);
          if (bdefer != kInvalid) {
            tri_normal(S, bdefer, hit.u, hit.v, hit.nx, hit.ny, hit.nz);
            const DevPrim *p = &S.prims[bdefer];
            hit.geom = p->geom_id; hit.prim = p->prim_id;
          } else {
            hit.nx = rd(side_n, row, 0); hit.ny = rd(side_n, row, 1); hit.nz = rd(side_n, row, 2);
            hit.geom = rd(best, row, B_BGEOM); hit.prim = __float_as_uint(rd(side_n, row, 3));
          }
          const uint32_t binst = rd(best, row, B_BINST);
          if (CRT_DEFER_NORMAL && binst != kInvalid) {  // the hit lies in a level-1 instance: its normal is still local
            instance_normal(S, binst, rd(side_d, row, 3), hit.nx, hit.ny, hit.nz);
            hit.geom = rd(best, row, B_BGEOM);  // the instance's geometry id; prim_id stays the inner one
          }
        }
        // the ray's own direction: back in side_d once every instance frame has been left (closest-hit rays only
        // finish at level 0), so the caller need not load it again to orient the normal
        emit(slot, is_hit, hit, rd(side_d, row, 0), rd(side_d, row, 1), rd(side_d, row, 2));
        next = PH_FREE;
      }
      // REFILL (CRT_EMIT_REFILL=1, off): the lanes that just reported a ray take their next one here, into the slot
      // that fell free, instead of waiting for a fetch round of their own — one scheduling round less per ray. Measured
      // (round 3): the four-wave kernel goes from 28 to 39 spilled registers and the bench loses 10 % (extend 88 -> 103
      // ms per step); the three-wave kernels (MedCity, openpbr_showcase) do not move.
      if (CRT_EMIT_REFILL && more) {  // wave-uniform
        RayIn in;
        const bool got = fetch(mine, in);
        if (got) next = setup_new(row, in);
        if (__ballot(mine && !got)) more = false;
      }
    }
    if (mine) {
#pragma unroll
      for (int k = 0; k < ROWS; k++)
        if (k == row) ph[k] = next;
    }
    lap(q == PH_NODE ? 2 : q == PH_PACKET ? 3 : q == PH_SCALAR ? 4 : q == PH_EXIT ? 5 : 6);
  }
}

// ---- the engine as the kernels use it ----
#ifndef CRT_POOL_ROWS
#define CRT_POOL_ROWS 2
#endif
constexpr int kPoolNodes = CRT_POOL_NODES;          // nodes staged in LDS per workgroup, flat scenes (crt_internal.h)
constexpr int kPoolNodesDeep = CRT_POOL_NODES_DEEP;  // ... with the deep stack
constexpr int kEngineLdsFlat = (kBlock / 64) * pool_lds_dwords<CRT_POOL_ROWS>(kPoolStack) + kPoolNodes * kLdsNodeStride;
constexpr int kEngineLdsDeep = (kBlock / 64) * pool_lds_dwords<CRT_POOL_ROWS>(kPoolStackDeep) + kPoolNodesDeep * kLdsNodeStride;
constexpr int kEngineLdsDwords = kEngineLdsFlat > kEngineLdsDeep ? kEngineLdsFlat : kEngineLdsDeep;
// WIDE: the split for kernels that run FOUR workgroups per CU (four waves per SIMD: 128 registers, 40 KB of LDS each)
// instead of three — three stack entries per ray in LDS, a 26-node window, the staged packet loads (4 + 8: cornellbox
// equal, veach_mis -1.3 %, sun_sky -1.9 %; 2 + 44: cornellbox -2 %). On small flat scenes the fourth wave per SIMD buys
// more than the shorter LDS stack and the 30 spilled registers cost (cornellbox +4-6 %, veach_mis +9 %, sun_sky +8 % for
// the per-stage pipeline on it over the fused kernel on three; an instanced city, whose rays stack the parent's entries
// under the instance's, loses 6 %: profiles/README.md). The renderer and the batched queries pick it per scene
// (crt_internal.h, wide_split).
constexpr int kPoolStackWide = CRT_POOL_STACK_WIDE;
constexpr int kPoolNodesWide = CRT_POOL_NODES_WIDE;  // crt_internal.h
// Round 4: the wide arena's split follows the tree too. A large tree or an instance-heavy scene (DevScene::pool_stack = the
// deep split's, scene.cpp) trades the node window — 26 of its tens of thousands of nodes — for stack entries: 4 + 12 instead
// of 3 + 26; and the renderer's kernels, whose rays all carry one mask (UMASK: no mask plane), have one more entry in the
// same bytes: 4 + 26 for small trees, 5 + 12 for large ones — 12 nodes being what the 40 KB arena has left beside five
// entries (measured at 5 + 8 first; 12: PointInstancedMedCity +1.3 %, its root and eight prototype roots then all sit in
// the window). (Four-wave kernels at 3 + 26 against three-wave ones at 10 + 16, Mray/s: stress 2521 / 2466, the
// 7 M-triangle scene 3172 / 3307; at 6 + 72 on three waves: stress 2249 — the depth of the LDS part of the stack is what
// large trees pay for, profiles/README.md.)
#ifndef CRT_POOL_NODES_WIDE_DEEP
#define CRT_POOL_NODES_WIDE_DEEP 12
#endif
__host__ __device__ constexpr int wide_stack(bool deep_tree, bool umask) { return kPoolStackWide + (deep_tree ? 1 : 0) + (umask ? 1 : 0); }
__host__ __device__ constexpr int wide_nodes(bool deep_tree) { return deep_tree ? CRT_POOL_NODES_WIDE_DEEP : kPoolNodesWide; }
constexpr int wide_arena(bool deep_tree, bool umask) {
  return (kBlock / 64) * pool_lds_dwords<CRT_POOL_ROWS>(wide_stack(deep_tree, umask), umask) + wide_nodes(deep_tree) * kLdsNodeStride;
}
constexpr int imax(int a, int b) { return a > b ? a : b; }
constexpr int kEngineLdsWide = imax(imax(wide_arena(false, false), wide_arena(false, true)), imax(wide_arena(true, false), wide_arena(true, true)));
static_assert(kEngineLdsWide * 4 + 512 <= 40 * 1024, "four workgroups of the wide split per CU");
// Runs the traversal for one workgroup. `lds` = kEngineLdsDwords dwords (WIDE: kEngineLdsWide), 16-byte aligned.
// Contains a workgroup barrier: call from uniform control flow, after the shared variables the callbacks use are
// initialised.
// WIDE: 0 the three-wave kernels (both engine copies, the scene's split), 1 the four-wave kernels (flat engine only),
// 2 the four-wave kernels of direct-leaf images (DIRECT engine only; the renderer's, round 4).
template <bool ANY, bool STATS, int WIDE, int COLD = (int)kColdAll, bool UMASK = false, class Fetch, class Emit>
__device__ __forceinline__ void run_traversal(const DevScene &S, uint32_t *lds, float t_min, uint32_t &err, LaneStats &st,
                                              Fetch fetch, Emit emit, uint32_t umask = 0u) {
  // the split of the arena is the scene's (uniform): clamp to what the arena was sized for
  const bool deep_tree = S.pool_stack >= (uint32_t)kPoolStackDeep;
  const uint32_t pstack = WIDE ? (uint32_t)wide_stack(deep_tree, UMASK)
                               : (deep_tree ? (uint32_t)kPoolStackDeep : (uint32_t)kPoolStack);
  const int pnodes = WIDE ? wide_nodes(deep_tree) : (deep_tree ? kPoolNodesDeep : kPoolNodes);
  const int wave_dwords = pool_lds_dwords<CRT_POOL_ROWS>((int)pstack, UMASK);
  uint32_t *lds_nodes = lds + (kBlock / 64) * wave_dwords;
  uint32_t n_lds_pk;
  const uint32_t n_lds = stage_nodes(S, lds_nodes, pnodes, n_lds_pk);  // ends with a barrier
  // Two instantiations, chosen by the scene (uniform, DevScene::direct_leaves: instance-heavy or packet-free scenes):
  // the DIRECT engine reads direct leaf words and tests an instanced tree's root at entry; flat scenes run the engine
  // that knows nothing of either — carrying the direct form's state through the node loop costs the bench scene 2 %,
  // the root test 1.5 % (profiles/README.md).
  uint32_t *wave_lds = lds + (threadIdx.x >> 6) * wave_dwords;
  // COLD (what cold per-ray state the engine keeps, see traverse_pool) is the KERNEL's: the renderer's closest-hit
  // kernels are instantiated per DevScene::cold and chosen on the host — several engine copies inside one kernel cost
  // the register allocator more than the leaner copy saves (the fused kernel went from 11 to 89 spilled registers).
  // Any-hit traversal keeps none of that state by construction; the batched queries report u and v and take kColdAll.
  // The four-wave kernels never meet a direct-leaf scene (crt_internal.h, select_engine).
#ifndef CRT_WIDE_LEAN
#define CRT_WIDE_LEAN 1  // the four-wave kernels fetch a packet in stages (LEAN); 0: whole, as the three-wave ones (A/B)
#endif
#define CRT_ENGINE(D) traverse_pool<ANY, STATS, CRT_POOL_ROWS, D, (WIDE && CRT_WIDE_LEAN != 0), COLD, UMASK>(S, wave_lds, t_min, umask, lds_nodes, n_lds, n_lds_pk, pstack, err, st, fetch, emit)
  if ((COLD & (int)kNoPackets) != 0 || WIDE == 2) CRT_ENGINE(true);  // one copy, the direct engine (select_engine)
  else if (CRT_DIRECT_LEAVES != 0 && !WIDE && S.direct_leaves != 0) CRT_ENGINE(true);
  else CRT_ENGINE(false);
#undef CRT_ENGINE
}

}  // namespace dev
}  // namespace crt
