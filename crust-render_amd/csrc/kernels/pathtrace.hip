// Wavefront path tracer for gfx950: crust-core's integrator (surface arm + carried interior media) as data-parallel
// stages over device-resident queues.
//
// What it computes: crates/crust-core/src/tracer.rs:515-636 (render_pixel), :1086-1558 (trace_path: surface arm,
// :1159-1165 / :1256-1319 / :1352-1360 carried medium), :930-953 (bounce_emission_weight), :1016-1035 (shadow
// test), :85-104 (MIS weights), :1478-1490 (Russian roulette), :1327-1340 (sky), rt_world.rs:207-237,
// light.rs:421-440 — with the backward gather (tracer.rs:1537-1557) restated as the algebraically identical forward
// accumulation
//     L += beta * (atten * emitted) * w     (bounce-hit emission, weighted for the previous vertex)
//     L += (beta * atten) * emit_here       (primary vertex / after a medium scatter)
//     L += (beta * atten) * nee             (after the shadow ray)
//     beta *= atten * value * cos / pdf     (then / p_survive under roulette)
// in exactly that order, which is the order the CPU oracle's forward mode uses, so the two agree bit for bit.
//
// Stages (device functions; k_path runs them all for a workgroup-private queue segment in one launch, the k_*
// kernels one at a time — no host sync either way):
//   generate : camera samples -> path state (SoA), one slot per (pixel, sample), dealt to the segments round-robin
//   extend   : closest-hit traversal of the segment's live paths            -> hit records
//   shade    : medium free flight, emission + MIS, light sampling -> shadow queue, BSDF sampling, roulette;
//              survivors are compacted into the other state buffer of the SAME segment with a wave ballot +
//              popcount prefix and an LDS counter (no global atomics); finished paths write their radiance to the
//              film staging plane
//   shadow   : any-hit traversal of the segment's shadow queue; unoccluded requests add their contribution
//   resolve  : staging planes are folded into the film in sample order (sum += color, tracer.rs:599)
// Path state is struct-of-arrays of 16-byte records, so every stage's loads and stores are unit-stride across a wave.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "shade.hip.h"
#include "traverse_pool.hip.h"

namespace crt {

using namespace dev;

namespace {

// Occupancy targets (second __launch_bounds__ argument = waves per SIMD; caps the VGPR allocation).
#ifndef CRT_EXTEND_WAVES
#define CRT_EXTEND_WAVES 3
#endif
#ifndef CRT_SHADE_WAVES
#define CRT_SHADE_WAVES 3
#endif
#ifndef CRT_SHADOW_WAVES
#define CRT_SHADOW_WAVES 3
#endif

enum { K_CAMERA = 0, K_PATH = 1, K_TIME = 2 };                 // tracer.rs:20-22 (off root)
enum { K_NEE = 0, K_BSDF = 2, K_PHASE = 4, K_RR = 5, K_MEDIUM = 6 };  // tracer.rs:23-30 (off vertex)
constexpr int kRrStartBounce = 3;                              // tracer.rs:46
constexpr float kRrMinProb = 0.05f;                            // tracer.rs:47
constexpr uint32_t kFilmTarget = 0x80000000u;
constexpr uint32_t kPrevValid = 1u << 16, kPrevDelta = 1u << 17;
constexpr int kMediumShift = 18;             // aux[18:32): 1-based compact id of the interior medium the ray travels in
constexpr uint32_t kMaxMedia = (1u << 14) - 1;

// Streaming accesses (path state, hit records, queues, staging film: written once, read once a whole segment pass
// later — far beyond any cache) carry the non-temporal hint, so they do not push the BVH's nodes, instance records
// and instance records out of L2: +0.8 % (veach_mis) to +2.6 % (PointInstancedMedCity 3840x2160) over plain accesses
// (profiles/README.md, r02h). CRT_NT=0 builds plain accesses (A/B).
#ifndef CRT_NT
#define CRT_NT 1
#endif
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(float4 *p, float4 v) {
  if (CRT_NT) { nt_f4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<nt_f4 *>(p)); }
  else *p = v;
}
__device__ __forceinline__ void st_nt(uint4 *p, uint4 v) {
  if (CRT_NT) { nt_u4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<nt_u4 *>(p)); }
  else *p = v;
}
__device__ __forceinline__ void st_nt(uint32_t *p, uint32_t v) {
  if (CRT_NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
__device__ __forceinline__ void st_nt(float *p, float v) {
  if (CRT_NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
// Hit records are the one stream written OUT OF ORDER (a ray's record goes out when its traversal ends): 16 + 4 bytes
// into two planes, each a partial 32-byte sector. CRT_NT_HIT=0 writes them as plain stores, which L2 may merge with the
// neighbouring slots' before they leave for HBM (A/B: profiles/README.md, round 3).
#ifndef CRT_NT_HIT
#define CRT_NT_HIT CRT_NT
#endif
__device__ __forceinline__ void st_hit(float4 *p, float4 v) { if (CRT_NT_HIT) st_nt(p, v); else *p = v; }
__device__ __forceinline__ void st_hit(uint32_t *p, uint32_t v) { if (CRT_NT_HIT) st_nt(p, v); else *p = v; }
__device__ __forceinline__ float4 ld_nt(const float4 *p) {
  if (CRT_NT) { const nt_f4 x = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p)); return make_float4(x.x, x.y, x.z, x.w); }
  return *p;
}

// Path state: struct-of-arrays of 16-byte records, so each plane is read and written with one
// 16-byte-per-lane (1 KiB per wave) instruction — the widest, most efficient global access on CDNA4.
struct PathSoA {
  float4 *a;    // origin.xyz, dir.x
  float4 *b;    // dir.yz, beta.xy          beta: the running throughput (tracer.rs:1114)
  float4 *c;    // beta.z, L.xyz            L: radiance gathered so far
  uint4 *d;     // sampler pattern of the K_PATH domain | owned-pixel index |
                // n_rec (low 16) + remaining depth (high 16) | sample-in-batch (low 16) + kPrevValid + kPrevDelta
                // + carried-medium id (high 14)
  float4 *e;    // previous vertex position + previous bounce pdf (PrevBounce, tracer.rs:899-906); lit scenes only
  float *time;  // shutter time; scenes with motion only
};
struct HitSoA { float4 *h; uint32_t *geom; };  // t, ray-facing normal; geom id | front_face << 31, ~0 = miss
struct ShadowSoA { float4 *a, *b, *c; };       // origin + tmax; dir + time; contribution + target (bits)

// Queues are segmented per workgroup: workgroup b owns slots [b * seg_cap, (b + 1) * seg_cap) of every
// state buffer and of the shadow queue, and keeps the number of live entries of its segment in seg[]. A
// workgroup compacts its own survivors with a wave ballot + popcount prefix and an LDS counter, so the
// wavefront has NO global atomics on its data path (a single returning atomic per wave on one queue head
// saturates near 88 M/s on this chip — more than the whole shade stage costs).
//
// Direction bins: a workgroup's segment of a PATH buffer is split into kBins sub-segments, one per octant of the
// ray direction, and shade appends every continuing path to the sub-segment of its new direction. The next
// extend then walks the sub-segments in turn, so the 64 rays of a wave share their direction signs (same
// near-to-far child order, similar subtrees) and come from neighbouring pixels: secondary rays regain most of
// the coherence of camera rays without a sort pass. Price: kBins x the slot index space of the path and hit
// planes (address space only — untouched slots cost neither bandwidth nor cache; HBM has 288 GB).
// Measured on cornellbox 1080p (profiles/README.md, r01c): 8 octant bins cut extend by only 3 % and cost shade
// 15 % — the divergence is between traversal PHASES, not between child orders — so the default is 1 bin.
#ifndef CRT_BINS
#define CRT_BINS 1
#endif
constexpr int kBins = CRT_BINS;
constexpr int kMaxGrid = 16384;
struct Counters {
  uint32_t err;
  uint32_t pad;
  unsigned long long stats[8];      // RayStats (stats.rs:128-147) in declaration order
  uint32_t seg[2][kMaxGrid * kBins];  // live paths of state buffer 0 / 1, per (workgroup, bin)
  uint32_t shadow[kMaxGrid];          // shadow requests per workgroup
  // crt_renderer_shade_class_stats: per material class, the wave executions of the vertex step that contained a
  // vertex of the class, and how many of their 64 lanes held one (lanes / (64 * waves) = the class's lane utilisation)
  unsigned long long cls_waves[4], cls_lanes[4];
};

__device__ __forceinline__ uint32_t dir_bin(float dx, float dy, float dz) {
  if (kBins == 1) return 0;
  return (dx < 0.0f ? 1u : 0u) | (dy < 0.0f ? 2u : 0u) | (dz < 0.0f ? 4u : 0u);
}
// Prefix sums of a workgroup's kBins sub-segment counts (pre[kBins] = total); ends with a barrier.
// The counts are read as volatile: in the fused kernel they were written earlier in the same launch by another wave
// of this workgroup, so the read must be a fresh vector load, never a scalar-cache or hoisted one.
__device__ __forceinline__ void bins_prefix(const uint32_t *counts, uint32_t *pre /* LDS, kBins + 1 */) {
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int b = 0; b < kBins; b++) { pre[b] = acc; acc += ((const volatile uint32_t *)counts)[b]; }
    pre[kBins] = acc;
  }
  __syncthreads();
}
// Slot of the k-th live path of this workgroup (sub-segments concatenated in bin order).
__device__ __forceinline__ uint32_t bin_slot(uint32_t k, const uint32_t *pre, uint32_t seg_cap) {
  uint32_t b = 0;
#pragma unroll
  for (int t = 1; t < kBins; t++) b += k >= pre[t] ? 1u : 0u;
  return (blockIdx.x * kBins + b) * seg_cap + (k - pre[b]);
}

struct Params {
  DevScene scene;
  uint32_t sample_begin;
  const CrtMaterial *materials;
  uint32_t n_materials;
  // geom_id -> record of `materials` (and of `media`, `mat_class`), or nullptr: the table is indexed by geom_id itself.
  // World::attach binds a material per geometry (rt_world.rs:111-122) and an instanced city binds the same few looks
  // tens of thousands of times: PointInstancedMedCity has 40 008 geometry ids and 9.3 MB of records that are 133
  // distinct ones or fewer — deduplicated (crt_renderer_new) the table stages in LDS like every other scene's.
  const uint16_t *mat_index;
  const DevMedium *media;          // per geom_id: the material's interior medium (present == 0: none)
  const DevMedium *media_by_id;    // the present ones, by compact id - 1
  const CrtLight *lights;
  uint32_t n_lights;
  uint32_t has_inf_lights;  // any DISTANT / DOME entry: escaping rays then run escaped_emission
  CrtCamera camera;
  uint32_t width, height, max_depth;
  int32_t frame, strategy, filter_kind;
  float filter_radius;
  uint32_t has_motion;
  const uint32_t *pixel_index;  // owned pixels: j * width + i
  uint32_t n_pix;
  // Adaptive stopping (tracer.rs:609-617): the pixels still sampling. active == nullptr: all n_pix, in order.
  // A path carries its ACTIVE slot; the staging film is indexed [sample][active slot].
  const uint32_t *active;
  uint32_t n_act;
  uint32_t seg_cap;             // slots per workgroup segment
  // Material classes (shade's partition key): one byte per geom_id, and whether more than one class occurs at all —
  // a one-class scene (cornellbox, MedCity) keeps the single pending ring and pays nothing.
  const uint8_t *mat_class;
  uint32_t partition;
  uint32_t class_stats;         // count Counters::cls_* in the vertex step (diagnostic, off by default)
  // Per-stage pipeline, pinhole camera, no motion blur: a camera path is the 16 bytes (direction, sampler pattern) in
  // plane `a` — its origin is the camera's, its throughput 1, its radiance 0, pixel / sample / depth follow from the slot
  // number (generate_segment). Set per batch by Renderer::render.
  uint32_t cam_compact;
};

__device__ __forceinline__ float light_weight(int s, float light_pdf, float bounce_pdf) {  // tracer.rs:85-92
  switch (s) {
    case CRT_STRATEGY_POWER: return power_heuristic(light_pdf, bounce_pdf);
    case CRT_STRATEGY_BALANCE: return balance_heuristic(light_pdf, bounce_pdf);
    case CRT_STRATEGY_LIGHT: return 1.0f;
    default: return 0.0f;
  }
}
__device__ __forceinline__ float bounce_weight(int s, float bounce_pdf, float light_pdf) {  // tracer.rs:97-104
  switch (s) {
    case CRT_STRATEGY_POWER: return power_heuristic(bounce_pdf, light_pdf);
    case CRT_STRATEGY_BALANCE: return balance_heuristic(bounce_pdf, light_pdf);
    case CRT_STRATEGY_LIGHT: return 0.0f;
    default: return 1.0f;
  }
}
// What an escaping ray picks up (tracer.rs:1321-1342 with escaped_emission, :966-1009): every light at infinity
// covering the direction, MIS-weighted against the previous vertex's NEE; the sky gradient only where none covers.
__device__ __forceinline__ V3 sky_gradient(V3 unit_direction) {
  const float t = 0.5f * (unit_direction.y + 1.0f);
  return v3(1.0f, 1.0f, 1.0f) * (1.0f - t) + v3(0.5f, 0.7f, 1.0f) * t;
}
__device__ __forceinline__ V3 escaped_background(const CrtLight *lights, uint32_t n_lights, int strategy,
                                                 V3 unit_direction, bool competing, float prev_pdf) {
  V3 background = splat(0.0f);
  bool covered = false;
  for (uint32_t k = 0; k < n_lights; k++) {
    V3 emitted;
    float pdf;
    if (!light_escaped(lights[k], unit_direction, emitted, pdf)) continue;
    covered = true;
    float weight = 1.0f;
    if (competing && strategy != CRT_STRATEGY_BSDF) {
      const float light_pdf = rmax(pdf / (float)n_lights, 1e-6f);
      weight = bounce_weight(strategy, prev_pdf, light_pdf);
    }
    background = background + emitted * weight;
  }
  if (!covered) background = background + sky_gradient(unit_direction);
  return background;
}

__device__ __forceinline__ bool lds_take(bool want, uint32_t *next, uint32_t limit, uint32_t &idx) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0) return false;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(next, (uint32_t)__popcll(mask));
  base = __shfl(base, leader, 64);
  idx = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  return want && idx < limit;
}

// Segment-local append: every lane with `want` gets a unique slot of its workgroup's segment. The wave's
// lanes are ranked with a ballot + popcount prefix; one LDS atomic per wave reserves the run.
__device__ __forceinline__ uint32_t seg_append(bool want, uint32_t *lds_counter) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0) return 0;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(lds_counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader, 64);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Per-workgroup reduction of a statistics counter: LDS first, then one global atomic per workgroup.
__device__ __forceinline__ void add_stat(uint32_t *lds_slot, uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(lds_slot, v);
}

// ---- interior media of the materials (openpbr.rs:225-258), derived once on the device so that the logarithms
// are the same deterministic sequence the shading kernels use. Two launches: every record in parallel (40 008 of them on
// the instanced city: one thread looping took 31.6 ms), then ONE workgroup numbers the present ones in material order —
// each thread counts a contiguous run of the table, an LDS prefix over the 1024 runs, ids handed out run by run. ----
__global__ void k_build_media(const CrtMaterial *materials, uint32_t n, DevMedium *media) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DevMedium m;
  medium_from_material(materials[i], m);
  m.id = 0;
  media[i] = m;
}
constexpr int kNumberBlock = 1024;
__global__ __launch_bounds__(kNumberBlock) void k_number_media(uint32_t n, DevMedium *media, DevMedium *by_id, uint32_t *count_out) {
  __shared__ uint32_t run_sum[kNumberBlock];
  const uint32_t per = (n + kNumberBlock - 1) / kNumberBlock;
  const uint32_t lo = threadIdx.x * per < n ? threadIdx.x * per : n, hi = lo + per < n ? lo + per : n;
  uint32_t mine = 0;
  for (uint32_t i = lo; i < hi; i++) mine += media[i].present ? 1u : 0u;
  run_sum[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < kNumberBlock; off <<= 1) {  // inclusive Hillis-Steele prefix
    const uint32_t add = (int)threadIdx.x >= off ? run_sum[threadIdx.x - off] : 0u;
    __syncthreads();
    run_sum[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t count = run_sum[threadIdx.x] - mine;  // present records before this run
  for (uint32_t i = lo; i < hi; i++) {
    if (!media[i].present) continue;
    count++;
    if (count <= kMaxMedia) {
      media[i].id = count;
      by_id[count - 1] = media[i];
    }
  }
  if (threadIdx.x == kNumberBlock - 1) *count_out = run_sum[threadIdx.x];
}

// ---- generate: PathSampler::new(...).new_domain(tile), camera sample, camera ray (tracer.rs:559-585) ----
// Every stage body below is a device function working on the calling workgroup's own queue segment, with its big
// LDS buffer passed in: the per-stage kernels hand it their own array, the fused kernel (k_path) one shared arena.
constexpr int kArenaDwords = kEngineLdsDwords > kSobolLdsWords ? kEngineLdsDwords : kSobolLdsWords;
static_assert(sizeof(CrtMaterial) % 4 == 0, "material records are copied as dwords");
// Shade's pending rings (one per material class) sit at the END of the arena, the material table between the Sobol
// tables and the rings.
constexpr int kClasses = 4;
constexpr uint32_t kRing = 2 * kBlock;                 // entries per class ring: < kBlock pending + one classify batch
constexpr int kRingDwords = kClasses * (int)kRing;
constexpr int kClassLdsMax = 1024;                     // geom ids whose class byte is staged in LDS (1 KB)
constexpr int mat_lds_max(int arena) { return (arena - kSobolLdsWords - kRingDwords - kClassLdsMax / 4) / (int)(sizeof(CrtMaterial) / 4); }
// The per-stage shade kernel of simple-material scenes without lights at infinity also runs four workgroups per CU
// (128 registers, a 40 KB arena: cornellbox +2 %); the other instances spill too much at 128 registers (sun_sky -16 %).
#ifndef CRT_SHADE_WIDE_WAVES
#define CRT_SHADE_WIDE_WAVES 4  // workgroups per CU of the WIDE shade kernel (5: 96 registers, a 31 KB arena — A/B, round 3)
#endif
constexpr int kArenaWide = (160 / CRT_SHADE_WIDE_WAVES) * 1024 / 4 - 256;
static_assert(mat_lds_max(kArenaWide) >= 16 && mat_lds_max(kArenaDwords) >= 16, "the arena holds the Sobol tables, the rings and a material table");

// Material class = which arms of the vertex code a hit on this material runs (openpbr.rs:1026-1136 dispatches one of
// five lobes + thin film + dispersion per lane; rt_world.rs:219-231 binds the material per geom_id). Waves whose
// 64 vertices share a class skip the other classes' lobes with a scalar branch instead of idling through them.
//   0 emissive (Emissive: emission only, never scatters)      1 base: diffuse + specular (dielectric or metal)
//   2 layered: coat and / or fuzz and / or thin film on top    3 transmissive or subsurface (refraction, interior medium)
inline uint8_t material_class(const CrtMaterial &m) {
  if (m.kind == CRT_MAT_EMISSIVE) return 0;
  if (m.transmission_weight > 0.0f || m.subsurface_weight > 0.0f) return 3;
  if (m.coat_weight > 0.0f || m.fuzz_weight > 0.0f || m.thin_film_weight > 0.0f) return 2;
  return 1;
}

// One camera sample (tracer.rs:559-585): PathSampler::new(...).new_domain(tile), filter jitter, lens, primary ray, the
// K_PATH domain's pattern. g = the sample's global number within the batch: active pixel g % n_act, sample g / n_act.
struct CameraSample { V3 o, d; float time; uint32_t pattern, pix, sl; };
__device__ __forceinline__ CameraSample camera_sample(const Params &P, size_t g, uint32_t sample_begin, const uint32_t *sobol_tab) {
  CameraSample cs;
  cs.pix = (uint32_t)(g % P.n_act); cs.sl = (uint32_t)(g / P.n_act);  // pix: active slot
  const uint32_t lin = P.pixel_index[P.active ? P.active[cs.pix] : cs.pix];
  const uint32_t px = lin % P.width, py = lin / P.width;
  const int tile = (int)(px >> 8) + (int)(py >> 8) * 4096;  // tracer.rs:543
  const Sampler root = new_domain(sampler_new((int)px, (int)py, P.frame, (int)(sample_begin + cs.sl)), tile);
  float cam[4];
  draw_sample4(new_domain(root, K_CAMERA), cam, sobol_tab);
  const float fx = filter_offset(P.filter_kind, P.filter_radius, cam[0]);
  const float fy = filter_offset(P.filter_kind, P.filter_radius, cam[1]);
  const float u = ((float)px + fx) / (float)P.width;
  const float v = ((float)py + fy) / (float)P.height;
  cs.time = 0.0f;
  if (P.has_motion) {  // tracer.rs:579-583
    float t4[4];
    draw_sample4(new_domain(root, K_TIME), t4, sobol_tab);
    cs.time = t4[0];
  }
  camera_get_ray(P.camera, u, v, cam[2], cam[3], cs.o, cs.d);
  cs.pattern = new_domain(root, K_PATH).pattern;  // tracer.rs:1101
  return cs;
}
// Camera samples are dealt to the workgroup segments in round-robin chunks of one workgroup's width: slot k of
// segment b holds global sample g = ((k / 256) * G + b) * 256 + k % 256. Every segment is then a uniform
// sample of the frame (sky and geometry alike), so the per-segment work stays balanced at every bounce, while
// a wave still holds 64 consecutive pixels of one 16x16 tile (coherent primary rays).
__device__ __forceinline__ size_t camera_slot_sample(size_t k) {
  return ((k / kBlock) * (size_t)gridDim.x + blockIdx.x) * kBlock + (k % kBlock);
}

// A compact camera path (Params::cam_compact) read back: direction and pattern from plane `a`, origin = the pinhole
// camera's (camera_get_ray with no lens offset: origin + 0), pixel and sample from the slot's sample number (< 2^31:
// ensure_buffers). D = plane d's words as generate would have written them.
__device__ __forceinline__ void compact_camera_path(const Params &P, uint32_t k, float4 a, float4 &A, float4 &B, uint4 &D) {
  const uint32_t g = (uint32_t)camera_slot_sample(k);
  const V3 o = ld3(P.camera.origin) + splat(0.0f);
  A = make_float4(o.x, o.y, o.z, a.x);
  B = make_float4(a.y, a.z, 1.0f, 1.0f);
  D = make_uint4(__float_as_uint(a.w), g % P.n_act, (P.max_depth & 0xffffu) << 16, g / P.n_act);
}

// Plane `c` of a camera path is a constant (throughput 1, radiance 0): the per-stage pipeline neither writes it here nor
// reads it in its first shade (17 GB per 531 M-path batch, +1.3 %; the fused kernel keeps the plain form — measured
// slower there with the special case). CRT_REGEN_FIRST=1 goes further — plane `d` (sampler pattern, pixel, depth, sample)
// is not written either and the first shade reads NOTHING of the path state: it works the camera sample out again from
// the slot number (Sobol tables in LDS). Measured on the bench (round 3, profiles/README.md): generate + resolve -1.2 ms
// per step, shade +3 ms — the first shade is bound by its dependent round trips and barriers, not by the 68-116 bytes
// per path this saves, and the ~350 more instructions per path are not free. Off.
#ifndef CRT_REGEN_FIRST
#define CRT_REGEN_FIRST 0
#endif
constexpr bool kRegenFirst = CRT_REGEN_FIRST != 0;
#ifndef CRT_CAM_COMPACT_BUILD
#define CRT_CAM_COMPACT_BUILD 1  // 0: the 16-byte camera path form is compiled out of generate, extend and shade (A/B)
#endif
#ifndef CRT_MAT_INDEX_BUILD
#define CRT_MAT_INDEX_BUILD 1    // 0: shade indexes the material table by geometry id only (A/B; the host then never deduplicates)
#endif
static_assert(kBins == 1 || !CRT_REGEN_FIRST, "the first shade takes a camera path's sample number from its slot: one sub-segment per workgroup");
__device__ __forceinline__ void generate_segment(const Params &P, const PathSoA &S, Counters *C, uint32_t sample_begin,
                                                 uint32_t n_samples, uint32_t *sobol_tab /* kSobolLdsWords */, bool write_c) {
  sobol_tables_init(sobol_tab);
  const size_t total = (size_t)P.n_act * n_samples;
  const size_t seg0 = (size_t)blockIdx.x * kBins * P.seg_cap;  // camera rays are coherent as dealt: all in bin 0
  uint32_t seg_n = 0;
  for (size_t k = threadIdx.x; k < P.seg_cap; k += kBlock) {
    const size_t g = camera_slot_sample(k);
    if (g >= total) break;
    seg_n = (uint32_t)k + 1;
    const size_t i = seg0 + k;
    const CameraSample cs = camera_sample(P, g, sample_begin, sobol_tab);
    if (CRT_CAM_COMPACT_BUILD && !write_c && P.cam_compact) {  // 16 of the 48-64 bytes: the rest is constant or follows from the slot
      st_nt(&S.a[i], make_float4(cs.d.x, cs.d.y, cs.d.z, __uint_as_float(cs.pattern)));
      continue;
    }
    st_nt(&S.a[i], make_float4(cs.o.x, cs.o.y, cs.o.z, cs.d.x));
    st_nt(&S.b[i], make_float4(cs.d.y, cs.d.z, 1.0f, 1.0f));
    if (write_c) st_nt(&S.c[i], make_float4(1.0f, 0.0f, 0.0f, 0.0f));
    if (write_c || !kRegenFirst) st_nt(&S.d[i], make_uint4(cs.pattern, cs.pix, (P.max_depth & 0xffffu) << 16, cs.sl));
    if (P.has_motion) st_nt(&S.time[i], cs.time);
  }
  // live slots of this segment = 1 + the largest valid k over the workgroup (valid k form a prefix)
  __shared__ uint32_t seg_max;
  if (threadIdx.x == 0) seg_max = 0;
  __syncthreads();
  if (seg_n) atomicMax(&seg_max, seg_n);
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int b = 0; b < kBins; b++) {
      C->seg[0][blockIdx.x * kBins + b] = b == 0 ? seg_max : 0;
      C->seg[1][blockIdx.x * kBins + b] = 0;
    }
    C->shadow[blockIdx.x] = 0;
    if (blockIdx.x == 0) atomicAdd(&C->stats[0], (unsigned long long)total);  // camera_rays (tracer.rs:585)
  }
}
__global__ __launch_bounds__(kBlock) void k_generate(Params P, PathSoA S, Counters *C, uint32_t sample_begin,
                                                     uint32_t n_samples) {
  __shared__ uint32_t sobol_tab[kSobolLdsWords];
  generate_segment(P, S, C, sample_begin, n_samples, sobol_tab, false);
}

// ---- extend: World::intersect (rt_world.rs:207-232) for every live path ----
// COMPACT: the 16-byte camera path form may occur (per-stage launches only: the fused kernel never sets it, and its
// fetch has no registers for the branch — 13 -> 27, 33 -> 97, 52 -> 122 spilled registers in the lit k_path instances).
template <bool STATS, int WIDE, int COLD, bool COMPACT>
__device__ __forceinline__ void extend_segment(const Params &P, const PathSoA &S, const HitSoA &H, Counters *C, int cur,
                                               int first, CrtTravStats *tstats, uint32_t *engine_lds) {
  __shared__ uint32_t pre[kBins + 1];
  bins_prefix(&C->seg[cur][blockIdx.x * kBins], pre);
  const uint32_t n = pre[kBins];
  if (n == 0) return;  // uniform per workgroup
  __shared__ uint32_t next;
  if (threadIdx.x == 0) next = 0;
  LaneStats st = {};
  uint32_t err = 0, done = 0;
  // every path of the first round is a camera ray, every later one an indirect ray (camera.rs:83, tracer.rs:1516-1519)
  const uint32_t mask = first ? CRT_MASK_CAMERA : CRT_MASK_INDIRECT;
  auto fetch = [&](bool want, RayIn &in) -> bool {
    uint32_t k;
    if (!lds_take(want, &next, n, k)) return false;
    const uint32_t i = bin_slot(k, pre, P.seg_cap);
    if (CRT_CAM_COMPACT_BUILD && COMPACT && first && P.cam_compact) {  // uniform: camera paths as 16-byte records (generate_segment)
      const float4 A = ld_nt(&S.a[i]);
      const V3 o = ld3(P.camera.origin) + splat(0.0f);
      in.ox = o.x; in.oy = o.y; in.oz = o.z; in.dx = A.x; in.dy = A.y; in.dz = A.z;
      in.time = 0.0f;
    } else {
      const float4 A = ld_nt(&S.a[i]), B = ld_nt(&S.b[i]);
      in.ox = A.x; in.oy = A.y; in.oz = A.z; in.dx = A.w; in.dy = B.x; in.dz = B.y;
      in.time = P.has_motion ? S.time[i] : 0.0f;
    }
    in.mask = mask; in.t_min = 0.001f; in.t_max = CRT_INF; in.slot = i;
    return true;
  };
  auto emit = [&](uint32_t i, bool hit, const Hit &h, float dx, float dy, float dz) {
    if (hit) {
      const bool front = dot3(dx, dy, dz, h.nx, h.ny, h.nz) < 0.0f;  // scene.rs:356-359
      st_hit(&H.h[i], make_float4(h.t, front ? h.nx : -h.nx, front ? h.ny : -h.ny, front ? h.nz : -h.nz));
      st_hit(&H.geom[i], h.geom | (front ? 0x80000000u : 0u));
    } else {
      st_hit(&H.geom[i], kInvalid);
    }
    done++;
  };
  // the four-wave kernels keep no mask plane: every ray of the launch carries `mask` (UMASK, traverse_pool.hip.h)
  run_traversal<false, STATS, WIDE, COLD, WIDE != 0>(P.scene, engine_lds, 0.001f, err, st, fetch, emit, mask);
  if (err) atomicOr(&C->err, err);
  if (STATS) flush_stats(st, tstats, done);
}
// WIDE: the four-workgroups-per-CU split of the engine (traverse_pool.hip.h), the per-stage pipeline of flat scenes.
// COLD: the cold per-ray state the scene can need (DevScene::cold; the hit record carries no u, v, so a scene needs
// kColdUV only for shading normals): 0 for flat-shaded static triangle scenes, kColdNormal with spheres or nested
// instances, kColdAll otherwise and for the stats build.
template <bool STATS, int WIDE, int COLD>
__global__ __launch_bounds__(kBlock, WIDE ? 4 : CRT_EXTEND_WAVES) void k_extend(Params P, PathSoA S, HitSoA H, Counters *C, int cur,
                                                                     int first, CrtTravStats *tstats) {
  __shared__ __attribute__((aligned(16))) uint32_t engine_lds[WIDE ? kEngineLdsWide : kEngineLdsDwords];
  extend_segment<STATS, WIDE, COLD, true>(P, S, H, C, cur, first, tstats, engine_lds);
}

// ---- shade: one iteration of trace_path's loop body for every live path (tracer.rs:1118-1530) ----
// LIT: the light list is not empty. Unlit scenes (cornellbox, the bench scene: sky only) run an instance without the
// light-sampling block, its second BSDF evaluation, the emitter MIS weights and the previous-vertex plane — code they
// never execute, but whose registers and scalar constants the fused kernel otherwise carries through every bounce
// (cornellbox +1.3 %, MedCity +2 %, profiles/README.md). With lights present the strategy is checked at run time.
// MATS: what the scene's material table holds — 0 simple (no coat, fuzz, thin film, transmission, subsurface anywhere:
// the OpenPBR code is instantiated without those arms, shade.hip.h), 1 general, 2 general with interior media.
template <int MATS, bool INF, bool LIT, int ARENA>
__device__ __forceinline__ void shade_segment(const Params &P, const PathSoA &S, const PathSoA &N, const HitSoA &H,
                                              const ShadowSoA &Q, Counters *C, int cur, float4 *staging,
                                              uint32_t *sobol_tab /* ARENA dwords: Sobol tables, then the material table */,
                                              bool first /* camera paths whose plane c was not written (generate_segment) */,
                                              bool all_pending = false /* no CLASSIFY pass: every path takes the vertex step */) {
  constexpr bool MEDIA = MATS == 2, SIMPLE = MATS == 0;
  // Two of round 3's forms are compiled only into the instances that have registers to spare: the lit four-wave shade
  // kernel sits at its 128-register limit with 17 spilled, and the few branches of either form cost it 3.6 % / 0.9 %
  // (cornellbox_guided) where the unlit one gains 1 % (profiles/README.md). The host sets Params::cam_compact for
  // unlit scenes only and runs a deduplicated table (Params::mat_index) through the three-wave shade kernel.
  constexpr bool COMPACT = CRT_CAM_COMPACT_BUILD && !LIT;
  constexpr bool MAT_INDEX = CRT_MAT_INDEX_BUILD && ARENA != kArenaWide;
  __shared__ uint32_t lds_ctr[10];  // [1] shadow requests, [2..8] statistics
  __shared__ uint32_t out_n[kBins];  // survivors per direction bin
  __shared__ uint32_t pre[kBins + 1];
  bins_prefix(&C->seg[cur][blockIdx.x * kBins], pre);
  const uint32_t n = pre[kBins];
  if (n == 0) {  // nothing lives in this segment (uniform per workgroup): publish empty outputs and leave
    if (threadIdx.x < kBins) C->seg[1 - cur][blockIdx.x * kBins + threadIdx.x] = 0;
    if (threadIdx.x == 0) C->shadow[blockIdx.x] = 0;
    return;
  }
  if (threadIdx.x < 10) lds_ctr[threadIdx.x] = 0;
  if (threadIdx.x < kBins) out_n[threadIdx.x] = 0;
  sobol_tables_init(sobol_tab);  // ends with a workgroup barrier
  // The vertex code reads its material record a field at a time, where each lobe needs it: two dozen dependent
  // round trips to L2 per vertex. A table that fits the rest of the arena is staged in LDS once per call instead
  // (generic pointer: the reads become FLAT loads served by LDS); larger tables stay in global memory.
  const CrtMaterial *mats = P.materials;
  if (P.n_materials <= (uint32_t)mat_lds_max(ARENA)) {
    uint32_t *dst = sobol_tab + kSobolLdsWords;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(P.materials);
    const uint32_t words = P.n_materials * (uint32_t)(sizeof(CrtMaterial) / 4);
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) dst[w] = src[w];
    mats = reinterpret_cast<const CrtMaterial *>(dst);
    __syncthreads();
  }
  const uint32_t seg0 = blockIdx.x * P.seg_cap;  // shadow queue: one unbinned segment per workgroup
  const uint32_t n_lights = LIT ? P.n_lights : 0u;
  uint32_t s_closest = 0, s_shadow = 0, s_vertices = 0, s_rr_t = 0, s_rr_k = 0, s_esc = 0, s_depth = 0;
  // Two kinds of work per path: a ray that escaped just adds the sky and ends (cheap), everything else runs the full
  // vertex (expensive). Mixed in one wave the escaped lanes would idle through the vertex code — a third of the
  // rays on an open scene — so the segment is swept in two interleaved steps: CLASSIFY takes the next 256 paths,
  // finishes the escaped ones on the spot and appends the others to a ring of pending indices in LDS; SHADE takes
  // 256 pending paths at a time (fewer only when the input is exhausted), so the vertex code runs on full waves.
  // PARTITION (scenes with more than one material class): the pending vertices are kept in one ring PER CLASS and a
  // SHADE step takes its 256 from ONE ring — the fullest — so all four waves of the step run one class's arms. The
  // class comes from a byte table by geom_id (LDS when it fits). Per-path arithmetic does not depend on which lanes
  // share a wave, and every output slot is private to its path, so images and counters do not change.
  uint32_t *ring = sobol_tab + ARENA - kRingDwords;               // [kClasses][kRing]
  uint8_t *cls_lds = reinterpret_cast<uint8_t *>(ring) - kClassLdsMax;
  __shared__ uint32_t ring_tail[kClasses];  // entries ever appended, per class (heads are tracked in registers: uniform)
  const bool part = P.partition != 0;
  const bool cls_in_lds = part && P.n_materials <= (uint32_t)kClassLdsMax;
  if (threadIdx.x < kClasses) ring_tail[threadIdx.x] = 0;
  if (cls_in_lds)
    for (uint32_t g = threadIdx.x; g < P.n_materials; g += kBlock) cls_lds[g] = P.mat_class[g];
  __syncthreads();
  uint32_t next_in = 0;
  uint32_t head[kClasses], tail[kClasses];
#pragma unroll
  for (int c = 0; c < kClasses; c++) head[c] = 0;
  for (;;) {
#pragma unroll
    for (int c = 0; c < kClasses; c++) tail[c] = ring_tail[c];  // uniform: read between two barriers, nobody appends meanwhile
    __syncthreads();
    auto most_pending = [&](int &c_best) {
      uint32_t best = tail[0] - head[0];
      c_best = 0;
#pragma unroll
      for (int c = 1; c < kClasses; c++)
        if (tail[c] - head[c] > best) { best = tail[c] - head[c]; c_best = c; }
      return best;
    };
    int c_sel;
    // ---- CLASSIFY while no class has a workgroup's worth of vertices pending and input remains ----
    while (most_pending(c_sel) < (uint32_t)kBlock && next_in < n) {
      const uint32_t k_c = next_in + threadIdx.x;
      next_in += kBlock;
      bool pending = false;
      uint32_t cls = 0;
      // all_pending (per-stage pipeline, bounces after the first, one material class): CLASSIFY reads nothing and sends
      // every path to the vertex step, whose own escaped arm adds the sky (same expression, same counters). The escaped
      // lanes then idle through the vertex code of their wave — a quarter to a third of the lanes at bounces 1-3 of the
      // bench — but the path state is read ONCE: CLASSIFY's 68 bytes per path were read a second time, from HBM, by the
      // vertex step of every path that had hit something.
      if (all_pending && !part) pending = k_c < n;
      else if (k_c < n) {
        const uint32_t i_c = bin_slot(k_c, pre, P.seg_cap);
        const uint32_t hg = H.geom[i_c];
        // the escaped arm's operands are requested together with the classification's (one memory round trip, not
        // two); for a path that hit something they are loaded again by the vertex step, from L2. (Requested for the
        // escaped lanes only — 48 B less per hit path — the per-stage shade is 1.5 % slower: profiles/README.md.)
        uint4 D;
        float4 A, B, Cc = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
        if (COMPACT && first && P.cam_compact) {  // uniform
          float4 a = S.a[i_c];
          asm volatile("" : "+v"(a.x));
          compact_camera_path(P, k_c, a, A, B, D);
        } else if (!(kRegenFirst && first)) {
          D = S.d[i_c]; A = S.a[i_c]; B = S.b[i_c];
          if (!first) Cc = S.c[i_c];
          asm volatile("" : "+v"(A.x), "+v"(B.x), "+v"(Cc.x));
        } else {
          // a camera path: only the direction (escaped rays) and the film slot are needed here, and both follow from the
          // slot number; nothing of the path state is read (generate_segment)
          D = make_uint4(0u, 0u, (P.max_depth & 0xffffu) << 16, 0u);
          A = make_float4(0.0f, 0.0f, 0.0f, 0.0f); B = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
          if (hg == kInvalid && P.max_depth > 0) {
            const CameraSample cs = camera_sample(P, camera_slot_sample(k_c), P.sample_begin, sobol_tab);
            A.w = cs.d.x; B.x = cs.d.y; B.y = cs.d.z;
            D.y = cs.pix; D.w = cs.sl;
          }
        }
        const int remaining = (int)(D.z >> 16);
        const bool carries_medium = MEDIA && (D.w >> kMediumShift) != 0;
        if (!INF && hg == kInvalid && remaining > 0 && !carries_medium) {
          // tracer.rs:1321-1342: the path ends on the sky gradient. (With lights at infinity — INF — escaped rays
          // take the full vertex step instead: their background is a loop over the light list.)
          const V3 rd = v3(A.w, B.x, B.y), beta = v3(B.z, B.w, Cc.x);
          V3 L = v3(Cc.y, Cc.z, Cc.w);
          s_closest++;
          s_esc++;
          const V3 unit_direction = normalize(rd);
          const V3 background = splat(0.0f) + sky_gradient(unit_direction);
          L = L + beta * background;
          st_nt(&staging[(D.w & 0xffffu) * P.n_act + D.y], make_float4(L.x, L.y, L.z, 0.0f));
        } else {
          pending = true;
          if (part && hg != kInvalid) {
            const uint32_t g = hg & 0x7fffffffu;
            const uint32_t mi_c = (MAT_INDEX && P.mat_index) ? (uint32_t)P.mat_index[g] : g;
            cls = cls_in_lds ? cls_lds[mi_c] : P.mat_class[mi_c];
          }
        }
      }
      if (part) {
#pragma unroll
        for (int c = 0; c < kClasses; c++) {
          const bool mine = pending && cls == (uint32_t)c;
          const uint32_t at = seg_append(mine, &ring_tail[c]);
          if (mine) ring[c * kRing + at % kRing] = k_c;
        }
      } else {
        const uint32_t at = seg_append(pending, &ring_tail[0]);
        if (pending) ring[at % kRing] = k_c;
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < kClasses; c++) tail[c] = ring_tail[c];
      __syncthreads();
    }
    const uint32_t avail = most_pending(c_sel);
    if (avail == 0) break;  // uniform: input exhausted and nothing pending
    // ---- SHADE up to a workgroup's worth of pending vertices of the fullest class ----
    const uint32_t take = avail < (uint32_t)kBlock ? avail : (uint32_t)kBlock;
    const bool active = threadIdx.x < take;
    uint32_t head_sel = head[0];
#pragma unroll
    for (int c = 1; c < kClasses; c++) head_sel = c_sel == c ? head[c] : head_sel;
    const uint32_t k_in = active ? ring[c_sel * kRing + (head_sel + threadIdx.x) % kRing] : 0;
#pragma unroll
    for (int c = 0; c < kClasses; c++) head[c] += c_sel == c ? take : 0u;
    const uint32_t i = active ? bin_slot(k_in, pre, P.seg_cap) : 0;
    bool alive = false, want_shadow = false;
    V3 L = splat(0.0f), beta = splat(1.0f);
    V3 n_o = splat(0.0f), n_d = splat(0.0f), sh_d = splat(0.0f), sh_c = splat(0.0f), hit_p = splat(0.0f);
    float n_ppdf = 0.0f, sh_tmax = 0.0f, time = 0.0f;
    uint32_t meta = 0, aux = 0, pix = 0, pattern = 0, n_med = 0;
    bool n_delta = false, n_prev_valid = true;
    if (active) {
      float4 A, B, Cc = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
      uint4 D;
      if (COMPACT && first && P.cam_compact) {  // uniform
        compact_camera_path(P, k_in, S.a[i], A, B, D);
        time = 0.0f;
      } else if (!(kRegenFirst && first)) {
        A = S.a[i]; B = S.b[i]; D = S.d[i];
        if (!first) Cc = S.c[i];
        time = P.has_motion ? S.time[i] : 0.0f;
      } else {  // a camera path: the sample is worked out again from the slot number (generate_segment)
        const CameraSample cs = camera_sample(P, camera_slot_sample(k_in), P.sample_begin, sobol_tab);
        A = make_float4(cs.o.x, cs.o.y, cs.o.z, cs.d.x); B = make_float4(cs.d.y, cs.d.z, 1.0f, 1.0f);
        D = make_uint4(cs.pattern, cs.pix, (P.max_depth & 0xffffu) << 16, cs.sl);
        time = cs.time;
      }
      const V3 ro = v3(A.x, A.y, A.z), rd = v3(A.w, B.x, B.y);
      beta = v3(B.z, B.w, Cc.x);
      L = v3(Cc.y, Cc.z, Cc.w);
      pattern = D.x; pix = D.y; meta = D.z; aux = D.w;
      const uint32_t n_rec = meta & 0xffffu;
      const int remaining = (int)(meta >> 16);
      const bool prev_valid = (aux & kPrevValid) != 0, prev_delta = (aux & kPrevDelta) != 0;
      const uint32_t med_id = MEDIA ? aux >> kMediumShift : 0u;  // Ray::medium: the interior this segment travels in
      DevMedium med;
      med.present = 0; med.scattering = 0;
      if (MEDIA && med_id) med = P.media_by_id[med_id - 1];
      n_med = med_id;
      const uint32_t hg = H.geom[i];
      const bool has_hit = hg != kInvalid;
      const uint32_t geom = hg & 0x7fffffffu;
      // the material record's index: the geometry id, or (deduplicated tables) one 2-byte load away
      const uint32_t mat_i = (MAT_INDEX && has_hit && P.mat_index) ? (uint32_t)P.mat_index[geom] : geom;
      if (P.class_stats) {  // uniform; diagnostic only
        const uint32_t my = has_hit ? (uint32_t)P.mat_class[(MAT_INDEX && P.mat_index) ? (uint32_t)P.mat_index[geom] : geom] : 0u;
#pragma unroll
        for (int c = 0; c < kClasses; c++) {
          const unsigned long long m = __ballot(my == (uint32_t)c);
          if (m && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
            atomicAdd(&C->cls_waves[c], 1ull);
            atomicAdd(&C->cls_lanes[c], (unsigned long long)__popcll(m));
          }
        }
      }
      HitRec rec;
      rec.front_face = ((hg >> 31) & 1u) != 0;
      rec.t = 0.0f; rec.normal = splat(0.0f); rec.p = splat(0.0f);
      if (has_hit) {
        const float4 hh = H.h[i];
        rec.t = hh.x;
        rec.normal = v3(hh.y, hh.z, hh.w);
        rec.p = ro + rd * rec.t;  // ray.at(t), rt_world.rs:221
      }
      hit_p = rec.p;

      // bounce_emission_weight (tracer.rs:930-953). A previous bounce exists only where scatter returned a
      // sample, i.e. where eval is available, so the "eval is None" arm reduces to the delta flag.
      auto emission_weight = [&]() -> float {
        if (prev_delta) return 1.0f;
        for (uint32_t k = 0; k < n_lights; k++) {  // LightList::find_by_geom, light.rs:436-440
          if (P.lights[k].geom_id == geom) {
            const float4 E = S.e[i];
            const V3 from = v3(E.x, E.y, E.z);
            const float light_pdf = rmax(solid_angle_pdf(P.lights[k], from, rec.p) / (float)n_lights, 1e-6f);
            return bounce_weight(P.strategy, E.w, light_pdf);
          }
        }
        return 1.0f;
      };

      if (remaining <= 0) {  // tracer.rs:1123-1149: depth exhausted, last-vertex emission only
        s_depth++;
        if (prev_valid) {
          s_closest++;
          if (has_hit) {
            const CrtMaterial &mat = mats[mat_i];
            const float cos_o = fabs_(dot(normalize(rd), rec.normal));
            V3 emitted = mat_emitted_directional<SIMPLE>(mat, cos_o);
            if (len2(emitted) > 0.0f) {
              if (MEDIA && med.present) emitted = emitted * medium_transmittance(med, rec.t);  // tracer.rs:1134-1136
              L = L + beta * (emitted * emission_weight());
            }
          }
        }
      } else {
        s_closest++;
        const Sampler vdom = new_domain(Sampler{pattern, P.sample_begin + (aux & 0xffffu)}, (int)n_rec);  // :1121
        // free-flight candidate in a scattering carried medium (tracer.rs:1159-1165)
        float t_med = CRT_INF;
        if (MEDIA && med.present && med.scattering) t_med = -(log_det(draw_rnd1(new_domain(vdom, K_MEDIUM)))) / med.sigma_bar;
        const float t_surf = has_hit ? rec.t : CRT_INF;
        if (MEDIA && t_med < t_surf) {  // === carried-medium scatter vertex (tracer.rs:1256-1319): no NEE, next emission in full ===
          const V3 pos = ro + rd * t_med;
          float phase_uv[4];
          draw_sample4(new_domain(vdom, K_PHASE), phase_uv, sobol_tab);
          const V3 dir = sample_henyey_greenstein(normalize(rd), med.g, phase_uv[0], phase_uv[1]);
          const V3 factor = (ld3(med.sigma_s) / med.sigma_bar) * medium_chromatic(med, t_med);
          beta = beta * (splat(1.0f) * factor);
          bool survived = true;
          if (n_rec >= (uint32_t)kRrStartBounce) {
            s_rr_t++;
            const float p_survive = rclamp(max_elem(beta), kRrMinProb, 1.0f);
            if (p_survive < 1.0f) {
              if (draw_rnd1(new_domain(vdom, K_RR)) >= p_survive) {
                survived = false;
                s_rr_k++;
              } else {
                beta = beta / p_survive;
              }
            }
          }
          s_vertices++;
          if (survived) {
            alive = true;
            n_o = pos; n_d = dir; n_ppdf = 0.0f; n_delta = false; n_prev_valid = false;  // same medium: n_med stays
          }
        } else if (!has_hit) {  // tracer.rs:1321-1342 (rays that carried a medium out of the scene end here)
          s_esc++;
          const V3 unit_direction = normalize(rd);
          // plane e exists from the first bounce on (generate does not write it): a camera path never loads it
          const bool competing = prev_valid && !prev_delta;
          float prev_pdf = 0.0f;
          if (INF && competing) prev_pdf = S.e[i].w;
          const V3 background = INF ? escaped_background(P.lights, n_lights, P.strategy, unit_direction, competing, prev_pdf)
                                    : splat(0.0f) + sky_gradient(unit_direction);
          L = L + beta * background;
        } else {
          const CrtMaterial &mat = mats[mat_i];
          // tracer.rs:1352-1361: a scattering medium already paid e^{-sigma_bar t} through the free-flight
          // competition, only the chromatic correction remains; a clear one keeps pure Beer-Lambert.
          V3 atten = splat(1.0f);
          if (MEDIA && med.present) atten = splat(1.0f) * (med.scattering ? medium_chromatic(med, rec.t) : medium_transmittance(med, rec.t));
          const float cos_o = fabs_(dot(normalize(rd), rec.normal));
          const V3 emitted = mat_emitted_directional<SIMPLE>(mat, cos_o);
          V3 emit_here = splat(0.0f);
          if (prev_valid) {  // tracer.rs:1372-1381
            if (len2(emitted) > 0.0f) L = L + beta * ((atten * emitted) * emission_weight());
          } else {
            emit_here = emitted;
          }
          const V3 ba = beta * atten;

          // === 1. direct lighting by light sampling (tracer.rs:1394-1445) ===
          if (LIT && P.strategy != CRT_STRATEGY_BSDF && n_lights > 0) {
            float nee_s[4];
            draw_sample4(new_domain(vdom, K_NEE), nee_s, sobol_tab);
            uint32_t li = (uint32_t)(nee_s[0] * (float)n_lights);  // LightList::pick, light.rs:421-429
            if (li > n_lights - 1) li = n_lights - 1;
            const CrtLight &light = P.lights[li];
            LightSample ls;
            if (light_sample_li<INF>(light, rec.p, nee_s[1], nee_s[2], ls)) {
              s_shadow++;  // the reference traces the shadow ray before it evaluates the BSDF (tracer.rs:1412-1425)
              want_shadow = true;
              sh_d = ls.direction;
              sh_tmax = ls.distance - 0.001f;
              const float cosine = fabs_(dot(rec.normal, ls.direction));
              const float light_pdf = rmax(ls.pdf / (float)n_lights, 1e-6f);
              V3 brdf_value; float brdf_pdf;
              V3 nee = splat(0.0f);
              if (mat_eval<SIMPLE>(mat, rd, rec, ls.direction, brdf_value, brdf_pdf)) {
                const float weight = light_weight(P.strategy, light_pdf, brdf_pdf);
                V3 c = (ls.radiance * brdf_value) * cosine;
                c = c * splat(1.0f);  // shadow_tr == ONE when unoccluded
                nee = nee + (c * weight) / light_pdf;
              }
              sh_c = ba * nee;  // added to L by the shadow stage iff the segment is unoccluded
            }
          }
          L = L + ba * emit_here;

          // === 2. indirect lighting by BSDF sampling (tracer.rs:1459-1523) ===
          Scatter sample;
          if (mat_scatter<SIMPLE>(mat, rd, rec, new_domain(vdom, K_BSDF), sample, sobol_tab)) {
            const V3 dir = normalize(sample.dir);
            const float cosine = sample.delta ? 1.0f : fabs_(dot(rec.normal, dir));
            const V3 factor = (sample.value * cosine) / sample.pdf;
            beta = beta * (atten * factor);
            bool survived = true;
            if (n_rec >= (uint32_t)kRrStartBounce) {
              s_rr_t++;
              const float p_survive = rclamp(max_elem(beta), kRrMinProb, 1.0f);
              if (p_survive < 1.0f) {
                if (draw_rnd1(new_domain(vdom, K_RR)) >= p_survive) {
                  survived = false;
                  s_rr_k++;
                } else {
                  beta = beta / p_survive;
                }
              }
            }
            if (survived) {
              alive = true;
              n_o = sample.origin; n_d = sample.dir; n_ppdf = sample.pdf; n_delta = sample.delta;
              // materials build the ray: it carries the interior only when it refracts into the front face
              n_med = (MEDIA && sample.medium) ? P.media[mat_i].id : 0u;
            }
          }
          s_vertices++;
        }
      }
    }

    // ---- wave-level compaction: survivors go to the other state buffer, finished paths to the film ----
    uint32_t j = 0;
    {
      const uint32_t my_bin = dir_bin(n_d.x, n_d.y, n_d.z);
#pragma unroll
      for (int b = 0; b < kBins; b++) {
        const bool mine = alive && my_bin == (uint32_t)b;
        const uint32_t pos = seg_append(mine, &out_n[b]);
        if (mine) j = (blockIdx.x * kBins + b) * P.seg_cap + pos;
      }
    }
    const uint32_t sl = aux & 0xffffu;
    const uint32_t film_idx = sl * P.n_act + pix;
    if (alive) {
      st_nt(&N.a[j], make_float4(n_o.x, n_o.y, n_o.z, n_d.x));
      st_nt(&N.b[j], make_float4(n_d.y, n_d.z, beta.x, beta.y));
      st_nt(&N.c[j], make_float4(beta.z, L.x, L.y, L.z));
      st_nt(&N.d[j], make_uint4(pattern, pix, ((meta & 0xffffu) + 1u) | (((meta >> 16) - 1u) << 16),
                                sl | (n_prev_valid ? kPrevValid : 0u) | (n_delta ? kPrevDelta : 0u) | (n_med << kMediumShift)));
      if (n_lights) st_nt(&N.e[j], make_float4(hit_p.x, hit_p.y, hit_p.z, n_ppdf));
      if (P.has_motion) st_nt(&N.time[j], time);
    } else if (active) {
      st_nt(&staging[film_idx], make_float4(L.x, L.y, L.z, 0.0f));
    }
    const uint32_t q = seg0 + seg_append(want_shadow, &lds_ctr[1]);
    if (want_shadow) {
      st_nt(&Q.a[q], make_float4(hit_p.x, hit_p.y, hit_p.z, sh_tmax));
      st_nt(&Q.b[q], make_float4(sh_d.x, sh_d.y, sh_d.z, time));
      st_nt(&Q.c[q], make_float4(sh_c.x, sh_c.y, sh_c.z, __uint_as_float(alive ? j : (kFilmTarget | film_idx))));
    }
    __syncthreads();  // every lane has read its ring entry before CLASSIFY appends again
  }
  add_stat(&lds_ctr[2], s_closest); add_stat(&lds_ctr[3], s_shadow); add_stat(&lds_ctr[4], s_vertices);
  add_stat(&lds_ctr[5], s_rr_t); add_stat(&lds_ctr[6], s_rr_k); add_stat(&lds_ctr[7], s_esc);
  add_stat(&lds_ctr[8], s_depth);
  __syncthreads();
  if (threadIdx.x == 0) {  // publish this segment's queues for the next stages (same workgroup index there)
    C->shadow[blockIdx.x] = lds_ctr[1];
  }
  if (threadIdx.x < kBins) {
    C->seg[1 - cur][blockIdx.x * kBins + threadIdx.x] = out_n[threadIdx.x];
    C->seg[cur][blockIdx.x * kBins + threadIdx.x] = 0;  // consumed: this buffer is the output of the next round
  }
  if (threadIdx.x >= 1 && threadIdx.x <= 7 && lds_ctr[threadIdx.x + 1])
    atomicAdd(&C->stats[threadIdx.x], (unsigned long long)lds_ctr[threadIdx.x + 1]);
}
template <int MATS, bool INF, bool WIDE, bool LIT>
__global__ __launch_bounds__(kBlock, WIDE ? CRT_SHADE_WIDE_WAVES : CRT_SHADE_WAVES) void k_shade(Params P, PathSoA S, PathSoA N, HitSoA H, ShadowSoA Q,
                                                                   Counters *C, int cur, float4 *staging, int first) {
  static_assert(LIT || !INF, "lights at infinity are lights");
  constexpr int ARENA = WIDE ? kArenaWide : kArenaDwords;
  __shared__ uint32_t sobol_tab[ARENA];
  shade_segment<MATS, INF, LIT, ARENA>(P, S, N, H, Q, C, cur, staging, sobol_tab, (first & 1) != 0, (first & 2) != 0);
}

// ---- shadow: World::occluded (rt_world.rs:235-237) for the queue; unoccluded requests pay out ----
template <bool STATS, int WIDE>
__device__ __forceinline__ void shadow_segment(const Params &P, const PathSoA &N, const ShadowSoA &Q, Counters *C,
                                               float4 *staging, CrtTravStats *tstats, uint32_t *engine_lds) {
  const uint32_t n = ((const volatile uint32_t *)C->shadow)[blockIdx.x];
  if (n == 0) return;  // uniform per workgroup
  __shared__ uint32_t next;
  if (threadIdx.x == 0) next = 0;
  const uint32_t seg0 = blockIdx.x * P.seg_cap;
  LaneStats st = {};
  uint32_t err = 0, done = 0;
  auto fetch = [&](bool want, RayIn &in) -> bool {
    uint32_t k;
    if (!lds_take(want, &next, n, k)) return false;
    const uint32_t q = seg0 + k;
    const float4 A = Q.a[q], B = Q.b[q];
    in.ox = A.x; in.oy = A.y; in.oz = A.z; in.dx = B.x; in.dy = B.y; in.dz = B.z;
    in.time = B.w; in.mask = CRT_MASK_SHADOW; in.t_min = 0.001f; in.t_max = A.w; in.slot = q;
    return true;
  };
  auto emit = [&](uint32_t q, bool occ, const Hit &, float, float, float) {
    done++;
    if (occ) return;
    const float4 Cc = Q.c[q];
    const uint32_t tg = __float_as_uint(Cc.w);
    if (tg & kFilmTarget) {  // the path ended at this vertex: its radiance already sits in the staging plane
      const uint32_t f = tg & ~kFilmTarget;
      float4 v = staging[f];
      v.x = v.x + Cc.x; v.y = v.y + Cc.y; v.z = v.z + Cc.z;
      staging[f] = v;
    } else {  // beta.z | L.xyz
      float4 v = N.c[tg];
      v.y = v.y + Cc.x; v.z = v.z + Cc.y; v.w = v.w + Cc.z;
      N.c[tg] = v;
    }
  };
  run_traversal<true, STATS, WIDE, (int)kColdAll, WIDE != 0>(P.scene, engine_lds, 0.001f, err, st, fetch, emit, CRT_MASK_SHADOW);
  if (err) atomicOr(&C->err, err);
  if (STATS) flush_stats(st, tstats, done);
}
template <bool STATS, int WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? 4 : CRT_SHADOW_WAVES) void k_shadow(Params P, PathSoA N, ShadowSoA Q, Counters *C,
                                                                     float4 *staging, CrtTravStats *tstats) {
  __shared__ __attribute__((aligned(16))) uint32_t engine_lds[WIDE ? kEngineLdsWide : kEngineLdsDwords];
  shadow_segment<STATS, WIDE>(P, N, Q, C, staging, tstats, engine_lds);
}

// ---- the whole path loop of one wavefront batch in ONE launch. Queue segments are private to their workgroup at
// every stage (generate, extend, shade and shadow of segment b all run in workgroup b), so nothing but a workgroup
// barrier separates the stages: no grid-wide barrier, no launch per bounce, and workgroups drift apart freely — while
// one is shading (VALU-bound) its neighbours on the same SIMDs are traversing (latency-bound). A workgroup leaves as
// soon as its segment is empty. LIT: the scene has lights and the strategy samples them (shadow stage present). ----
// No instance for lights at infinity: those scenes (outside SURVEY §8) always run the per-stage launches. The one fault
// this code base ever showed — round 2, run-to-run differences of a few ulps — was confined to k_path<., LIT, INF> of
// one build, whose per-stage launches of the same shade code were exact; its cause was never established
// (profiles/README.md, "The round-2 nondeterminism"), so that kernel family is not built.
template <int MATS, bool LIT, int COLD>
__global__ __launch_bounds__(kBlock, CRT_EXTEND_WAVES) void k_path(Params P, PathSoA S0, PathSoA S1, HitSoA H, ShadowSoA Q, Counters *C,
                                                 float4 *staging, uint32_t sample_begin, uint32_t n_samples,
                                                 uint32_t start_it, int cur0) {
  __shared__ __attribute__((aligned(16))) uint32_t arena[kArenaDwords];
  __shared__ uint32_t live;
  // start_it > 0: the TAIL of a per-stage batch — the paths still alive after `start_it` bounces sit in buffer `cur0`
  // of their segments (written by the last k_shade launch); this launch runs what is left of the path loop
  if (start_it == 0) {
    generate_segment(P, S0, C, sample_begin, n_samples, arena, true);
    __syncthreads();
  }
  int cur = cur0;
  for (uint32_t it = start_it; it <= P.max_depth; it++) {
    const PathSoA &S = cur ? S1 : S0;
    const PathSoA &N = cur ? S0 : S1;
    extend_segment<false, false, COLD, false>(P, S, H, C, cur, it == 0 ? 1 : 0, nullptr, arena);
    __syncthreads();  // hit records of this segment are complete; the arena changes hands
    shade_segment<MATS, false, LIT, kArenaDwords>(P, S, N, H, Q, C, cur, staging, arena, false);
    __syncthreads();
    if (LIT) {
      shadow_segment<false, false>(P, N, Q, C, staging, nullptr, arena);
      __syncthreads();
    }
    cur = 1 - cur;
    if (threadIdx.x == 0) {
      uint32_t n = 0;
      for (int b = 0; b < kBins; b++) n += ((const volatile uint32_t *)C->seg[cur])[blockIdx.x * kBins + b];
      live = n;
    }
    __syncthreads();
    if (live == 0) break;  // uniform: every path of this segment has ended
  }
}

// ---- resolve: sum += color, in sample order; weight_sum += wx*wy = 1 (tracer.rs:599-600) ----
__global__ __launch_bounds__(kBlock) void k_resolve(const float4 *staging, uint32_t n_pix, uint32_t n_samples,
                                                    float4 *film /* r, g, b sums and weight_sum */) {
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n_pix; p += gridDim.x * kBlock) {
    float4 acc = film[p];
    for (uint32_t s = 0; s < n_samples; s++) {
      const float4 v = staging[(size_t)s * n_pix + p];
      acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z;
      acc.w = acc.w + 1.0f;
    }
    film[p] = acc;
  }
}

// ---- adaptive variant (tracer.rs:599-617): the same fold, plus the luminance statistics in f64 and the stopping
// rule evaluated after every 4th sample once min_spp are in — exactly where render_pixel evaluates it. A pixel that
// stops ignores the rest of the batch (those samples were traced but are not part of its estimate). ----
struct PixelStats { double lum_sum, lum_sq; };
__global__ __launch_bounds__(kBlock) void k_resolve_adaptive(const float4 *staging, const uint32_t *active, uint32_t n_act,
                                                             uint32_t n_samples, float4 *film, PixelStats *stats,
                                                             uint32_t *state /* taken | stopped << 31 */,
                                                             uint32_t min_spp, float variance_threshold) {
  const double threshold = (double)variance_threshold;
  for (uint32_t a = blockIdx.x * kBlock + threadIdx.x; a < n_act; a += gridDim.x * kBlock) {
    const uint32_t p = active ? active[a] : a;
    float4 acc = film[p];
    PixelStats ps = stats[p];
    uint32_t st = state[p];
    uint32_t taken = st & 0x7fffffffu;
    bool stopped = (st >> 31) != 0;
    for (uint32_t s = 0; s < n_samples && !stopped; s++) {
      const float4 v = staging[(size_t)s * n_act + a];
      acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z;
      acc.w = acc.w + 1.0f;
      const double lum = (double)(0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z);  // guiding/mod.rs:18-20
      ps.lum_sum += lum;
      ps.lum_sq += lum * lum;
      taken++;
      if (threshold > 0.0 && taken >= min_spp && taken % 4 == 0) {
        const double n = (double)taken;
        double var_of_mean = (ps.lum_sq - ps.lum_sum * ps.lum_sum / n) / (n - 1.0) / n;
        if (!(var_of_mean > 0.0)) var_of_mean = 0.0;
        double mean = ps.lum_sum / n;
        if (!(mean > 1e-4)) mean = 1e-4;
        if (sqrt(var_of_mean) / mean < threshold) stopped = true;
      }
    }
    film[p] = acc;
    stats[p] = ps;
    state[p] = taken | (stopped ? 0x80000000u : 0u);
  }
}
// The pixels still sampling, as a list of owned-pixel indices (order irrelevant: every pixel is independent).
__global__ __launch_bounds__(kBlock) void k_compact_active(const uint32_t *state, uint32_t n_pix, uint32_t *active,
                                                           uint32_t *count) {
  for (uint32_t base = blockIdx.x * kBlock; base < n_pix; base += gridDim.x * kBlock) {
    const uint32_t p = base + threadIdx.x;
    const bool on = p < n_pix && (state[p] >> 31) == 0;
    const unsigned long long m = __ballot(on);
    if (m == 0) continue;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(count, (uint32_t)__popcll(m));
    at = __shfl(at, leader, 64) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (on) active[at] = p;
  }
}

// pixel = sum / weight_sum (tracer.rs:630-634); rgb interleaved per owned pixel.
__global__ __launch_bounds__(kBlock) void k_film_out(const float4 *film, uint32_t n_pix, float *rgb) {
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n_pix; p += gridDim.x * kBlock) {
    const float4 v = film[p];
    // weight_sum > 0 always holds for box/triangle (w = taken); the fallback arm divides by `taken` = w too.
    rgb[3 * (size_t)p] = v.x / v.w; rgb[3 * (size_t)p + 1] = v.y / v.w; rgb[3 * (size_t)p + 2] = v.z / v.w;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Host side: Renderer (tracer.rs:137-148) over the kernels above
// ---------------------------------------------------------------------------------------------
struct Renderer {
  std::shared_ptr<Scene> scene;
  Params P{};
  std::vector<uint32_t> pixels;  // owned pixels (linear buffer index), tile order
  // One wavefront batch's buffers: path state (two buffers), hit records, shadow queue, staging film, queue counters.
  // LANES: a batch of n samples runs as up to kMaxLanes sub-batches of n / lanes consecutive samples, each with its own
  // buffers and counters on its own HIP stream, so the launches of one sub-batch overlap the other's — the ramp-up and
  // the drain of a launch (workgroups of its last round finishing one by one) are filled by the other lane's work, and a
  // traversal launch (latency and issue bound) runs beside a shade launch (the one that moves bytes). The film fold
  // stays on the caller's stream, lane after lane, so samples are summed in the same order as in one batch: bits unchanged.
  struct Lane {
    size_t cap_slots = 0;          // path slots (segment capacity x segments) the buffers are sized for
    char *blob = nullptr;
    size_t blob_bytes = 0;
    PathSoA S[2]{};
    HitSoA H{};
    ShadowSoA Q{};
    Counters *C = nullptr;
    float4 *staging = nullptr;
    hipStream_t stream = nullptr;  // lanes 1.. only (lane 0 runs on the caller's stream)
    hipEvent_t done = nullptr;
  };
  static constexpr int kMaxLanes = 4;
  Lane lanes[kMaxLanes];
  int n_lanes = 4;               // CRT_LANES; a batch is split only while every lane keeps lane_min_paths paths
  int last_lanes = 1;            // of the last batch (crt_renderer_lanes)
  size_t lane_min_paths = (size_t)96 << 20;
  hipEvent_t ev_start = nullptr;
  float4 *film = nullptr;
  CrtMaterial *d_materials = nullptr;
  uint8_t *d_mat_class = nullptr;  // material_class() per record of d_materials
  uint16_t *d_mat_index = nullptr; // geom_id -> record, when the table is deduplicated (Params::mat_index)
  DevMedium *d_media = nullptr;  // [n_materials] by geom_id, then [n_materials] by compact id
  bool has_media = false;
  int mats_kind = 1;  // 0 simple / 1 general / 2 general with interior media: which instance of the shading code runs
  // adaptive stopping (variance_threshold > 0): per-pixel luminance statistics, sample counts and the active list
  PixelStats *d_pstats = nullptr;
  uint32_t *d_state = nullptr, *d_active = nullptr, *d_count = nullptr;
  uint32_t n_act = 0, min_spp = 2;
  float variance_threshold = 0.0f;
  // Two pipelines: FUSED — the whole path loop of a batch in one launch (k_path, three workgroups per CU) — and PER-STAGE
  // with the WIDE traversal kernels (four workgroups per CU). The scene decides which one it prefers (crt_renderer_new:
  // `wide`), the batch whether the per-stage form pays: its 2-3 launches per bounce cost ~0.8 ms per batch, worth it from
  // `stage_min_paths` paths up. CRT_FUSED / CRT_WIDE / CRT_STAGE_MIN_PATHS override (A/B, per-stage timing, tests).
  EngineSelect engine;     // which traversal-engine instance runs this scene's image (crt_internal.h, select_engine)
  bool wide = false;       // = engine.wide: per-stage launches run the four-wave traversal kernels
  // The scene prefers one launch per stage for large batches — since round 4 every scene does: the lanes then overlap
  // launches of different stages and bounces, which one fused launch per lane cannot (measured first on the three-wave
  // kernels, 1080p / 4K, Mray/s fused -> per-stage: stress 2460 -> 2559, PointInstancedMedCity 2191 -> 2224, the
  // 7 M-triangle synthetic scene 3307 -> 3400; the per-stage traversal kernels are the four-wave instances select_engine
  // names — flat engine copy or, for direct-leaf images, the direct one; profiles/README.md).
  bool prefer_stage = false;
  bool cam_compact_ok = true;  // CRT_CAM_COMPACT
  int shade_wide = -1;            // CRT_SHADE_WIDE: 0 = the three-wave shade kernels even beside four-wave traversal kernels (A/B)
  size_t max_batch_slots = 0;     // CRT_MAX_BATCH_SLOTS (tests): ensure_buffers fails above this many slots; 0 = no limit
  int tail_from = 12;             // CRT_TAIL_FROM: the bounce from which a per-stage batch finishes in one fused launch
  int noclassify_from = 1 << 30;  // CRT_NOCLASSIFY_FROM: per-stage shade without its CLASSIFY pass from this bounce on
  int force_fused = -1;    // CRT_FUSED: -1 unset
  size_t stage_min_paths = (size_t)96 << 20;  // cornellbox 1080p, fused / per-stage Mray/s: 66 M paths 7507 / 7300, 133 M 7658 / 7900
  bool fused = true;       // what the LAST batch ran (crt_renderer_pipeline)
  int cus = 256, fused_mult = 3, stage_mult = 8, mult_forced = 0;
  CrtLight *d_lights = nullptr;
  uint32_t *d_pixels = nullptr;
  int grid = 768;          // of the last batch; the film kernels use it too
  hipStream_t last_stream = nullptr;
  bool profile = false;
  struct Ev { hipEvent_t a, b; int cls; };
  std::vector<Ev> events;
  double prof_ms[4] = {0, 0, 0, 0};
  uint64_t prof_launches[4] = {0, 0, 0, 0};

  ~Renderer() {
    drain_events();
    for (Lane &B : lanes) {
      if (B.stream) (void)hipStreamSynchronize(B.stream);  // nothing of a lane may be in flight when its buffers go
      if (B.blob) (void)hipFree(B.blob);
      if (B.C) (void)hipFree(B.C);
      if (B.stream) (void)hipStreamDestroy(B.stream);
      if (B.done) (void)hipEventDestroy(B.done);
    }
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (film) (void)hipFree(film);
    if (d_materials) (void)hipFree(d_materials);
    if (d_mat_class) (void)hipFree(d_mat_class);
    if (d_mat_index) (void)hipFree(d_mat_index);
    if (d_media) (void)hipFree(d_media);
    if (d_pstats) (void)hipFree(d_pstats);
    if (d_state) (void)hipFree(d_state);
    if (d_active) (void)hipFree(d_active);
    if (d_count) (void)hipFree(d_count);
    if (d_lights) (void)hipFree(d_lights);
    if (d_pixels) (void)hipFree(d_pixels);
  }

  void drain_events() {
    for (Ev &e : events) {
      float ms = 0.0f;
      if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
        prof_ms[e.cls] += ms;
        prof_launches[e.cls]++;
      }
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
    events.clear();
  }

  // Workgroups (= queue segments) of a batch of `total` paths. Fused: 3 per CU, what stays resident (LDS of the
  // traversal engine), so a wave's ray pool drains once per launch; per-stage: 8 — two rounds of the four resident
  // workgroups. Either is doubled while a segment would hold more than 192 Ki paths: past that, shorter segments balance
  // better than longer ones drain (per-stage, cornellbox 1080p, Mray/s: 128 spp x8 8179, x12 8011, x16 7490; 256 spp x8
  // 8420, x16 8630, x32 8655 — fused, 531 M paths: MedCity x3 1985, x6 1997, x12 2028, x24 2041; openpbr_showcase 12077,
  // 12142, 12179, 12067; 133 M paths: openpbr_showcase x3 11579, x6 11426, x12 11152). Only power-of-two multiples deal
  // the camera samples evenly (per-stage x6 7066, x10 7397, x12 8011, x20 8318).
  int batch_grid(size_t total, bool fused_pipeline) const {
    int mult = fused_pipeline ? fused_mult : stage_mult;
    if (!mult_forced)
      while (mult < 64 && total / ((size_t)cus * mult) > ((size_t)192 << 10)) mult *= 2;
    const int g = cus * mult;
    return g > kMaxGrid ? kMaxGrid : g;
  }

  int ensure_buffers(Lane &B, size_t slots) {  // slots = segment capacity x segments of the batch about to run
    size_t &cap_slots = B.cap_slots, &blob_bytes = B.blob_bytes;
    char *&blob = B.blob;
    PathSoA (&S)[2] = B.S;
    HitSoA &H = B.H;
    ShadowSoA &Q = B.Q;
    float4 *&staging = B.staging;
    // grow only — except that a lane of a multi-lane batch gives back a buffer sized for a whole batch (a single-lane
    // call before it, e.g. the stats pass): four lanes next to one whole-batch buffer would not fit the HBM
    if (slots <= cap_slots && blob && !(last_lanes > 1 && cap_slots > slots + slots / 2)) return CRT_OK;
    if (blob) { (void)hipFree(blob); blob = nullptr; }
    const size_t total = slots, cap = slots;  // shadow queue and staging film: one slot per path
    const size_t bcap = cap * kBins;      // path and hit planes: one sub-segment per (workgroup, direction bin)
    if (total == 0 || bcap >= (size_t)0x7fffffff) return CRT_ERR_BAD_ARG;
    const size_t p16 = (cap * 16 + 255) & ~size_t(255);
    const size_t b16 = (bcap * 16 + 255) & ~size_t(255), b4 = (bcap * 4 + 255) & ~size_t(255);
    const bool lit = P.n_lights > 0, motion = P.has_motion != 0;
    // 2 x (a b c d [e] [16 B] + [time 4 B]) + hit (16 + 4) + shadow 3 x 16 + staging 16
    blob_bytes = 2 * ((lit ? 5 : 4) * b16 + (motion ? b4 : 0)) + (b16 + b4) + (lit ? 3 * p16 : 0) + p16;
    // CRT_MAX_BATCH_SLOTS (tests): a batch of more slots asks for more memory than any device has, so the allocation
    // fails the way it does on a part with too little free HBM (bench.py halves the batch then).
    const size_t want_bytes = (max_batch_slots && slots > max_batch_slots) ? ((size_t)1 << 46) : blob_bytes;
    if (!CRT_HIP_OK(hipMalloc(&blob, want_bytes))) {
      // hipGetLastError returns the last ERROR, not the last call's status: left in place, the out-of-memory would be
      // reported by the next successful batch's launch check (a retry with a smaller batch would fail for no reason)
      (void)hipGetLastError();
      blob = nullptr; blob_bytes = 0; cap_slots = 0;
      return CRT_ERR_NO_DEVICE;
    }
    char *cur_p = blob;
    auto take = [&](size_t bytes) { char *r = cur_p; cur_p += bytes; return r; };
    for (int b = 0; b < 2; b++) {
      S[b].a = (float4 *)take(b16); S[b].b = (float4 *)take(b16); S[b].c = (float4 *)take(b16);
      S[b].d = (uint4 *)take(b16);
      S[b].e = lit ? (float4 *)take(b16) : nullptr;
      S[b].time = motion ? (float *)take(b4) : nullptr;
    }
    H.h = (float4 *)take(b16); H.geom = (uint32_t *)take(b4);
    if (lit) { Q.a = (float4 *)take(p16); Q.b = (float4 *)take(p16); Q.c = (float4 *)take(p16); }
    staging = (float4 *)take(p16);
    cap_slots = cap;
    return CRT_OK;
  }

  template <class F>
  void timed(int cls, hipStream_t st, F &&launch) {
    if (!profile) { launch(); return; }
    Ev e; e.cls = cls;
    (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
    (void)hipEventRecord(e.a, st);
    launch();
    (void)hipEventRecord(e.b, st);
    events.push_back(e);
  }

  int render(uint32_t sample_begin, uint32_t n_samples, hipStream_t st, CrtTravStats *d_tstats) {
    if (n_samples == 0) return CRT_OK;
    if (n_samples > 0xffffu) return CRT_ERR_BAD_ARG;
    (void)hipGetLastError();  // launches below are checked against a clean slate, not against an earlier call's error
    last_stream = st;
    // how many lanes: one for the stats build and adaptive stopping (the active list changes between batches) and for
    // batches too small to fill the chip twice over
    int L = n_lanes;
    const size_t total = (size_t)P.n_pix * n_samples;
    if (d_tstats || variance_threshold > 0.0f) L = 1;
    while (L > 1 && (total / L < lane_min_paths || n_samples < (uint32_t)L)) L--;
    last_lanes = L < 1 ? 1 : L;
    // lanes this batch does not use give their buffers back first: a whole batch in lane 0 next to the quarter batches of
    // an earlier call's other lanes would not fit (the stats pass behind a four-lane batch of a lit scene: 259 + 195 GB)
    for (int l = L < 1 ? 1 : L; l < kMaxLanes; l++) {
      Lane &B = lanes[l];
      if (!B.blob) continue;
      if (B.stream) (void)hipStreamSynchronize(B.stream);
      (void)hipFree(B.blob);
      B.blob = nullptr; B.cap_slots = 0; B.blob_bytes = 0;
    }
    if (L <= 1) return render_lane(lanes[0], sample_begin, n_samples, st, d_tstats, true);
    if (!ev_start && !CRT_HIP_OK(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming))) return CRT_ERR_NO_DEVICE;
    for (int l = 1; l < L; l++) {
      Lane &B = lanes[l];
      if (!B.stream && !CRT_HIP_OK(hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking))) return CRT_ERR_NO_DEVICE;
      if (!B.done && !CRT_HIP_OK(hipEventCreateWithFlags(&B.done, hipEventDisableTiming))) return CRT_ERR_NO_DEVICE;
    }
    uint32_t begin[kMaxLanes], count[kMaxLanes];
    for (int l = 0; l < L; l++) {
      count[l] = n_samples / L + ((uint32_t)l < n_samples % L ? 1u : 0u);
      begin[l] = l == 0 ? sample_begin : begin[l - 1] + count[l - 1];
    }
    // every lane's buffers BEFORE any launch: an allocation that fails then fails with nothing of this batch in flight
    for (int l = 0; l < L; l++) {
      Params scratch = P;
      const int rc = plan_lane(lanes[l], scratch, count[l], nullptr);
      if (rc != CRT_OK) return rc;
    }
    // every lane starts behind what the caller's stream has queued (the previous batch's film fold reads the staging
    // films this batch overwrites); lane 0 IS the caller's stream
    if (!CRT_HIP_OK(hipEventRecord(ev_start, st))) return CRT_ERR_NO_DEVICE;
    // a failure from here on leaves launches in flight on the lanes' own (non-blocking) streams, which nothing the caller
    // does afterwards waits for (crt_render_stats, crt_film_clear, a retry sync only the caller's stream): drain them
    auto fail = [&](int rc) {
      for (int l = 1; l < L; l++)
        if (lanes[l].stream) (void)hipStreamSynchronize(lanes[l].stream);
      (void)hipStreamSynchronize(st);
      return rc;
    };
    // the lanes on their own streams first, lane 0 on the caller's last: its launches then queue behind nothing of ours
    for (int l = L - 1; l >= 0; l--) {
      hipStream_t ls = l == 0 ? st : lanes[l].stream;
      if (l > 0 && !CRT_HIP_OK(hipStreamWaitEvent(ls, ev_start, 0))) return fail(CRT_ERR_NO_DEVICE);
      const int rc = render_lane(lanes[l], begin[l], count[l], ls, nullptr, false);
      if (rc != CRT_OK) return fail(rc);
      if (l > 0 && !CRT_HIP_OK(hipEventRecord(lanes[l].done, ls))) return fail(CRT_ERR_NO_DEVICE);
    }
    // the film fold: lane after lane on the caller's stream = sample order
    for (int l = 0; l < L; l++) {
      if (l > 0 && !CRT_HIP_OK(hipStreamWaitEvent(st, lanes[l].done, 0))) return fail(CRT_ERR_NO_DEVICE);
      timed(3, st, [&] { hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(kBlock), 0, st, lanes[l].staging, P.n_pix, count[l], film); });
    }
    return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : fail(CRT_ERR_NO_DEVICE);
  }

  // Pipeline, grid and segment size of one lane's batch of n_samples, and its buffers (they grow when a batch needs
  // more slots than any before). Sets fused / grid for the launches that follow and p.seg_cap.
  int plan_lane(Lane &B, Params &p, uint32_t n_samples, CrtTravStats *d_tstats) {
    const size_t total = (size_t)p.n_act * n_samples;
    fused = force_fused >= 0 ? force_fused != 0 : !(prefer_stage && total >= stage_min_paths);
    if (d_tstats) fused = false;  // the stats build is the per-stage one
    if (P.has_inf_lights) fused = false;  // no fused instance for lights at infinity (see k_path)
    grid = batch_grid(total, fused);
    p.seg_cap = (uint32_t)(((total + (size_t)grid * kBlock - 1) / ((size_t)grid * kBlock)) * kBlock);
    return ensure_buffers(B, (size_t)p.seg_cap * grid);  // >= total: the staging film's (sample, active pixel) slots too
  }

  int render_lane(Lane &B, uint32_t sample_begin, uint32_t n_samples, hipStream_t st, CrtTravStats *d_tstats, bool fold_here) {
    PathSoA (&S)[2] = B.S;
    HitSoA &H = B.H;
    ShadowSoA &Q = B.Q;
    Counters *C = B.C;
    const bool adaptive = variance_threshold > 0.0f;
    if (adaptive && n_act == 0) return CRT_OK;  // every pixel has stopped
    Params p = P;
    p.sample_begin = sample_begin;
    p.n_act = adaptive ? n_act : P.n_pix;
    p.active = (adaptive && n_act < P.n_pix) ? d_active : nullptr;
    if (const int rc = plan_lane(B, p, n_samples, d_tstats)) return rc;
    float4 *staging = B.staging;
    const bool wide = !fused && this->wide;  // per-stage launches take the scene's preferred traversal kernels
    const bool wdirect = wide && engine.wide_direct && CRT_WIDE_DIRECT_BUILD != 0;  // ... their direct-engine instances
    // camera paths as 16-byte records: per-stage launches of an UNLIT scene, pinhole camera, static scene
    // (CRT_CAM_COMPACT=0: A/B, tests)
    p.cam_compact = (CRT_CAM_COMPACT_BUILD && !fused && cam_compact_ok && P.n_lights == 0 && !(P.camera.lens_radius > 0.0f) && !P.has_motion) ? 1u : 0u;
    // the film fold: plain sum, or with the luminance statistics and the stopping rule, then the new active list
    auto fold = [&]() -> int {
      if (!fold_here) return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : CRT_ERR_NO_DEVICE;  // the caller folds the lanes in order
      if (!adaptive) {
        timed(3, st, [&] { hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(kBlock), 0, st, staging, P.n_pix, n_samples, film); });
        return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : CRT_ERR_NO_DEVICE;
      }
      timed(3, st, [&] {
        hipLaunchKernelGGL(k_resolve_adaptive, dim3(grid), dim3(kBlock), 0, st, staging, p.active, p.n_act, n_samples, film,
                           d_pstats, d_state, min_spp, variance_threshold);
        (void)hipMemsetAsync(d_count, 0, 4, st);
        hipLaunchKernelGGL(k_compact_active, dim3(grid), dim3(kBlock), 0, st, d_state, P.n_pix, d_active, d_count);
      });
      uint32_t h = 0;  // the next batch's size is a launch parameter: one 4-byte read-back per batch
      if (!CRT_HIP_OK(hipMemcpyAsync(&h, d_count, 4, hipMemcpyDeviceToHost, st)) || !CRT_HIP_OK(hipStreamSynchronize(st)))
        return CRT_ERR_NO_DEVICE;
      n_act = h;
      return CRT_OK;
    };
    const bool lit = P.n_lights > 0;  // the kernel instance; whether the strategy samples the lights is checked in shade
    // the cold per-ray state the scene can need (DevScene::cold) picks the closest-hit kernels' instance: none / the
    // pending normal only / everything for the per-stage k_extend, none / everything for the fused kernel of simple
    // scenes — decided by select_engine, with the image in hand; a launch the image cannot take is refused, never made
    const int ext_cold = d_tstats ? (int)kColdAll : engine.ext_cold;
    // (general material tables run the full-cold fused kernel — or its packet-free instance, for images without a Tri4 packet)
    const bool nopk = mats_kind != 0 && (engine.path_cold & (int)kNoPackets) != 0;
    const int path_cold = mats_kind == 0 ? (engine.path_cold & (int)kColdAll) : (int)(kColdAll | (nopk ? kNoPackets : 0u));
    {
      EngineSelect launched = engine;
      launched.wide = wide;  // the fused kernel is a three-wave kernel whatever the scene prefers
      launched.wide_direct = wide && engine.wide_direct;
      launched.direct = (!wide || launched.wide_direct) && CRT_DIRECT_LEAVES != 0 && P.scene.direct_leaves != 0;  // run_traversal picks the copy from the image
      const bool tail = !fused && tail_from > 0 && !d_tstats && !P.has_inf_lights;
      if (!engine_accepts(launched, P.scene, fused ? path_cold : (tail ? (ext_cold & path_cold) : ext_cold))) {
        set_error_text("render refused: the selected kernels (wide %d, cold %d / %d) cannot run this image (direct words %u, cold %u)",
                       (int)wide, ext_cold, path_cold, P.scene.direct_leaves, P.scene.cold);
        return CRT_ERR_UNSUPPORTED;
      }
    }
    if (fused) {  // one launch for the whole path loop (class 0 of the profile), then the film fold
      timed(0, st, [&] {
#define CRT_PATH(M, L, CO) \
  hipLaunchKernelGGL((k_path<M, L, CO>), dim3(grid), dim3(kBlock), 0, st, p, S[0], S[1], H, Q, C, staging, sample_begin, n_samples, 0u, 0)
#if CRT_NOPK_BUILD  // the packet-free instances exist only in builds that ask for them (A/B: profiles/README.md, round 4)
#define CRT_PATH_NP(M, L) do { if (nopk) CRT_PATH(M, L, kColdAll | kNoPackets); else CRT_PATH(M, L, kColdAll); } while (0)
#else
#define CRT_PATH_NP(M, L) CRT_PATH(M, L, kColdAll)
#endif
        switch (mats_kind * 2 + (lit ? 1 : 0)) {
          case 0: if (path_cold == 0) CRT_PATH(0, false, 0); else CRT_PATH(0, false, kColdAll); break;
          case 1: if (path_cold == 0) CRT_PATH(0, true, 0); else CRT_PATH(0, true, kColdAll); break;
          case 2: CRT_PATH_NP(1, false); break;
          case 3: CRT_PATH_NP(1, true); break;
          case 4: CRT_PATH_NP(2, false); break;
          default: CRT_PATH_NP(2, true); break;
        }
#undef CRT_PATH_NP
#undef CRT_PATH
      });
      return fold();
    }
    timed(3, st, [&] { hipLaunchKernelGGL(k_generate, dim3(grid), dim3(kBlock), 0, st, p, S[0], C, sample_begin, n_samples); });
    int cur = 0;
    // The TAIL: from bounce `tail_from` on, what is left of the batch — roulette has ended all but a few paths per ten
    // thousand by then (bench: 6.6 M of 531 M rays at bounce 4, 0.1 M at bounce 6) — runs as ONE launch of the fused
    // path-loop kernel over the same segments instead of two or three launches per bounce up to the depth limit (bench,
    // depth 32: 52 launches, ~2 ms of a 137 ms step). Not for the stats build (its kernels count) nor for lights at
    // infinity (no fused instance). CRT_TAIL_FROM=n moves it (0: never).
    const uint32_t tail_at = (tail_from > 0 && !d_tstats && !P.has_inf_lights) ? (uint32_t)tail_from : 0xffffffffu;
    for (uint32_t it = 0; it <= P.max_depth; it++) {
      if (it >= tail_at) {
        timed(3, st, [&] {
#define CRT_TAIL(M, L, CO) \
  hipLaunchKernelGGL((k_path<M, L, CO>), dim3(grid), dim3(kBlock), 0, st, p, S[0], S[1], H, Q, C, staging, sample_begin, n_samples, it, cur)
#if CRT_NOPK_BUILD
#define CRT_TAIL_NP(M, L) do { if (nopk) CRT_TAIL(M, L, kColdAll | kNoPackets); else CRT_TAIL(M, L, kColdAll); } while (0)
#else
#define CRT_TAIL_NP(M, L) CRT_TAIL(M, L, kColdAll)
#endif
          switch (mats_kind * 2 + (lit ? 1 : 0)) {
            case 0: if (path_cold == 0) CRT_TAIL(0, false, 0); else CRT_TAIL(0, false, kColdAll); break;
            case 1: if (path_cold == 0) CRT_TAIL(0, true, 0); else CRT_TAIL(0, true, kColdAll); break;
            case 2: CRT_TAIL_NP(1, false); break;
            case 3: CRT_TAIL_NP(1, true); break;
            case 4: CRT_TAIL_NP(2, false); break;
            default: CRT_TAIL_NP(2, true); break;
          }
#undef CRT_TAIL_NP
#undef CRT_TAIL
        });
        break;
      }
#define CRT_EXTEND(ST, W, CO) \
  timed(0, st, [&] { hipLaunchKernelGGL((k_extend<ST, W, CO>), dim3(grid), dim3(kBlock), 0, st, p, S[cur], H, C, cur, it == 0 ? 1 : 0, d_tstats); })
      // (the stats build of a direct-leaf image counts on the three-wave kernels: the counters do not depend on the engine split)
      if (d_tstats) { if (wide && !wdirect) CRT_EXTEND(true, 1, kColdAll); else CRT_EXTEND(true, 0, kColdAll); }
#if CRT_WIDE_DIRECT_BUILD
      else if (wdirect) { if (ext_cold == 0) CRT_EXTEND(false, 2, 0); else if (ext_cold == (int)kColdNormal) CRT_EXTEND(false, 2, kColdNormal); else CRT_EXTEND(false, 2, kColdAll); }
#endif
      else if (wide) { if (ext_cold == 0) CRT_EXTEND(false, 1, 0); else if (ext_cold == (int)kColdNormal) CRT_EXTEND(false, 1, kColdNormal); else CRT_EXTEND(false, 1, kColdAll); }
      else { if (ext_cold == 0) CRT_EXTEND(false, 0, 0); else if (ext_cold == (int)kColdNormal) CRT_EXTEND(false, 0, kColdNormal); else CRT_EXTEND(false, 0, kColdAll); }
#undef CRT_EXTEND
#define CRT_SHADE(M, I, W, L) \
  timed(1, st, [&] { hipLaunchKernelGGL((k_shade<M, I, W, L>), dim3(grid), dim3(kBlock), 0, st, p, S[cur], S[1 - cur], H, Q, C, cur, staging, (it == 0 ? 1 : 0) | ((int)it >= noclassify_from ? 2 : 0)); })
      // the instance: material table (MATS), lights at infinity (INF), four waves (simple materials without lights at
      // infinity, when the scene runs the wide kernels), and — as for k_path — whether the light list is empty
      if (mats_kind == 2) { if (P.has_inf_lights) CRT_SHADE(2, true, false, true); else if (lit) CRT_SHADE(2, false, false, true); else CRT_SHADE(2, false, false, false); }
      else if (mats_kind == 1) { if (P.has_inf_lights) CRT_SHADE(1, true, false, true); else if (lit) CRT_SHADE(1, false, false, true); else CRT_SHADE(1, false, false, false); }
      else if (P.has_inf_lights) CRT_SHADE(0, true, false, true);
      else if (wide && !P.mat_index && shade_wide != 0) { if (lit) CRT_SHADE(0, false, true, true); else CRT_SHADE(0, false, true, false); }
      else { if (lit) CRT_SHADE(0, false, false, true); else CRT_SHADE(0, false, false, false); }
#undef CRT_SHADE
      if (P.n_lights > 0 && P.strategy != CRT_STRATEGY_BSDF) {
#define CRT_SHADOW(ST, W) \
  timed(2, st, [&] { hipLaunchKernelGGL((k_shadow<ST, W>), dim3(grid), dim3(kBlock), 0, st, p, S[1 - cur], Q, C, staging, d_tstats ? d_tstats + 1 : nullptr); })
        if (d_tstats) { if (wide && !wdirect) CRT_SHADOW(true, 1); else CRT_SHADOW(true, 0); }
#if CRT_WIDE_DIRECT_BUILD
        else if (wdirect) CRT_SHADOW(false, 2);
#endif
        else { if (wide) CRT_SHADOW(false, 1); else CRT_SHADOW(false, 0); }
#undef CRT_SHADOW
      }
      cur = 1 - cur;
    }
    return fold();
  }
};

}  // namespace crt

struct CrtRenderer { crt::Renderer r; };

using namespace crt;

extern "C" {

void crt_camera_new(CrtCamera *c, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov_deg,
                    float aspect, float aperture, float focus_dist) {  // camera.rs:27-63 (host-side setup)
  if (!c) return;
  auto sub = [](const float a[3], const float b[3], float o[3]) { for (int i = 0; i < 3; i++) o[i] = a[i] - b[i]; };
  auto nrm = [](float a[3]) {
    const float l = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    for (int i = 0; i < 3; i++) a[i] = a[i] / l;
  };
  auto crs = [](const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - b[1] * a[2]; o[1] = a[2] * b[0] - b[2] * a[0]; o[2] = a[0] * b[1] - b[0] * a[1];
  };
  const float theta = vfov_deg * 3.14159265358979323846f / 180.0f;
  const float h = tanf(theta / 2.0f);
  const float viewport_height = 2.0f * h;
  const float viewport_width = aspect * viewport_height;
  float w[3], u[3], v[3];
  sub(lookfrom, lookat, w); nrm(w);
  crs(vup, w, u); nrm(u);
  crs(w, u, v);
  for (int i = 0; i < 3; i++) {
    c->origin[i] = lookfrom[i];
    c->horizontal[i] = u[i] * (focus_dist * viewport_width);
    c->vertical[i] = v[i] * (focus_dist * viewport_height);
    c->lower_left[i] = ((lookfrom[i] - c->horizontal[i] / 2.0f) - c->vertical[i] / 2.0f) - w[i] * focus_dist;
    c->u[i] = u[i];
    c->v[i] = v[i];
  }
  c->lens_radius = aperture / 2.0f;
}

// The body of crt_renderer_new; it may throw (std containers of the pixel list, the material table's deduplication): the
// entry point below catches, `hold` frees what was built.
static CrtRenderer *renderer_new(CrtScene *scene, const CrtMaterial *materials, size_t n_materials, const CrtLight *lights,
                                 size_t n_lights, const CrtCamera *camera, const CrtRenderSettings *settings,
                                 uint32_t tile_rank, uint32_t tile_world) {
  if (!scene || !camera || !settings || (n_materials && !materials) || (n_lights && !lights)) return nullptr;
  if (tile_world == 0 || tile_rank >= tile_world) return nullptr;
  if (!(settings->variance_threshold >= 0.0f)) return nullptr;
  if (n_materials < scene->p->n_geoms) return nullptr;        // one material per geom_id (rt_world.rs:111-122)
  if (settings->max_depth > 0xffffu) return nullptr;
  if (scene->p->ensure_device() != CRT_OK) return nullptr;
  CrtRenderer *R = new (std::nothrow) CrtRenderer();
  if (!R) return nullptr;
  struct Hold { CrtRenderer *p; ~Hold() { delete p; } } hold{R};  // every way out but the last frees the renderer
  const Knobs knobs = read_knobs();
  Renderer &r = R->r;
  r.scene = scene->p;
  Params &P = r.P;
  P.scene = scene->p->dev->view;
  P.n_lights = (uint32_t)n_lights;
  P.n_materials = (uint32_t)n_materials;
  P.has_inf_lights = 0;
  for (size_t k = 0; k < n_lights; k++) {
    if (lights[k].kind > CRT_LIGHT_DOME) return nullptr;
    if (lights[k].kind >= CRT_LIGHT_DISTANT) P.has_inf_lights = 1;
  }
  P.camera = *camera;
  P.width = settings->width; P.height = settings->height; P.max_depth = settings->max_depth;
  P.frame = settings->frame; P.strategy = settings->strategy; P.filter_kind = settings->filter_kind;
  P.filter_radius = settings->filter_radius;
  P.has_motion = scene->p->has_motion ? 1u : 0u;
  // Pixel-tile shard: 16x16 tiles (tracer.rs:424, :1671-1686) dealt round-robin over the ranks.
  r.pixels.resize(crt_shard_pixels(P.width, P.height, tile_rank, tile_world, nullptr));
  crt_shard_pixels(P.width, P.height, tile_rank, tile_world, r.pixels.data());
  P.n_pix = (uint32_t)r.pixels.size();
  P.n_act = P.n_pix; P.active = nullptr;
  r.variance_threshold = settings->variance_threshold;
  r.min_spp = settings->min_spp > 2 ? settings->min_spp : 2;  // tracer.rs:526
  r.n_act = P.n_pix;
  bool ok = P.n_pix > 0;
  if (ok && r.variance_threshold > 0.0f) {
    ok = CRT_HIP_OK(hipMalloc(&r.d_pstats, (size_t)P.n_pix * sizeof(PixelStats))) &&
         CRT_HIP_OK(hipMemset(r.d_pstats, 0, (size_t)P.n_pix * sizeof(PixelStats))) &&
         CRT_HIP_OK(hipMalloc(&r.d_state, (size_t)P.n_pix * 4)) && CRT_HIP_OK(hipMemset(r.d_state, 0, (size_t)P.n_pix * 4)) &&
         CRT_HIP_OK(hipMalloc(&r.d_active, (size_t)P.n_pix * 4)) && CRT_HIP_OK(hipMalloc(&r.d_count, 4));
  }
  for (Renderer::Lane &B : r.lanes)
    ok = ok && CRT_HIP_OK(hipMalloc(&B.C, sizeof(Counters))) && CRT_HIP_OK(hipMemset(B.C, 0, sizeof(Counters)));
  ok = ok && CRT_HIP_OK(hipMalloc(&r.film, (size_t)P.n_pix * 16)) && CRT_HIP_OK(hipMemset(r.film, 0, (size_t)P.n_pix * 16));
  ok = ok && CRT_HIP_OK(hipMalloc(&r.d_pixels, (size_t)P.n_pix * 4)) &&
       CRT_HIP_OK(hipMemcpy(r.d_pixels, r.pixels.data(), (size_t)P.n_pix * 4, hipMemcpyHostToDevice));
  // The material table: one record per geometry id, or — when that table would not fit shade's LDS arena and holds
  // repeats — the distinct records and a 2-byte index per geometry id (Params::mat_index). Everything below that is
  // "per material" (class byte, interior medium) is then per DISTINCT material. CRT_MAT_DEDUP=0: never (A/B, tests).
  std::vector<CrtMaterial> unique_mats;
  std::vector<uint16_t> mat_index;
  {
    bool dedup = CRT_MAT_INDEX_BUILD && n_materials > (size_t)mat_lds_max(kArenaDwords);
    dedup = dedup && knobs.mat_dedup != 0;  // CRT_MAT_DEDUP=0
    if (dedup) {
      auto bytes_of = [](const CrtMaterial &m) { return std::string(reinterpret_cast<const char *>(&m), sizeof(CrtMaterial)); };
      std::unordered_map<std::string, uint32_t> seen;
      mat_index.resize(n_materials);
      for (size_t k = 0; k < n_materials && dedup; k++) {
        auto it = seen.emplace(bytes_of(materials[k]), (uint32_t)unique_mats.size());
        if (it.second) unique_mats.push_back(materials[k]);
        if (unique_mats.size() > 0xffffu) dedup = false;  // a 2-byte index
        mat_index[k] = (uint16_t)it.first->second;
      }
      if (!dedup || unique_mats.size() == n_materials) { unique_mats.clear(); mat_index.clear(); }
    }
  }
  if (!unique_mats.empty()) {
    ok = ok && CRT_HIP_OK(hipMalloc(&r.d_mat_index, n_materials * sizeof(uint16_t))) &&
         CRT_HIP_OK(hipMemcpy(r.d_mat_index, mat_index.data(), n_materials * sizeof(uint16_t), hipMemcpyHostToDevice));
    materials = unique_mats.data();
    n_materials = unique_mats.size();
    P.n_materials = (uint32_t)n_materials;
  }
  P.mat_index = r.d_mat_index;
  if (ok && n_materials) {
    ok = CRT_HIP_OK(hipMalloc(&r.d_materials, n_materials * sizeof(CrtMaterial))) &&
         CRT_HIP_OK(hipMemcpy(r.d_materials, materials, n_materials * sizeof(CrtMaterial), hipMemcpyHostToDevice));
  }
  if (ok && n_materials) {  // shade's partition key: one class byte per geom_id
    std::vector<uint8_t> cls(n_materials);
    bool seen[kClasses] = {false, false, false, false};
    for (size_t k = 0; k < n_materials; k++) { cls[k] = material_class(materials[k]); seen[cls[k]] = true; }
    int distinct = 0;
    for (int c = 0; c < kClasses; c++) distinct += seen[c] ? 1 : 0;
    P.partition = distinct > 1 ? 1u : 0u;
    if (knobs.partition >= 0) P.partition = knobs.partition ? 1u : 0u;  // CRT_PARTITION (A/B runs)
    ok = CRT_HIP_OK(hipMalloc(&r.d_mat_class, n_materials)) &&
         CRT_HIP_OK(hipMemcpy(r.d_mat_class, cls.data(), n_materials, hipMemcpyHostToDevice));
  }
  if (ok && n_lights) {
    ok = CRT_HIP_OK(hipMalloc(&r.d_lights, n_lights * sizeof(CrtLight))) &&
         CRT_HIP_OK(hipMemcpy(r.d_lights, lights, n_lights * sizeof(CrtLight), hipMemcpyHostToDevice));
  }
  if (ok && n_materials) {  // interior media of the materials, derived on the device
    uint32_t *d_count = nullptr, h_count = 0;
    ok = CRT_HIP_OK(hipMalloc(&r.d_media, 2 * n_materials * sizeof(DevMedium))) && CRT_HIP_OK(hipMalloc(&d_count, 4));
    if (ok) {
      hipLaunchKernelGGL(k_build_media, dim3((unsigned)((n_materials + 255) / 256)), dim3(256), 0, nullptr, r.d_materials,
                         (uint32_t)n_materials, r.d_media);
      hipLaunchKernelGGL(k_number_media, dim3(1), dim3(kNumberBlock), 0, nullptr, (uint32_t)n_materials, r.d_media,
                         r.d_media + n_materials, d_count);
      ok = CRT_HIP_OK(hipGetLastError()) && CRT_HIP_OK(hipMemcpy(&h_count, d_count, 4, hipMemcpyDeviceToHost));
      ok = ok && h_count <= kMaxMedia;  // the path state carries a 14-bit medium id
      r.has_media = h_count > 0;
    }
    if (d_count) (void)hipFree(d_count);
  }
  if (!ok) return nullptr;
  {
    bool simple = true;
    for (size_t k = 0; k < n_materials; k++) simple = simple && material_class(materials[k]) <= 1;
    simple = simple && knobs.simple != 0;  // CRT_SIMPLE=0: the general instance (A/B, tests)
    r.mats_kind = r.has_media ? 2 : (simple ? 0 : 1);
  }
  P.materials = r.d_materials; P.lights = r.d_lights; P.pixel_index = r.d_pixels;
  P.mat_class = r.d_mat_class;
  P.media = r.d_media; P.media_by_id = r.d_media ? r.d_media + n_materials : nullptr;
  hipDeviceProp_t prop;
  int dev = 0;
  (void)hipGetDevice(&dev);
  // Pipeline by scene (see Renderer::fused; crt_internal.h, wide_split): small flat triangle scenes gain 4-9 % from
  // the fourth wave per SIMD of the per-stage kernels; an instanced city loses 6 % to their four-entry LDS stack, and a
  // scene of analytic spheres only (openpbr_showcase: next to no traversal, all shading) 1 % to the hit records' round
  // trip (profiles/README.md).
  // select_engine decides, from the image, which engine instance runs it; CRT_WIDE=0/1 (A/B, tests) is a request it
  // honours only where the image can be decoded by what was asked for (never the four-wave kernels on direct words)
  if (select_engine_env(P.scene, r.engine, true) != CRT_OK) {
    set_error_text("crt_renderer_new: no traversal-engine instance of this build can decode the scene image");
    return nullptr;
  }
  r.wide = r.engine.wide;
  // ... and, once lanes overlap a batch's launches, every scene: the sphere-only showcase 12 176 -> 13 078 Mray/s (+7.4 %;
  // round 2, one stream: -1 %). The fused kernel remains what small batches and the tails of large ones run.
  r.prefer_stage = true;
  if (knobs.prefer_stage >= 0) r.prefer_stage = knobs.prefer_stage != 0;  // the A/B knobs (crt_internal.h, Knobs): CRT_PREFER_STAGE ...
  r.cam_compact_ok = knobs.cam_compact != 0;
  r.noclassify_from = knobs.noclassify_from;
  r.tail_from = knobs.tail_from;
  r.max_batch_slots = knobs.max_batch_slots;
  r.n_lanes = knobs.lanes < 1 ? 1 : (knobs.lanes > Renderer::kMaxLanes ? Renderer::kMaxLanes : knobs.lanes);
  r.lane_min_paths = knobs.lane_min_paths;
  r.shade_wide = knobs.shade_wide;
  r.force_fused = knobs.fused;
  r.stage_min_paths = knobs.stage_min_paths;
  // Workgroups per CU = queue segments per CU: Renderer::batch_grid.
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) r.cus = prop.multiProcessorCount;
  r.stage_mult = r.wide ? 8 : 3;
  if (knobs.grid_mult > 0) { r.fused_mult = r.stage_mult = knobs.grid_mult; r.mult_forced = 1; }  // CRT_GRID_MULT: tuning knob
  r.fused = r.force_fused >= 0 ? r.force_fused != 0 : !r.prefer_stage;  // until the first batch: the scene's preference
  r.grid = r.batch_grid(0, r.fused);
  hold.p = nullptr;
  return R;
}
CrtRenderer *crt_renderer_new(CrtScene *scene, const CrtMaterial *materials, size_t n_materials, const CrtLight *lights,
                              size_t n_lights, const CrtCamera *camera, const CrtRenderSettings *settings,
                              uint32_t tile_rank, uint32_t tile_world) {
  CrtRenderer *R = nullptr;
  (void)abi_guard("crt_renderer_new", [&] {  // nothing unwinds through the ABI: NULL + crt_last_error
    R = renderer_new(scene, materials, n_materials, lights, n_lights, camera, settings, tile_rank, tile_world);
    return (int)CRT_OK;
  });
  return R;
}
void crt_renderer_free(CrtRenderer *r) { delete r; }
size_t crt_renderer_pixel_count(const CrtRenderer *r) { return r ? r->r.pixels.size() : 0; }
int crt_renderer_pixel_indices(const CrtRenderer *r, uint32_t *out) {
  if (!r || !out) return CRT_ERR_BAD_ARG;
  std::memcpy(out, r->r.pixels.data(), r->r.pixels.size() * 4);
  return CRT_OK;
}
int crt_render_samples(CrtRenderer *r, uint32_t sample_begin, uint32_t sample_count, void *stream) {
  if (!r) return CRT_ERR_BAD_ARG;
  return abi_guard("crt_render_samples", [&] { return r->r.render(sample_begin, sample_count, (hipStream_t)stream, nullptr); });
}
int crt_render_samples_stats(CrtRenderer *r, uint32_t sample_begin, uint32_t sample_count, void *stream,
                             CrtTravStats host_stats[2]) {
  if (!r || !host_stats) return CRT_ERR_BAD_ARG;
  CrtTravStats *d = nullptr;
  if (!CRT_HIP_OK(hipMalloc(&d, 2 * sizeof(CrtTravStats)))) return CRT_ERR_NO_DEVICE;
  (void)hipMemsetAsync(d, 0, 2 * sizeof(CrtTravStats), (hipStream_t)stream);
  int rc = abi_guard("crt_render_samples_stats", [&] { return r->r.render(sample_begin, sample_count, (hipStream_t)stream, d); });
  CrtTravStats h[2] = {};
  if (rc == CRT_OK && !CRT_HIP_OK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream))) rc = CRT_ERR_NO_DEVICE;
  if (!CRT_HIP_OK(hipStreamSynchronize((hipStream_t)stream))) rc = CRT_ERR_NO_DEVICE;
  (void)hipFree(d);
  if (rc == CRT_OK) {
    for (int w = 0; w < 2; w++) {
      for (int k = 0; k < 2; k++) {
        host_stats[w].queries[k] += h[w].queries[k]; host_stats[w].nodes[k] += h[w].nodes[k];
        host_stats[w].leaves[k] += h[w].leaves[k]; host_stats[w].packets[k] += h[w].packets[k];
        host_stats[w].prims[k] += h[w].prims[k];
      }
      host_stats[w].accepted_hits += h[w].accepted_hits; host_stats[w].instance_descents += h[w].instance_descents;
      host_stats[w].rays += h[w].rays;
      for (int k = 0; k < 8; k++) { host_stats[w].phase_waves[k] += h[w].phase_waves[k]; host_stats[w].phase_lanes[k] += h[w].phase_lanes[k]; host_stats[w].phase_cycles[k] += h[w].phase_cycles[k]; }
    }
  }
  return rc;
}
int crt_film_resolve(CrtRenderer *r, float *d_rgb, void *stream) {
  if (!r || !d_rgb) return CRT_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_film_out, dim3(r->r.grid), dim3(kBlock), 0, (hipStream_t)stream, r->r.film, r->r.P.n_pix, d_rgb);
  return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : CRT_ERR_NO_DEVICE;
}
int crt_film_read(CrtRenderer *r, float *host_rgb) {
  if (!r || !host_rgb) return CRT_ERR_BAD_ARG;
  float *d = nullptr;
  const size_t bytes = (size_t)r->r.P.n_pix * 12;
  if (!CRT_HIP_OK(hipMalloc(&d, bytes))) return CRT_ERR_NO_DEVICE;
  int rc = crt_film_resolve(r, d, r->r.last_stream);
  if (rc == CRT_OK && !CRT_HIP_OK(hipStreamSynchronize(r->r.last_stream))) rc = CRT_ERR_NO_DEVICE;
  if (rc == CRT_OK && !CRT_HIP_OK(hipMemcpy(host_rgb, d, bytes, hipMemcpyDeviceToHost))) rc = CRT_ERR_NO_DEVICE;
  (void)hipFree(d);
  return rc;
}
int crt_film_clear(CrtRenderer *r, void *stream) {
  if (!r) return CRT_ERR_BAD_ARG;
  bool ok = CRT_HIP_OK(hipMemsetAsync(r->r.film, 0, (size_t)r->r.P.n_pix * 16, (hipStream_t)stream));
  for (Renderer::Lane &B : r->r.lanes) ok = ok && CRT_HIP_OK(hipMemsetAsync(B.C, 0, sizeof(Counters), (hipStream_t)stream));
  if (ok && r->r.d_state) {
    ok = CRT_HIP_OK(hipMemsetAsync(r->r.d_state, 0, (size_t)r->r.P.n_pix * 4, (hipStream_t)stream)) &&
         CRT_HIP_OK(hipMemsetAsync(r->r.d_pstats, 0, (size_t)r->r.P.n_pix * sizeof(PixelStats), (hipStream_t)stream));
    r->r.n_act = r->r.P.n_pix;
  }
  return ok ? CRT_OK : CRT_ERR_NO_DEVICE;
}
int crt_render_stats(CrtRenderer *r, CrtRayStats *out) {
  if (!r || !out) return CRT_ERR_BAD_ARG;
  struct { uint32_t err, pad; unsigned long long stats[8]; } h;  // the head of Counters
  static_assert(offsetof(Counters, seg) == sizeof h, "Counters head layout");
  if (!CRT_HIP_OK(hipStreamSynchronize(r->r.last_stream))) return CRT_ERR_NO_DEVICE;
  unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t err = 0;
  for (const Renderer::Lane &B : r->r.lanes) {  // the lanes of a batch count their own rays
    if (!CRT_HIP_OK(hipMemcpy(&h, B.C, sizeof h, hipMemcpyDeviceToHost))) return CRT_ERR_NO_DEVICE;
    for (int k = 0; k < 8; k++) sum[k] += h.stats[k];
    err |= h.err;
  }
  out->camera_rays = sum[0]; out->closest_hit = sum[1]; out->shadow_rays = sum[2];
  out->vertices = sum[3]; out->rr_tested = sum[4]; out->rr_killed = sum[5];
  out->ended_escaped = sum[6]; out->ended_depth = sum[7];
  return err ? CRT_ERR_STACK : CRT_OK;
}
size_t crt_renderer_active_pixels(const CrtRenderer *r) { return r ? (r->r.variance_threshold > 0.0f ? r->r.n_act : r->r.P.n_pix) : 0; }
int crt_renderer_sample_counts(CrtRenderer *r, uint32_t *host_out) {
  if (!r || !host_out) return CRT_ERR_BAD_ARG;
  if (!r->r.d_state) return CRT_ERR_UNSUPPORTED;
  if (!CRT_HIP_OK(hipStreamSynchronize(r->r.last_stream))) return CRT_ERR_NO_DEVICE;
  if (!CRT_HIP_OK(hipMemcpy(host_out, r->r.d_state, (size_t)r->r.P.n_pix * 4, hipMemcpyDeviceToHost))) return CRT_ERR_NO_DEVICE;
  for (size_t k = 0; k < r->r.P.n_pix; k++) host_out[k] &= 0x7fffffffu;
  return CRT_OK;
}
int crt_renderer_shade_class_stats(CrtRenderer *r, int enable, uint64_t out_waves[4], uint64_t out_lanes[4]) {
  if (!r) return CRT_ERR_BAD_ARG;
  Renderer &R = r->r;
  if (out_waves && out_lanes) {
    unsigned long long h[8];
    if (!CRT_HIP_OK(hipStreamSynchronize(R.last_stream))) return CRT_ERR_NO_DEVICE;
    for (int c = 0; c < 4; c++) out_waves[c] = out_lanes[c] = 0;
    for (Renderer::Lane &B : R.lanes) {
      if (!CRT_HIP_OK(hipMemcpy(h, B.C->cls_waves, sizeof h, hipMemcpyDeviceToHost))) return CRT_ERR_NO_DEVICE;
      if (!CRT_HIP_OK(hipMemset(B.C->cls_waves, 0, sizeof h))) return CRT_ERR_NO_DEVICE;
      for (int c = 0; c < 4; c++) { out_waves[c] += h[c]; out_lanes[c] += h[4 + c]; }
    }
  }
  if (enable >= 0) R.P.class_stats = enable ? 1u : 0u;
  return CRT_OK;
}
int crt_renderer_lanes(const CrtRenderer *r) { return r ? r->r.last_lanes : 0; }
int crt_renderer_set_lanes(CrtRenderer *r, int lanes) {
  if (!r) return CRT_ERR_BAD_ARG;
  r->r.n_lanes = lanes < 1 ? 1 : (lanes > Renderer::kMaxLanes ? Renderer::kMaxLanes : lanes);
  return r->r.n_lanes;
}
int crt_renderer_pipeline(const CrtRenderer *r, uint32_t out[3]) {
  if (!r || !out) return CRT_ERR_BAD_ARG;
  out[0] = r->r.fused ? 1u : 0u;
  out[1] = (!r->r.fused && r->r.wide) ? 1u : 0u;
  out[2] = (uint32_t)r->r.grid;
  return CRT_OK;
}
int crt_renderer_profile(CrtRenderer *r, int enable) {
  if (!r) return CRT_ERR_BAD_ARG;
  r->r.drain_events();
  r->r.profile = enable != 0;
  for (int k = 0; k < 4; k++) { r->r.prof_ms[k] = 0; r->r.prof_launches[k] = 0; }
  return CRT_OK;
}
int crt_renderer_profile_read(CrtRenderer *r, double out_ms[4], uint64_t out_launches[4]) {
  if (!r || !out_ms || !out_launches) return CRT_ERR_BAD_ARG;
  r->r.drain_events();
  for (int k = 0; k < 4; k++) { out_ms[k] = r->r.prof_ms[k]; out_launches[k] = r->r.prof_launches[k]; }
  return CRT_OK;
}

}  // extern "C"
