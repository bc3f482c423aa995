// Internal host/device declarations of libcrt_amd.so (not part of the ABI).
//
// Host side (C++17): geometry table, deterministic SBVH -> BVH4 build, flattening of a committed
// scene and everything it instances into ONE device image with absolute indices.
// Device side (HIP, gfx950): see kernels/.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/crt.h"

namespace crt {

// ---------------------------------------------------------------------------------------------
// Small float3 with glam Vec3A (SSE2) semantics. Compiled -ffp-contract=off.
// ---------------------------------------------------------------------------------------------
struct F3 {
  float x, y, z;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  float &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline F3 f3(float x, float y, float z) { return F3{x, y, z}; }
inline F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline F3 operator-(F3 a) { return {-a.x, -a.y, -a.z}; }
inline float sse_min(float a, float b) { return a < b ? a : b; }  // minps
inline float sse_max(float a, float b) { return a > b ? a : b; }  // maxps
inline F3 vmin(F3 a, F3 b) { return {sse_min(a.x, b.x), sse_min(a.y, b.y), sse_min(a.z, b.z)}; }
inline F3 vmax(F3 a, F3 b) { return {sse_max(a.x, b.x), sse_max(a.y, b.y), sse_max(a.z, b.z)}; }
inline float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

struct Aabb { F3 mn, mx; };
struct Affine { F3 x, y, z, t; };  // glam Affine3A: matrix3 columns + translation
struct Mat3 { F3 x, y, z; };

Affine affine_inverse(const Affine &a);
Mat3 mat3_transpose(const Mat3 &m);
F3 affine_point(const Affine &a, F3 p);
F3 affine_vector(const Affine &a, F3 p);
Aabb transformed_aabb(const Aabb &local, const Affine &m);  // prim.rs:298-319

// ---------------------------------------------------------------------------------------------
// Committed-tree records. These are the reference's layouts (bvh.rs:196-227, triangle.rs:182-196)
// and also the device layouts: one node = one 128-byte line, one packet = 192 bytes.
// ---------------------------------------------------------------------------------------------
struct alignas(16) WideNode {
  float bmin[3][4];
  float bmax[3][4];
  uint32_t child[4];
  uint32_t flags;  // bits 0-3 valid lane, bits 4-7 leaf lane
  uint32_t pad[3];
};
static_assert(sizeof(WideNode) == 128, "bvh.rs:1613-1616");

struct Leaf { uint32_t pkt_first, pkt_count, idx_first, idx_count; };
static_assert(sizeof(Leaf) == 16, "bvh.rs:220-227");

struct alignas(16) Tri4 {
  float v[3][3][4];  // [vertex][axis][lane]
  uint32_t prim[4];
  uint32_t active, mask_and, mask_or;
  uint32_t masks[4];
  // Device extra (the reference's padding word): bit k set iff lane k can produce a normal, i.e. it
  // has shading normals or a non-zero geometric cross product (prim.rs:76-86 rejects the others).
  uint32_t normal_ok;
};
static_assert(sizeof(Tri4) == 192, "triangle.rs:182-196");

enum PrimKind : uint32_t { PRIM_TRI = 0, PRIM_SPHERE = 1, PRIM_INSTANCE = 2 };

struct Scene;

// Host primitive (prim.rs:60-70, :125-130, :261-280).
struct Prim {
  PrimKind kind;
  uint32_t geom_id, prim_id, mask;
  F3 v0, v1, v2;
  bool has_normals;
  F3 n0, n1, n2;
  F3 center;
  float radius;
  std::shared_ptr<Scene> scene;
  Affine l2w, w2l;
  Mat3 normal_mat;
  bool has_end;
  Affine l2w_end;
  Aabb bounds;
};

struct Bvh {
  std::vector<WideNode> wide;
  std::vector<Leaf> leaves;
  std::vector<Tri4> packets;
  std::vector<uint32_t> indices;
  std::vector<Prim> prims;
  bool has_bbox = false;
  Aabb root_bbox{};
};

// Instance nesting the traversal kernels carry frames for (kernels/traverse.hip.h kMaxLevels; the importer's own
// limit, usd_import.rs:60 MAX_INSTANCE_NESTING): commit refuses deeper scenes instead of skipping instances at trace time.
constexpr uint32_t kMaxInstanceLevels = 8;

Aabb prim_bbox(const Prim &p);
void build_bvh(Bvh &out, std::vector<Prim> &&prims);  // bvh.rs:300-327

// ---------------------------------------------------------------------------------------------
// Device image (HBM): the queried scene and every scene it instances, flattened.
// ---------------------------------------------------------------------------------------------
struct DevPrim {  // 64 bytes
  uint32_t kind, geom_id, prim_id, mask;
  float d[12];  // tri: v0 v1 v2 (9) + [9]=smooth-normal slot (u32 bits, ~0 = none)
                // sphere: center (3), radius | instance: [0] = instance slot (u32 bits)
};
static_assert(sizeof(DevPrim) == 64, "");

struct DevInstance {  // 64 bytes: what an instance entry reads (half a cache line; 40 000 placements stay L2-resident)
  float w2l[12];      // cached world-to-local at time 0 (prim.rs:266); its matrix3 transposed is the normal matrix (:267)
  uint32_t root;      // absolute node index of the instanced scene's root
  uint32_t flags;     // bit 0: the instanced tree has Tri4 packets; bit 1: transform motion blur (prim.rs:276);
                      // bits 2..31: float offset of its DevInstanceMotion in `normals` (moving instances only)
  uint32_t geom_id;   // the instance's own geometry id and ray mask: a leaf's index list addresses an instance
  uint32_t mask;      //   record directly (kIndexInstance), so entering one never touches its DevPrim
};
static_assert(sizeof(DevInstance) == 64, "");
// The two placements of a MOVING instance (prim.rs:285-331; flags bit 1), read only by a ray with a shutter time > 0:
// 24 floats in the `normals` array at float offset flags >> 2. (Not an array of its own: every pointer of DevScene
// is a pair of scalar registers live through the whole traversal, and the path kernels already spill those.)
struct DevInstanceMotion {
  float l2w[12];
  float l2w_end[12];
};
// Entries of `indices` (a leaf's scalar list): primitive index, or kIndexInstance | instance slot. Instance slots
// follow the order of the scalar lists, so the placements of one leaf, and of neighbouring leaves, share cache lines.
constexpr uint32_t kIndexInstance = 0x80000000u;

struct DevScene {
  const WideNode *nodes;
  const Leaf *leaves;
  const Tri4 *packets;
  const uint32_t *indices;
  const DevPrim *prims;
  const DevInstance *instances;
  const float *normals;  // 9 floats per smooth triangle; then 24 per moving instance
  uint32_t root;         // CRT_INVALID_ID when the scene is empty (bvh.rs:442-444)
  uint32_t has_packets;
  uint32_t n_nodes;      // numbered for the LDS window: top of the top-level tree, the instanced trees' roots, then
                         // the rest breadth-first (see Scene::ensure_device)
  uint32_t pool_stack;   // LDS stack entries per ray the traversal engine uses for this scene (6 flat, 10 instanced)
  uint32_t n_packets;    // Tri4 packets, the queried tree's first (scene.cpp rotates them there; crt_scene_image_check): the first ones are staged in LDS behind the node window
  uint32_t direct_leaves;  // leaves without packets and with 1-3 scalar entries are encoded in the child word (below)
  uint32_t cold;           // kCold* bits: the rarely used per-ray state a traversal of this scene can need (traverse_pool.hip.h)
};
// DevScene::cold. kColdUV: some triangle has shading normals (prim.rs:76-95 interpolates them with u, v). kColdNormal:
// a sphere exists somewhere (its normal is computed at the hit and kept until emit), or instances nest deeper than one
// level (a hit below the first level is taken to its parent's space at exit). kColdTime: a moving instance exists.
enum : uint32_t { kColdUV = 1u, kColdNormal = 2u, kColdTime = 4u, kColdAll = 7u };
// A fourth bit of the KERNELS' cold argument (never of DevScene::cold): the image holds no Tri4 packet at all and carries
// direct leaf words (a scene of analytic spheres: openpbr_showcase) — the kernel instance then holds ONE engine copy, the
// direct one, without the packet phase's Woop test and without the per-ray shear constants (traverse_pool.hip.h, NOPK).
constexpr uint32_t kNoPackets = 8u;
// Device child words of a node: inner child = node index; leaf child = kLeafTag | leaf index; empty lane =
// CRT_INVALID_ID. With DevScene::direct_leaves a leaf that holds no Tri4 packet and one to three scalar entries
// (spheres, instances: every leaf of an instanced city's top-level tree) is written as
//   kLeafTag | kDirectLeafTag | count << 28 | idx_first                    entries indices[idx_first ..)
//   kLeafTag | kDirectLeafTag | count << 28 | kDirectInstTag | slot       entries = instances slot, slot + 1, ..
// so the ray goes from the node straight to the scalar list: no 16-byte Leaf fetch, no packet step that finds
// nothing and, when the entries are instances in consecutive slots, no index fetch either. Chosen per scene at
// upload (instance-heavy or packet-free scenes), like the LDS split.
constexpr uint32_t kDirectLeafTag = 0x40000000u;
constexpr uint32_t kDirectInstTag = 0x08000000u;
constexpr uint32_t kDirectIndexMask = 0x07ffffffu;
// Nodes of the top of the tree the traversal kernels stage in LDS per workgroup (kernels/traverse_pool.hip.h): the
// upload numbers the nodes for this window.
#ifndef CRT_POOL_NODES_WIDE
#define CRT_POOL_NODES_WIDE 26  // the four-workgroups-per-CU kernels (small flat scenes)
#endif
#ifndef CRT_POOL_NODES_WIDE_DEEP
#define CRT_POOL_NODES_WIDE_DEEP 12  // ... of the four-wave kernels on a large tree or an instance-heavy scene (they trade the window for stack
                                     // entries; 12 is what the 40 KB arena has left beside five entries: MedCity's root + its 8 prototype roots fit, +1.3 % over 8)
#endif
#ifndef CRT_DIRECT_INST
#define CRT_DIRECT_INST 1  // 0: the direct-instance form is neither written nor understood (A/B builds)
#endif
#ifndef CRT_POOL_NODES
#define CRT_POOL_NODES 72
#endif
#ifndef CRT_POOL_NODES_DEEP
#define CRT_POOL_NODES_DEEP 16  // with the deep stack: 16 still fit three workgroups per CU (+0.4 %); 24 do not (-33 %)
#endif
#ifndef CRT_WIDE_DIRECT_BUILD
#define CRT_WIDE_DIRECT_BUILD 1  // four-wave kernel instances with the direct engine copy (WIDE = 2); CRT_WIDE=2 asks for them
#endif
#ifndef CRT_NOPK_BUILD
// 1: packet-free instances of the fused kernel for images without a Tri4 packet (kNoPackets). Measured on openpbr_showcase
// (round 4): one engine copy and 13 % fewer instructions, but 53 -> 67 spilled registers in k_path<2, true, .>: 12 294 ->
// 12 176 Mray/s. Off; the switch stays for A/B builds.
#define CRT_NOPK_BUILD 0
#endif
#ifndef CRT_DIRECT_LEAVES
#define CRT_DIRECT_LEAVES 1  // 0: neither written by the upload nor understood by the engine (A/B builds)
#endif

// Which scenes the four-workgroups-per-CU traversal kernels (kernels/traverse_pool.hip.h, WIDE: three stack entries
// per ray in LDS) are launched for: flat (not instance-heavy) triangle scenes whose trees are shallow enough for that
// stack. Measured (profiles/README.md): the renderer's per-stage pipeline on them gains 4-9 % on cornellbox, veach_mis,
// sun_sky (at most a few hundred nodes); 16 M incoherent rays against 43 200 triangles (3 600 nodes, 15 nodes per ray)
// lose 7 %, an instanced city 6 %, sphere-only scenes are indifferent.
// ... and only images WITHOUT direct child words: the four-wave kernels are built without the direct-leaf engine copy
// (run_traversal), so a direct word would be read as a leaf index (the round-3 fault, profiles/README.md).
// Round 4: LARGE flat trees too, in the renderer — its four-wave kernels keep no mask plane and split their arena by tree
// (five stack entries + a twelve-node window; traverse_pool.hip.h): the reference's stress scene 2538 -> 2738 Mray/s
// (+7.9 %), the 7 M-triangle synthetic scene +0.7 %. The batched queries (one stack entry less: per-ray masks) keep the
// three-wave kernels on those trees, as do the mid-size ones (1 024 < nodes <= 2 048: the flat 6 + 72 split).
inline bool wide_split(const DevScene &s, bool renderer = false) {
  if (s.direct_leaves != 0 || s.n_packets == 0) return false;
  if (s.pool_stack >= 10u) return renderer && s.n_nodes > 2048u;  // the deep split chosen for a large tree, not for instances
  return s.n_nodes <= 1024u;
}

// LDS stack entries per ray of the three engine splits (kernels/traverse_pool.hip.h sizes its arenas from these).
#ifndef CRT_POOL_STACK
#define CRT_POOL_STACK 6        // flat scenes, three-wave kernels
#endif
#ifndef CRT_POOL_STACK_DEEP
#define CRT_POOL_STACK_DEEP 10  // instance-heavy scenes and large trees, three-wave kernels
#endif
#ifndef CRT_POOL_STACK_WIDE
#define CRT_POOL_STACK_WIDE 3   // four-wave kernels
#endif

// Which instance of the traversal engine runs an image. THE one decision: the renderer (pathtrace.hip), the batched and
// the single-ray queries (traverse.hip) and the host-only crt_scene_engine_select all call select_engine, and every
// launch goes through the EngineSelect it returns — an engine instance never meets an image form it was not built for
// (rounds 2 and 3 each lost a GPU box to exactly that: profiles/README.md, "The r02f abort", "The r03w fault").
struct EngineSelect {
  bool wide = false;       // the four-workgroups-per-CU kernels (WIDE): the flat engine copy, 3-5 LDS stack entries (lds_stack)
  bool wide_direct = false;  // ... their instances that hold the DIRECT engine copy only (WIDE = 2): direct-leaf images
  bool direct = false;     // the DIRECT engine copy of the three-wave kernels: reads direct child words, root test at entry
  uint32_t lds_stack = CRT_POOL_STACK;  // stack entries per ray in LDS
  uint32_t window = CRT_POOL_NODES;     // nodes staged in LDS per workgroup
  int ext_cold = (int)kColdAll;         // k_extend<., ., COLD>: none / the pending normal only / everything
  int path_cold = (int)kColdAll;        // k_path<., ., COLD> of simple scenes: none / everything
};
// want_wide: -1 = the scene's own preference (wide_split), 0 / 1 = asked for (CRT_WIDE, tests); renderer: the choice for
// the renderer's kernels (the batched queries' differs on large trees, wide_split). CRT_OK, or
// CRT_ERR_UNSUPPORTED when what was asked for cannot decode the image — nothing is launched then.
inline int select_engine(const DevScene &s, int want_wide, EngineSelect &e, bool renderer = false) {
  const bool has_direct_words = s.direct_leaves != 0;
  // the plain WIDE kernels carry no direct-leaf engine; their WIDE = 2 instances (the renderer's only) carry nothing else
  if (want_wide == 1 && has_direct_words) return CRT_ERR_UNSUPPORTED;
  if (want_wide == 2 && !(has_direct_words && renderer && CRT_DIRECT_LEAVES != 0 && CRT_WIDE_DIRECT_BUILD != 0)) return CRT_ERR_UNSUPPORTED;
  // Round 4: the renderer runs DIRECT-LEAF images (instance-heavy or packet-free scenes) on the four-wave kernels too, on
  // instances that hold the direct engine copy only — an entry the root test rejects never touches the stack, so the
  // direct engine lives with the short LDS stack where the flat one (gauged at 1 985 against 1 977 on MedCity) did not:
  // PointInstancedMedCity 2 234 -> 2 285 Mray/s, openpbr_showcase 13 070 -> 14 045, a 32 761-instance city 2 599 -> 2 790.
  const bool auto_wide_direct = want_wide < 0 && renderer && has_direct_words && CRT_DIRECT_LEAVES != 0 && CRT_WIDE_DIRECT_BUILD != 0;
  e.wide = auto_wide_direct || (want_wide < 0 ? wide_split(s, renderer) : want_wide != 0);
  e.wide_direct = auto_wide_direct || want_wide == 2;
  e.direct = (!e.wide || e.wide_direct) && CRT_DIRECT_LEAVES != 0 && has_direct_words;
  if (has_direct_words && !e.direct) return CRT_ERR_UNSUPPORTED;      // (a build without the direct form never writes one)
  const bool deep = s.pool_stack >= (uint32_t)CRT_POOL_STACK_DEEP;    // run_traversal's rule
  // (the wide arena's splits: the renderer's kernels, which keep no mask plane; the batched queries have one entry less)
  e.lds_stack = e.wide ? (uint32_t)CRT_POOL_STACK_WIDE + 1u + (deep ? 1u : 0u) : (deep ? (uint32_t)CRT_POOL_STACK_DEEP : (uint32_t)CRT_POOL_STACK);
  e.window = e.wide ? (deep ? (uint32_t)CRT_POOL_NODES_WIDE_DEEP : (uint32_t)CRT_POOL_NODES_WIDE)
                    : (deep ? (uint32_t)CRT_POOL_NODES_DEEP : (uint32_t)CRT_POOL_NODES);
  const int cold = (int)(s.cold & kColdAll);
  e.ext_cold = cold == 0 ? 0 : (cold == (int)kColdNormal ? (int)kColdNormal : (int)kColdAll);
  e.path_cold = cold == 0 ? 0 : (int)kColdAll;
  // a packet-free image with direct words: the fused kernel's packet-free instance (general materials; pathtrace.hip)
  if (CRT_NOPK_BUILD && s.n_packets == 0 && e.direct) e.path_cold = (int)(kColdAll | kNoPackets);
  return CRT_OK;
}
// The check every launch site makes on the EngineSelect it was handed (defence in depth: select_engine already
// guarantees it): the instance can decode every child word of the image and keeps every cold field the image can need.
inline bool engine_accepts(const EngineSelect &e, const DevScene &s, int kernel_cold) {
  if (s.direct_leaves != 0 && ((e.wide && !e.wide_direct) || !e.direct)) return false;
  if (e.wide_direct && s.direct_leaves == 0) return false;
  if ((kernel_cold & (int)kNoPackets) && (s.n_packets != 0 || s.direct_leaves == 0)) return false;  // a packet-free instance on packets
  return ((int)(s.cold & kColdAll) & ~kernel_cold) == 0;
}
// CRT_WIDE (A/B runs, tests): 1 asks for the four-wave kernels, 0 for the three-wave ones. A request the image cannot
// take falls back to the scene's own preference — the knob sweeps whole test sets, direct-leaf scenes included.
int wide_request();  // -1 unset

// Every A/B knob of the library. The environment is read in ONE place (scene.cpp, read_knobs) — when a scene image is
// flattened and when a renderer is created — validated and clamped there; nothing else in the library calls getenv.
// -1 = not set (the library's own choice). None of them changes a result (tests/test_gpu_render.py,
// test_round3_knobs_change_no_bit); profiles/README.md lists what each was measured for.
struct Knobs {
  // scene image (flatten_image)
  int pool_stack_deep = -1;   // CRT_POOL_STACK_RT: 1 the deep LDS split, 0 the flat one (any value >= 10 / below)
  int direct_leaves = -1;     // CRT_DIRECT_LEAVES
  int direct_inst = -1;       // CRT_DIRECT_INST (can only switch the form OFF)
  int stage_roots = -1;       // CRT_STAGE_ROOTS: at most n instanced roots in the LDS window
  uint32_t cold = 0;          // CRT_COLD: cold bits the image asks for on top of what it needs (0..7)
  int inst_order = 1;         // CRT_INST_ORDER
  int hot_packets = 1;        // CRT_HOT_PACKETS
  // engine choice
  int wide = -1;              // CRT_WIDE
  // renderer
  int mat_dedup = 1, partition = -1, simple = 1, prefer_stage = -1, cam_compact = 1, shade_wide = -1, fused = -1;
  int noclassify_from = 1 << 30, tail_from = 12, lanes = 4, grid_mult = 0;
  size_t max_batch_slots = 0, lane_min_paths = (size_t)96 << 20, stage_min_paths = (size_t)96 << 20;
};
Knobs read_knobs();
inline int select_engine_env(const DevScene &s, EngineSelect &e, bool renderer = false) {
  if (select_engine(s, wide_request(), e, renderer) == CRT_OK) return CRT_OK;
  return select_engine(s, -1, e, renderer);
}

struct DeviceImage {
  void *blob = nullptr;
  size_t bytes[7] = {0, 0, 0, 0, 0, 0, 0};  // nodes, leaves, packets, indices, prims, instances, normals (+ placements of moving instances)
  uint32_t *err = nullptr;  // this scene's traversal error word (the blob's last 256 bytes): crt_scene_traversal_error
  DevScene view{};
  ~DeviceImage();
};

struct Scene : std::enable_shared_from_this<Scene> {
  Bvh bvh;
  uint32_t n_geoms = 0;
  bool has_motion = false;
  uint32_t depth = 1;  // levels of instancing below and including this scene: 1 = no instances
  std::mutex dev_mu;
  std::unique_ptr<DeviceImage> dev;  // built on first query
  int ensure_device();               // CRT_OK or CRT_ERR_NO_DEVICE
};

// Host-only self-check of the image Scene::ensure_device would upload (scene.cpp): CRT_OK and eight counts, or
// CRT_ERR_BAD_ARG with the broken invariant in crt_last_error.
int scene_image_check(const Scene &scene, uint64_t out[8]);
// Host-only: select_engine on the image this scene would upload, verified against a census of the image (scene.cpp).
int scene_engine_select(const Scene &scene, int want_wide, uint32_t out[8]);

enum GeomKind { G_MESH, G_SPHERE, G_INSTANCE };
struct Geom {
  GeomKind kind = G_MESH;
  uint32_t mask = CRT_MASK_ALL;
  std::vector<float> verts;
  std::vector<uint32_t> idx;
  bool has_normals = false;
  std::vector<float> normals;
  F3 center{0, 0, 0};
  float radius = 0;
  std::shared_ptr<Scene> scene;
  Affine l2w{};
  bool has_end = false;
  Affine l2w_end{};
};

struct Builder { std::vector<Geom> geoms; };

std::shared_ptr<Scene> commit(Builder &&b);  // scene.rs:226-341

// ---------------------------------------------------------------------------------------------
// Device launches (kernels/traverse.hip)
// ---------------------------------------------------------------------------------------------
// d_err: device word the kernels OR their error bits into (1 = traversal stack overflow, 2 = instance nesting).
int launch_intersect_n(const DevScene &s, const CrtRay *d_rays, size_t n, float t_min, float t_max, CrtRayHit *d_hits,
                       void *stream, CrtTravStats *d_stats, uint32_t *d_err);
int launch_occluded_n(const DevScene &s, const CrtRay *d_rays, size_t n, float t_min, float t_max, uint32_t *d_out,
                      void *stream, CrtTravStats *d_stats, uint32_t *d_err);
int device_ok();
// printf-style text for crt_last_error() on this thread (failures that are not HIP calls).
void set_error_text(const char *fmt, ...);
// Nothing may unwind through the C ABI — a host in C or Rust cannot catch it, and unwinding into its frames is undefined:
// an entry point whose body allocates host memory (std containers) runs it through abi_guard; a failed allocation (or a
// length_error: crt_reserve(b, SIZE_MAX)) becomes CRT_ERR_NO_MEMORY with the reason in crt_last_error.
template <class F>
inline int abi_guard(const char *what, F &&body) noexcept {
  try { return body(); }
  catch (const std::exception &e) { set_error_text("%s: %s", what, e.what()); }
  catch (...) { set_error_text("%s: unknown failure", what); }
  return CRT_ERR_NO_MEMORY;
}

// Records the failing HIP call for crt_last_error() and returns false.
bool hip_failed(int /*hipError_t*/ err, const char *what, const char *file, int line);
#define CRT_HIP_OK(call) (!::crt::hip_failed((int)(call), #call, __FILE__, __LINE__))

}  // namespace crt

struct CrtScene { std::shared_ptr<crt::Scene> p; std::atomic<int> refs{1}; };
struct CrtBuilder { crt::Builder b; };
