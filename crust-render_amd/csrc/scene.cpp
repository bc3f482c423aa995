// SceneBuilder::commit and the device image.
//
// commit() follows crates/crust-rt/src/scene.rs:226-341: geometries expand to primitives in attach order,
// out-of-range triangles and empty instanced scenes are skipped silently, instances cache w2l and the
// inverse transpose. The device image flattens the queried scene plus everything it instances (to any
// depth) into seven HBM arrays with absolute indices, so the kernel follows an instance by jumping to
// another root node in the same arrays.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <utility>
#include <unordered_map>

#include <cstdlib>
#include "crt_internal.h"

namespace crt {

// ---- glam Affine3A / Mat3A (SSE2 code paths) ----
static F3 mat3_mul(const Mat3 &m, F3 r) {
  F3 res = m.x * r.x;
  res = res + m.y * r.y;
  res = res + m.z * r.z;
  return res;
}
Mat3 mat3_transpose(const Mat3 &m) {
  return {f3(m.x.x, m.y.x, m.z.x), f3(m.x.y, m.y.y, m.z.y), f3(m.x.z, m.y.z, m.z.z)};
}
static Mat3 mat3_inverse(const Mat3 &m) {
  F3 t0 = cross(m.y, m.z), t1 = cross(m.z, m.x), t2 = cross(m.x, m.y);
  float det = dot(m.z, t2);
  float inv = 1.0f / det;
  return mat3_transpose(Mat3{t0 * inv, t1 * inv, t2 * inv});
}
Affine affine_inverse(const Affine &a) {
  Mat3 inv = mat3_inverse(Mat3{a.x, a.y, a.z});
  F3 t = -mat3_mul(inv, a.t);
  return {inv.x, inv.y, inv.z, t};
}
F3 affine_point(const Affine &a, F3 p) { return mat3_mul(Mat3{a.x, a.y, a.z}, p) + a.t; }
F3 affine_vector(const Affine &a, F3 p) { return mat3_mul(Mat3{a.x, a.y, a.z}, p); }

Aabb transformed_aabb(const Aabb &local, const Affine &m) {  // prim.rs:298-319
  const float inf = __builtin_inff();
  F3 mn = f3(inf, inf, inf), mx = f3(-inf, -inf, -inf);
  for (int i = 0; i < 8; i++) {
    F3 corner = f3((i & 1) == 0 ? local.mn.x : local.mx.x, (i & 2) == 0 ? local.mn.y : local.mx.y,
                   (i & 4) == 0 ? local.mn.z : local.mx.z);
    F3 p = affine_point(m, corner);
    mn = vmin(mn, p);
    mx = vmax(mx, p);
  }
  constexpr float PAD = 1e-4f;
  for (int a = 0; a < 3; a++)
    if (mx[a] - mn[a] < PAD) { mn.at(a) -= PAD; mx.at(a) += PAD; }
  return {mn, mx};
}

std::shared_ptr<Scene> commit(Builder &&b) {  // scene.rs:226-341
  auto scene = std::make_shared<Scene>();
  scene->n_geoms = uint32_t(b.geoms.size());
  size_t total = 0;
  for (const Geom &g : b.geoms) total += g.kind == G_MESH ? g.idx.size() / 3 : 1;
  std::vector<Prim> prims;
  prims.reserve(total);
  bool has_motion = false;
  uint32_t depth = 1;
  for (size_t gi = 0; gi < b.geoms.size(); gi++) {
    Geom &g = b.geoms[gi];
    const uint32_t geom_id = uint32_t(gi);
    switch (g.kind) {
      case G_MESH: {
        const size_t nv = g.verts.size() / 3, nn = g.normals.size() / 3, nt = g.idx.size() / 3;
        auto vert = [&](const std::vector<float> &a, size_t i) { return f3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); };
        for (size_t t = 0; t < nt; t++) {
          size_t i0 = g.idx[3 * t], i1 = g.idx[3 * t + 1], i2 = g.idx[3 * t + 2];
          if (i0 >= nv || i1 >= nv || i2 >= nv) continue;  // scene.rs:251-253
          Prim p{};
          p.kind = PRIM_TRI;
          p.geom_id = geom_id;
          p.prim_id = uint32_t(t);
          p.mask = g.mask;
          p.v0 = vert(g.verts, i0);
          p.v1 = vert(g.verts, i1);
          p.v2 = vert(g.verts, i2);
          if (g.has_normals && i0 < nn && i1 < nn && i2 < nn) {  // scene.rs:254-257
            p.has_normals = true;
            p.n0 = vert(g.normals, i0);
            p.n1 = vert(g.normals, i1);
            p.n2 = vert(g.normals, i2);
          }
          prims.push_back(std::move(p));
        }
        break;
      }
      case G_SPHERE: {
        Prim p{};
        p.kind = PRIM_SPHERE;
        p.geom_id = geom_id;
        p.mask = g.mask;
        p.center = g.center;
        p.radius = g.radius;
        prims.push_back(std::move(p));
        break;
      }
      case G_INSTANCE: {
        if (!g.scene || !g.scene->bvh.has_bbox) break;  // empty instanced scene (scene.rs:307-309)
        const Aabb inner = g.scene->bvh.root_bbox;
        Prim p{};
        p.kind = PRIM_INSTANCE;
        p.geom_id = geom_id;
        p.mask = g.mask;
        p.scene = g.scene;
        p.l2w = g.l2w;
        p.w2l = affine_inverse(g.l2w);
        p.normal_mat = mat3_transpose(Mat3{p.w2l.x, p.w2l.y, p.w2l.z});
        p.has_end = g.has_end;
        p.l2w_end = g.l2w_end;
        if (g.has_end) {
          Aabb a = transformed_aabb(inner, g.l2w), e = transformed_aabb(inner, g.l2w_end);
          p.bounds = Aabb{vmin(a.mn, e.mn), vmax(a.mx, e.mx)};
        } else {
          p.bounds = transformed_aabb(inner, g.l2w);
        }
        has_motion |= g.has_end || g.scene->has_motion;  // scene.rs:322
        if (g.scene->depth + 1 > depth) depth = g.scene->depth + 1;
        prims.push_back(std::move(p));
        break;
      }
    }
  }
  if (depth > kMaxInstanceLevels) {
    // The reference's kernel recurses without a limit (prim.rs:345-378); its importer stops at 8 levels
    // (usd_import.rs:60). The device kernels keep one frame per level, so a deeper scene is refused here, loudly,
    // instead of silently skipping its innermost instances at trace time.
    set_error_text("commit: %u levels of instance nesting, the kernels carry %u", depth, kMaxInstanceLevels);
    return nullptr;
  }
  scene->depth = depth;
  scene->has_motion = has_motion;
  build_bvh(scene->bvh, std::move(prims));
  return scene;
}

// ---------------------------------------------------------------------------------------------
// Device image
// ---------------------------------------------------------------------------------------------
DeviceImage::~DeviceImage() {
  if (blob) (void)hipFree(blob);
}

static thread_local char g_last_error[512] = "";
const char *last_error_text() { return g_last_error; }
bool hip_failed(int err, const char *what, const char *file, int line) {
  if (err == (int)hipSuccess) return false;
  std::snprintf(g_last_error, sizeof g_last_error, "%s -> %s (%d) at %s:%d", what, hipGetErrorString((hipError_t)err), err,
                file, line);
  return true;
}

void set_error_text(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
  va_end(ap);
}

// The library's only reader of the environment (crt_internal.h, Knobs).
Knobs read_knobs() {
  Knobs k;
  auto num = [](const char *name, long long &out) {
    const char *e = getenv(name);
    if (!e || !*e) return false;
    char *end = nullptr;
    const long long v = strtoll(e, &end, 10);
    if (end == e) return false;  // not a number: as if unset
    out = v;
    return true;
  };
  long long v;
  if (num("CRT_POOL_STACK_RT", v)) k.pool_stack_deep = v >= CRT_POOL_STACK_DEEP ? 1 : 0;  // a split, not a number the kernels interpret
  if (num("CRT_DIRECT_LEAVES", v)) k.direct_leaves = v != 0;
  if (num("CRT_DIRECT_INST", v)) k.direct_inst = v != 0;
  if (num("CRT_STAGE_ROOTS", v)) k.stage_roots = (int)std::min<long long>(std::max<long long>(v, 0), 1 << 20);
  if (num("CRT_COLD", v)) k.cold = (uint32_t)v & kColdAll;
  if (num("CRT_INST_ORDER", v)) k.inst_order = v != 0;
  if (num("CRT_HOT_PACKETS", v)) k.hot_packets = v != 0;
  if (num("CRT_WIDE", v)) k.wide = v == 2 ? 2 : (v != 0);  // 2: the four-wave kernels' direct-engine instances (direct-leaf images)
  if (num("CRT_MAT_DEDUP", v)) k.mat_dedup = v != 0;
  if (num("CRT_PARTITION", v)) k.partition = v != 0;
  if (num("CRT_SIMPLE", v)) k.simple = v != 0;
  if (num("CRT_PREFER_STAGE", v)) k.prefer_stage = v != 0;
  if (num("CRT_CAM_COMPACT", v)) k.cam_compact = v != 0;
  if (num("CRT_SHADE_WIDE", v)) k.shade_wide = v != 0;
  if (num("CRT_FUSED", v)) k.fused = v != 0;
  if (num("CRT_NOCLASSIFY_FROM", v)) k.noclassify_from = (int)std::min<long long>(std::max<long long>(v, 0), 1 << 30);
  if (num("CRT_TAIL_FROM", v)) k.tail_from = (int)std::min<long long>(std::max<long long>(v, 0), 1 << 30);
  if (num("CRT_LANES", v)) k.lanes = (int)std::min<long long>(std::max<long long>(v, 1), 4);
  if (num("CRT_GRID_MULT", v)) k.grid_mult = (int)std::min<long long>(std::max<long long>(v, 0), 64);
  if (num("CRT_MAX_BATCH_SLOTS", v)) k.max_batch_slots = v > 0 ? (size_t)v : 0;
  if (num("CRT_LANE_MIN_PATHS", v)) k.lane_min_paths = v > 0 ? (size_t)v : 1;
  if (num("CRT_STAGE_MIN_PATHS", v)) k.stage_min_paths = v > 0 ? (size_t)v : 1;
  return k;
}
int wide_request() { return read_knobs().wide; }

int device_ok() {
  static int ok = [] {
    int n = 0;
    if (!CRT_HIP_OK(hipGetDeviceCount(&n)) || n <= 0) return 0;
    return 1;
  }();
  return ok;
}

namespace {
struct Flat {
  std::vector<WideNode> nodes;
  std::vector<Leaf> leaves;
  std::vector<Tri4> packets;
  std::vector<uint32_t> indices;
  std::vector<DevPrim> prims;
  std::vector<DevInstance> instances;
  std::vector<std::pair<uint32_t, DevInstanceMotion>> moving;  // (instance slot, placements): appended to `normals` at upload
  std::vector<float> normals;
  struct Placed { uint32_t root; uint32_t has_packets; };
  std::unordered_map<const Scene *, Placed> placed;
};

void put_affine(float out[12], const Affine &a) {
  const float v[12] = {a.x.x, a.x.y, a.x.z, a.y.x, a.y.y, a.y.z, a.z.x, a.z.y, a.z.z, a.t.x, a.t.y, a.t.z};
  std::memcpy(out, v, sizeof v);
}
uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Appends `s` (and, first, everything it instances) to the flat arrays; returns its root node index.
Flat::Placed place(Flat &f, const Scene &s) {
  auto it = f.placed.find(&s);
  if (it != f.placed.end()) return it->second;
  const Bvh &b = s.bvh;
  // Inner scenes first, so their roots are known when this scene's instance records are written.
  std::vector<Flat::Placed> inner(b.prims.size(), Flat::Placed{CRT_INVALID_ID, 0});
  for (size_t i = 0; i < b.prims.size(); i++)
    if (b.prims[i].kind == PRIM_INSTANCE) inner[i] = place(f, *b.prims[i].scene);

  const uint32_t node0 = uint32_t(f.nodes.size()), leaf0 = uint32_t(f.leaves.size());
  const uint32_t pkt0 = uint32_t(f.packets.size()), idx0 = uint32_t(f.indices.size());
  const uint32_t prim0 = uint32_t(f.prims.size());
  for (WideNode n : b.wide) {
    for (int l = 0; l < 4; l++) {
      if (!(n.flags & (1u << l))) continue;
      n.child[l] += (n.flags & (1u << (4 + l))) ? leaf0 : node0;
    }
    f.nodes.push_back(n);
  }
  for (Leaf l : b.leaves) {
    l.pkt_first += pkt0;
    l.idx_first += idx0;
    f.leaves.push_back(l);
  }
  for (Tri4 p : b.packets) {
    for (int l = 0; l < 4; l++)
      if (p.active & (1u << l)) p.prim[l] += prim0;
    f.packets.push_back(p);
  }
  // Instance slots in the order of the scalar lists (= the builder's leaf order, spatially coherent): the placements
  // of one leaf sit in consecutive slots, which is what the direct-instance leaf word needs (crt_internal.h).
  std::vector<uint32_t> inst_slot(b.prims.size(), CRT_INVALID_ID);
  {
    uint32_t next_slot = uint32_t(f.instances.size());
    if (read_knobs().inst_order)  // CRT_INST_ORDER=0: primitive order (A/B runs)
    for (uint32_t i : b.indices)
      if (b.prims[i].kind == PRIM_INSTANCE && inst_slot[i] == CRT_INVALID_ID) inst_slot[i] = next_slot++;
    for (size_t i = 0; i < b.prims.size(); i++)  // on no list (cannot happen with this builder): still gets a record
      if (b.prims[i].kind == PRIM_INSTANCE && inst_slot[i] == CRT_INVALID_ID) inst_slot[i] = next_slot++;
    f.instances.resize(next_slot);
  }
  for (size_t i = 0; i < b.prims.size(); i++) {
    const Prim &p = b.prims[i];
    DevPrim d{};
    d.kind = p.kind;
    d.geom_id = p.geom_id;
    d.prim_id = p.prim_id;
    d.mask = p.mask;
    if (p.kind == PRIM_TRI) {
      const float v[9] = {p.v0.x, p.v0.y, p.v0.z, p.v1.x, p.v1.y, p.v1.z, p.v2.x, p.v2.y, p.v2.z};
      std::memcpy(d.d, v, sizeof v);
      uint32_t slot = 0xFFFFFFFFu;
      if (p.has_normals) {
        slot = uint32_t(f.normals.size() / 9);
        const float n[9] = {p.n0.x, p.n0.y, p.n0.z, p.n1.x, p.n1.y, p.n1.z, p.n2.x, p.n2.y, p.n2.z};
        f.normals.insert(f.normals.end(), n, n + 9);
      }
      d.d[9] = u2f(slot);
    } else if (p.kind == PRIM_SPHERE) {
      d.d[0] = p.center.x; d.d[1] = p.center.y; d.d[2] = p.center.z; d.d[3] = p.radius;
    } else {
      DevInstance in{};
      put_affine(in.w2l, p.w2l);
      in.root = inner[i].root;
      in.flags = (inner[i].has_packets ? 1u : 0u) | (p.has_end ? 2u : 0u);
      in.geom_id = p.geom_id;
      in.mask = p.mask;
      d.d[0] = u2f(inst_slot[i]);
      f.instances[inst_slot[i]] = in;
      if (p.has_end) {
        DevInstanceMotion mo{};
        put_affine(mo.l2w, p.l2w);
        put_affine(mo.l2w_end, p.l2w_end);
        f.moving.emplace_back(inst_slot[i], mo);
      }
    }
    f.prims.push_back(d);
  }
  for (uint32_t i : b.indices)  // scalar lists: primitives by index, instances by their record
    f.indices.push_back(b.prims[i].kind == PRIM_INSTANCE ? (kIndexInstance | inst_slot[i]) : (i + prim0));
  Flat::Placed me{b.wide.empty() ? CRT_INVALID_ID : node0, b.packets.empty() ? 0u : 1u};
  f.placed.emplace(&s, me);
  return me;
}
}  // namespace

namespace {
// The device image on the host: everything Scene::ensure_device uploads, plus what crt_scene_image_check verifies.
struct FlatImage {
  Flat f;
  Flat::Placed me{CRT_INVALID_ID, 0};
  uint32_t pool_stack = 6;
  bool direct = false;
  std::vector<uint32_t> leaf_of_word;  // per node lane (4 per node): the Leaf a leaf child word stands for, else ~0
  uint32_t n_staged_roots = 0;
  uint32_t n_top_packets = 0;  // the queried scene's own Tri4 packets: the first ones of the array
  uint32_t cold = kColdAll;    // DevScene::cold
};

// Levels of instancing below `s` (0: no instance anywhere in it).
uint32_t instance_levels(const Scene &s) {
  uint32_t d = 0;
  for (const Prim &p : s.bvh.prims)
    if (p.kind == PRIM_INSTANCE && p.scene) d = std::max(d, 1u + instance_levels(*p.scene));
  return d;
}

int flatten_image(const Scene &scene, FlatImage &im) {
  Flat &f = im.f;
  im.me = place(f, scene);
  Flat::Placed &me = im.me;
  // place() appends a scene BEHIND everything it instances, so the queried scene's packets arrive last. The kernels
  // stage the FIRST packets of the array in LDS behind the node window (stage_nodes): rotate the queried tree's to the
  // front — where rays spend their time when a few instances sit under a real top-level tree (cornellbox) — and the
  // instanced trees' behind them. Packet numbers are opaque to traversal (a leaf names its range), results unchanged.
  {
    const size_t n_top = scene.bvh.packets.size(), n_all = f.packets.size();
    im.n_top_packets = uint32_t(n_top);
    if (n_top && n_top < n_all) {
      std::rotate(f.packets.begin(), f.packets.end() - n_top, f.packets.end());
      for (Leaf &l : f.leaves)
        if (l.pkt_count) l.pkt_first = l.pkt_first >= n_all - n_top ? uint32_t(l.pkt_first - (n_all - n_top)) : uint32_t(l.pkt_first + n_top);
    }
    // ... and among the queried tree's packets, the leaves a ray is most likely to reach first: by the surface area of
    // the leaf's box (the SAH's hit probability), largest first — the walls of a room before the detail inside it. Only
    // the first packets fit the window; a leaf's packets stay consecutive. CRT_HOT_PACKETS=0: builder order (A/B).
    const Knobs knobs = read_knobs();
    if (knobs.hot_packets && n_top > 1) {
      std::vector<std::pair<float, uint32_t>> order;  // (-area, leaf)
      for (const WideNode &n : f.nodes)
        for (int l = 0; l < 4; l++) {
          if (!(n.flags & (1u << l)) || !(n.flags & (1u << (4 + l)))) continue;
          const Leaf &lf = f.leaves[n.child[l]];
          if (!lf.pkt_count || lf.pkt_first >= n_top) continue;
          const float dx = n.bmax[0][l] - n.bmin[0][l], dy = n.bmax[1][l] - n.bmin[1][l], dz = n.bmax[2][l] - n.bmin[2][l];
          order.emplace_back(-((dx * dy + dy * dz) + dz * dx), n.child[l]);
        }
      std::stable_sort(order.begin(), order.end());  // equal areas keep tree order
      std::vector<Tri4> hot;
      hot.reserve(n_top);
      for (const auto &o : order) {
        Leaf &lf = f.leaves[o.second];
        const uint32_t at = uint32_t(hot.size());
        hot.insert(hot.end(), f.packets.begin() + lf.pkt_first, f.packets.begin() + lf.pkt_first + lf.pkt_count);
        lf.pkt_first = at;
      }
      if (hot.size() == n_top) std::copy(hot.begin(), hot.end(), f.packets.begin());
      else return CRT_ERR_BAD_ARG;  // a packet of the queried tree that no leaf names: cannot happen with this builder
    }
  }
  // kernels/traverse_pool.hip.h: the engine's LDS split. Deep stacks pay where rays spend their time inside
  // instances (thousands of placements); a handful of placements under a real top-level tree is still a flat scene.
  // ... and so do large trees: beyond a few thousand nodes a ray's stack regularly passes six entries, and every entry
  // past the LDS part goes to private memory (round 3, Mray/s closest / any hit with 6 + 72 -> 10 + 16: 43 200 triangles
  // in 3 600 nodes 3060 / 4618 -> 3286 / 4878, 7.08 M triangles 1789 / 2489 -> 2075 / 2982, the reference's stress scene
  // through the integrator 2082 -> 2292; 1 728 spheres 4204 / 6618 -> 4200 / 6463: small trees keep the window).
  const bool many_instances = f.instances.size() >= 64;
  const bool large_tree = f.nodes.size() > 2048;
  uint32_t pool_stack = (many_instances || large_tree) ? 10u : 6u;
  // A/B runs: the split, not a number — anything from the deep split's entry count up is the deep one, all else the flat
  const Knobs knobs = read_knobs();
  if (knobs.pool_stack_deep >= 0) pool_stack = knobs.pool_stack_deep ? (uint32_t)CRT_POOL_STACK_DEEP : (uint32_t)CRT_POOL_STACK;
  im.pool_stack = pool_stack;
  const bool deep = pool_stack >= 10u;  // run_traversal's rule for the window size
  // Renumber the nodes for the window the kernels stage in LDS (the first nodes of the array; 72 / 16 / 26 of them
  // by split): the queried scene's root, then the ROOTS of the instanced trees, most placed first — every instance
  // entry tests its tree's root (a city of 40 000 placements has 8 of them and a ray meets five per traversal), so
  // they earn their slots before the second level of the top-level tree does; at most half the smallest window the
  // scene's kernels use — then the top-level tree and the instanced trees, breadth-first. Node numbers are opaque to
  // traversal (children are visited by lane order and distance), so results and visit order do not change.
  if (!f.nodes.empty()) {
    std::vector<uint32_t> new_idx(f.nodes.size(), CRT_INVALID_ID), order;
    order.reserve(f.nodes.size());
    auto number = [&](uint32_t n) {
      if (new_idx[n] != CRT_INVALID_ID) return;
      new_idx[n] = uint32_t(order.size());
      order.push_back(n);
    };
    auto bfs_tree = [&](uint32_t root) {  // numbers the tree under `root` breadth-first (root first, if it is new)
      number(root);
      std::vector<uint32_t> q{root};
      for (size_t h = 0; h < q.size(); h++) {
        const WideNode &n = f.nodes[q[h]];
        for (int l = 0; l < 4; l++)
          if ((n.flags & (1u << l)) && !(n.flags & (1u << (4 + l)))) { number(n.child[l]); q.push_back(n.child[l]); }
      }
    };
    std::vector<std::pair<uint32_t, uint32_t>> inner_roots;  // (placements, root), distinct
    {
      std::unordered_map<uint32_t, uint32_t> count;
      for (const DevInstance &in : f.instances)
        if (in.root != CRT_INVALID_ID) count[in.root]++;
      for (const auto &kv : count) inner_roots.emplace_back(kv.second, kv.first);
      std::sort(inner_roots.begin(), inner_roots.end(), [](const auto &a, const auto &b) {
        return a.first != b.first ? a.first > b.first : a.second < b.second;
      });
    }
    const size_t window = deep ? CRT_POOL_NODES_DEEP : std::min<size_t>(CRT_POOL_NODES, CRT_POOL_NODES_WIDE);
    size_t n_staged_roots = std::min(inner_roots.size(), window / 2);
    if (knobs.stage_roots >= 0) n_staged_roots = std::min(n_staged_roots, (size_t)knobs.stage_roots);  // A/B runs
    im.n_staged_roots = uint32_t(n_staged_roots);
    if (me.root != CRT_INVALID_ID) number(me.root);
    for (size_t k = 0; k < n_staged_roots; k++) number(inner_roots[k].second);
    if (me.root != CRT_INVALID_ID) bfs_tree(me.root);
    for (const auto &r : inner_roots) bfs_tree(r.second);
    std::vector<WideNode> renum(order.size());
    for (size_t k = 0; k < order.size(); k++) {
      WideNode n = f.nodes[order[k]];
      for (int l = 0; l < 4; l++)
        if ((n.flags & (1u << l)) && !(n.flags & (1u << (4 + l)))) n.child[l] = new_idx[n.child[l]];
      renum[k] = n;
    }
    f.nodes.swap(renum);
    for (DevInstance &in : f.instances)
      if (in.root != CRT_INVALID_ID) in.root = new_idx[in.root];
    if (me.root != CRT_INVALID_ID) me.root = new_idx[me.root];
  }
  // Device form of a node's child words: leaf children carry the leaf tag (bit 31), empty lanes are CRT_INVALID_ID,
  // so the traversal derives everything from the word it has to load anyway and never touches `flags`.
  // child words carry node and leaf indices below their tag bits (crt_internal.h)
  if (f.nodes.size() >= (size_t(1) << 31) || f.leaves.size() >= (size_t(1) << 30) || f.indices.size() >= (size_t(1) << 31)) {
    set_error_text("scene image: %zu nodes / %zu leaves / %zu list entries exceed the child word's index range",
                   f.nodes.size(), f.leaves.size(), f.indices.size());
    return CRT_ERR_UNSUPPORTED;
  }
  bool direct = many_instances || f.packets.empty();
  if (knobs.direct_leaves >= 0) direct = knobs.direct_leaves != 0;  // A/B runs
  if (!CRT_DIRECT_LEAVES) direct = false;  // an engine built without the direct form must never meet one
  im.direct = direct;
  bool direct_inst = CRT_DIRECT_INST != 0;
  if (knobs.direct_inst >= 0) direct_inst = direct_inst && knobs.direct_inst != 0;  // A/B runs
  im.leaf_of_word.assign(f.nodes.size() * 4, CRT_INVALID_ID);
  for (size_t ni = 0; ni < f.nodes.size(); ni++) {
    WideNode &n = f.nodes[ni];
    for (int l = 0; l < 4; l++) {
      if (!(n.flags & (1u << l))) n.child[l] = CRT_INVALID_ID;
      else if (n.flags & (1u << (4 + l))) {
        im.leaf_of_word[ni * 4 + l] = n.child[l];
        const Leaf &lf = f.leaves[n.child[l]];
        if (direct && lf.pkt_count == 0 && lf.idx_count >= 1 && lf.idx_count <= 3) {
          // instances in consecutive slots (what place() makes of a leaf of instances): the word names the first slot
          const uint32_t e0 = f.indices[lf.idx_first];
          bool run = direct_inst && (e0 & kIndexInstance) && ((e0 & ~kIndexInstance) + lf.idx_count <= kDirectIndexMask);
          for (uint32_t k = 1; run && k < lf.idx_count; k++) run = f.indices[lf.idx_first + k] == e0 + k;
          if (run) {
            n.child[l] = 0x80000000u | kDirectLeafTag | (lf.idx_count << 28) | kDirectInstTag | (e0 & ~kIndexInstance);
            continue;
          }
          if (lf.idx_first <= kDirectIndexMask) {
            n.child[l] = 0x80000000u | kDirectLeafTag | (lf.idx_count << 28) | lf.idx_first;
            continue;
          }
        }
        n.child[l] |= 0x80000000u;
      }
    }
  }
  // What cold per-ray state a traversal of this image can need (crt_internal.h, kCold*): decided here, where every
  // primitive of every nested scene is in one array. CRT_COLD=7 keeps everything (A/B, tests).
  {
    uint32_t cold = 0;
    for (const DevPrim &d : f.prims) {
      if (d.kind == PRIM_SPHERE) cold |= kColdNormal;
      if (d.kind == PRIM_TRI && f2u(d.d[9]) != 0xFFFFFFFFu) cold |= kColdUV;
    }
    if (instance_levels(scene) > 1) cold |= kColdNormal;
    if (!f.moving.empty()) cold |= kColdTime;
    cold |= knobs.cold;  // CRT_COLD
    im.cold = cold;
  }
  // placements of the moving instances: behind the shading normals, addressed through the instance's flags word
  for (const auto &mv : f.moving) {
    const size_t at = (f.normals.size() + 3) & ~size_t(3);  // 16-byte aligned
    if (at + 24 >= (size_t(1) << 30)) {
      set_error_text("scene image: %zu normal floats leave no room to address a moving instance's placements", f.normals.size());
      return CRT_ERR_UNSUPPORTED;
    }
    f.normals.resize(at + 24);
    std::memcpy(&f.normals[at], &mv.second, sizeof(DevInstanceMotion));
    f.instances[mv.first].flags |= uint32_t(at) << 2;
  }
  return CRT_OK;
}
}  // namespace

// Host-only self-check of the image (no device needed; tests/test_build_parity.py): every child word decodes to exactly
// the node, leaf, scalar list or instance slots it stands for, the numbering is a permutation with the instanced roots
// where the kernels' LDS window expects them, every record index is in range. out: nodes, leaf words in plain / direct
// index / direct instance form, instance records, moving instances, instanced roots staged, 1 if direct leaves are on.
int scene_image_check(const Scene &scene, uint64_t out[8]) {
  FlatImage im;
  int rc = flatten_image(scene, im);
  if (rc != CRT_OK) return rc;
  const Flat &f = im.f;
  auto fail = [&](const char *what, size_t a, size_t b) {
    set_error_text("scene image check: %s (%zu, %zu)", what, a, b);
    return CRT_ERR_BAD_ARG;
  };
  uint64_t n_plain = 0, n_dindex = 0, n_dinst = 0;
  std::vector<uint32_t> parents(f.nodes.size(), 0), leaf_refs(f.leaves.size(), 0);
  for (size_t ni = 0; ni < f.nodes.size(); ni++)
    for (int l = 0; l < 4; l++) {
      const uint32_t w = f.nodes[ni].child[l], lo = im.leaf_of_word[ni * 4 + l];
      const bool valid = (f.nodes[ni].flags >> l) & 1u, leaf = (f.nodes[ni].flags >> (4 + l)) & 1u;
      if (!valid) { if (w != CRT_INVALID_ID) return fail("an empty lane is not the empty word", ni, l); continue; }
      if (!leaf) {
        if ((w & 0x80000000u) || w >= f.nodes.size()) return fail("inner child out of range", ni, w);
        parents[w]++;
        continue;
      }
      if (lo >= f.leaves.size() || !(w & 0x80000000u) || w == CRT_INVALID_ID) return fail("leaf word without its leaf", ni, l);
      leaf_refs[lo]++;
      const Leaf &lf = f.leaves[lo];
      if (!(w & kDirectLeafTag)) {
        if ((w & 0x7fffffffu) != lo) return fail("plain leaf word names another leaf", ni, lo);
        n_plain++;
        continue;
      }
      if (!im.direct || lf.pkt_count != 0) return fail("direct word for a leaf that has packets, or with direct leaves off", ni, lo);
      const uint32_t count = (w >> 28) & 3u;
      if (count != lf.idx_count || count == 0) return fail("direct word's count", count, lf.idx_count);
      for (uint32_t k = 0; k < count; k++) {
        const uint32_t want = f.indices[lf.idx_first + k];
        const uint32_t got = (w & kDirectInstTag) ? (kIndexInstance | ((w & kDirectIndexMask) + k))
                                                  : f.indices[(w & kDirectIndexMask) + k];
        if (got != want) return fail("direct word's entries differ from the leaf's list", got, want);
      }
      (w & kDirectInstTag) ? n_dinst++ : n_dindex++;
    }
  for (size_t k = 0; k < f.leaves.size(); k++)
    if (leaf_refs[k] != 1) return fail("a leaf is not referenced exactly once", k, leaf_refs[k]);
  for (const Leaf &lf : f.leaves)
    if (lf.pkt_count && size_t(lf.pkt_first) + lf.pkt_count > f.packets.size()) return fail("a leaf's packet range", lf.pkt_first, lf.pkt_count);
  if (im.me.root != CRT_INVALID_ID) {  // the queried tree's packets are the first ones of the array (the LDS packet window)
    std::vector<uint32_t> q{im.me.root};
    for (size_t h = 0; h < q.size(); h++)
      for (int l = 0; l < 4; l++) {
        const uint32_t lo = im.leaf_of_word[size_t(q[h]) * 4 + l];
        const bool valid = (f.nodes[q[h]].flags >> l) & 1u, leaf = (f.nodes[q[h]].flags >> (4 + l)) & 1u;
        if (valid && !leaf) q.push_back(f.nodes[q[h]].child[l]);
        else if (valid && f.leaves[lo].pkt_count && f.leaves[lo].pkt_first + f.leaves[lo].pkt_count > im.n_top_packets)
          return fail("a packet of the queried tree lies behind the instanced trees' packets", f.leaves[lo].pkt_first, im.n_top_packets);
      }
  }
  std::vector<uint8_t> is_root(f.nodes.size(), 0);
  if (im.me.root != CRT_INVALID_ID) { if (im.me.root != 0) return fail("the queried scene's root is not node 0", im.me.root, 0); is_root[0] = 1; }
  uint32_t staged = 0;
  for (size_t k = 0; k < f.instances.size(); k++) {
    const DevInstance &in = f.instances[k];
    if (in.root == CRT_INVALID_ID) continue;
    if (in.root >= f.nodes.size()) return fail("instance root out of range", k, in.root);
    if (!is_root[in.root] && in.root >= 1 && in.root <= im.n_staged_roots) staged++;
    is_root[in.root] = 1;
  }
  if (staged != im.n_staged_roots) return fail("instanced roots are not numbered right behind node 0", staged, im.n_staged_roots);
  for (size_t k = 0; k < f.nodes.size(); k++)
    if (parents[k] != (is_root[k] ? 0u : 1u)) return fail("node numbering is not a permutation of the trees", k, parents[k]);
  for (uint32_t e : f.indices) {
    if (e & kIndexInstance) { if ((e & ~kIndexInstance) >= f.instances.size()) return fail("list entry: instance slot", e, f.instances.size()); }
    else if (e >= f.prims.size()) return fail("list entry: primitive", e, f.prims.size());
  }
  uint64_t moving = 0;
  for (const auto &mv : f.moving) {
    const DevInstance &in = f.instances[mv.first];
    const size_t at = in.flags >> 2;
    if (!(in.flags & 2u) || (at & 3u) || at + 24 > f.normals.size() ||
        std::memcmp(&f.normals[at], &mv.second, sizeof(DevInstanceMotion)) != 0)
      return fail("a moving instance's placements are not where its flags word says", mv.first, at);
    moving++;
  }
  for (size_t k = 0; k < f.instances.size(); k++)
    if (!(f.instances[k].flags & 2u) && (f.instances[k].flags >> 2) != 0) return fail("a static instance carries a placement offset", k, 0);
  out[0] = f.nodes.size(); out[1] = n_plain; out[2] = n_dindex; out[3] = n_dinst;
  out[4] = f.instances.size(); out[5] = moving; out[6] = im.n_staged_roots; out[7] = im.direct ? 1 : 0;
  return CRT_OK;
}

namespace {
// The scene view's fields that do not depend on where the image lives (everything but the seven pointers).
void fill_view_fields(const FlatImage &im, DevScene &v) {
  v.root = im.me.root;
  v.has_packets = im.me.has_packets;
  v.n_nodes = uint32_t(im.f.nodes.size());
  v.n_packets = uint32_t(im.f.packets.size());
  v.direct_leaves = im.direct ? 1u : 0u;
  v.pool_stack = im.pool_stack;
  v.cold = im.cold;
}
}  // namespace

// Host-only: which traversal-engine instance select_engine picks for the image this scene would upload (want_wide: -1
// the scene's preference, 0 / 1 asked for), checked against a census of the image's child words and primitives — the
// instance must be able to decode every word and must keep every cold field a primitive of the image can need.
// out: wide | direct engine copy | LDS stack entries | node window | k_extend cold | k_path cold | direct words | DevScene::cold
int scene_engine_select(const Scene &scene, int want_wide, uint32_t out[8]) {
  FlatImage im;
  int rc = flatten_image(scene, im);
  if (rc != CRT_OK) return rc;
  DevScene v{};
  fill_view_fields(im, v);
  EngineSelect e;
  // -2 / -3: as the batched queries' / the renderer's launches decide, CRT_WIDE included; -4: the renderer's own preference
  const bool renderer = want_wide <= -3;
  rc = want_wide <= -2 && want_wide != -4 ? select_engine_env(v, e, renderer) : select_engine(v, want_wide == -4 ? -1 : want_wide, e, renderer);
  if (rc != CRT_OK) {
    set_error_text("select_engine: the %s kernels cannot decode this image (direct child words: %s)",
                   want_wide > 0 ? "four-wave" : "three-wave", v.direct_leaves ? "yes" : "no");
    return rc;
  }
  uint32_t n_direct = 0, need_cold = 0;
  for (const WideNode &n : im.f.nodes)
    for (int l = 0; l < 4; l++) {
      const uint32_t w = n.child[l];
      if (w != CRT_INVALID_ID && (w & 0x80000000u) && (w & kDirectLeafTag)) n_direct++;
    }
  for (const DevPrim &d : im.f.prims) {
    if (d.kind == PRIM_SPHERE) need_cold |= kColdNormal;
    if (d.kind == PRIM_TRI && f2u(d.d[9]) != 0xFFFFFFFFu) need_cold |= kColdUV;
  }
  if (instance_levels(scene) > 1) need_cold |= kColdNormal;
  if (!im.f.moving.empty()) need_cold |= kColdTime;
  auto fail = [&](const char *what) {
    set_error_text("engine select check: %s (wide %d, direct %d, direct words %u, cold %u / kernel %d %d)", what, (int)e.wide,
                   (int)e.direct, n_direct, need_cold, e.ext_cold, e.path_cold);
    return CRT_ERR_BAD_ARG;
  };
  if (n_direct && ((e.wide && !e.wide_direct) || !e.direct)) return fail("the image holds direct child words the selected instance cannot read");
  if (e.wide_direct && !v.direct_leaves) return fail("the direct-engine instances of the four-wave kernels selected for an image without direct words");
  if (n_direct && !v.direct_leaves) return fail("direct words in an image whose view says there are none");
  if ((need_cold & ~(uint32_t)e.ext_cold) || (need_cold & ~(uint32_t)e.path_cold)) return fail("the image needs cold state the selected kernels do not keep");
  if (!engine_accepts(e, v, e.ext_cold) || !engine_accepts(e, v, e.path_cold)) return fail("engine_accepts disagrees with select_engine");
  out[0] = e.wide ? (e.wide_direct ? 2u : 1u) : 0u; out[1] = e.direct; out[2] = e.lds_stack; out[3] = e.window;
  out[4] = (uint32_t)e.ext_cold; out[5] = (uint32_t)e.path_cold; out[6] = n_direct; out[7] = v.cold;
  return CRT_OK;
}

int Scene::ensure_device() {
  std::lock_guard<std::mutex> lock(dev_mu);
  if (dev) return CRT_OK;
  if (!device_ok()) return CRT_ERR_NO_DEVICE;
  FlatImage im;
  const int rc = flatten_image(*this, im);
  if (rc != CRT_OK) return rc;
  Flat &f = im.f;
  auto img = std::make_unique<DeviceImage>();
  constexpr int NA = 7;
  const size_t sz[NA] = {f.nodes.size() * sizeof(WideNode), f.leaves.size() * sizeof(Leaf),
                         f.packets.size() * sizeof(Tri4),   f.indices.size() * sizeof(uint32_t),
                         f.prims.size() * sizeof(DevPrim),  f.instances.size() * sizeof(DevInstance),
                         f.normals.size() * sizeof(float)};
  const void *src[NA] = {f.nodes.data(), f.leaves.data(), f.packets.data(), f.indices.data(),
                         f.prims.data(), f.instances.data(), f.normals.data()};
  size_t off[NA], total = 0;
  for (int i = 0; i < NA; i++) {
    off[i] = total;
    total += (sz[i] + 255) & ~size_t(255);  // every array starts on a 256-byte boundary
    img->bytes[i] = sz[i];
  }
  const size_t err_off = total;
  total += 256;  // the scene's traversal error word
  if (!CRT_HIP_OK(hipMalloc(&img->blob, total))) return CRT_ERR_NO_DEVICE;
  if (!CRT_HIP_OK(hipMemset(static_cast<char *>(img->blob) + err_off, 0, 256))) return CRT_ERR_NO_DEVICE;
  img->err = reinterpret_cast<uint32_t *>(static_cast<char *>(img->blob) + err_off);
  for (int i = 0; i < NA; i++)
    if (sz[i] && !CRT_HIP_OK(hipMemcpy(static_cast<char *>(img->blob) + off[i], src[i], sz[i], hipMemcpyHostToDevice)))
      return CRT_ERR_NO_DEVICE;
  char *base = static_cast<char *>(img->blob);
  img->view.nodes = reinterpret_cast<const WideNode *>(base + off[0]);
  img->view.leaves = reinterpret_cast<const Leaf *>(base + off[1]);
  img->view.packets = reinterpret_cast<const Tri4 *>(base + off[2]);
  img->view.indices = reinterpret_cast<const uint32_t *>(base + off[3]);
  img->view.prims = reinterpret_cast<const DevPrim *>(base + off[4]);
  img->view.instances = reinterpret_cast<const DevInstance *>(base + off[5]);
  img->view.normals = reinterpret_cast<const float *>(base + off[6]);
  fill_view_fields(im, img->view);
  dev = std::move(img);
  return CRT_OK;
}

}  // namespace crt
