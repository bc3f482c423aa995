// Deterministic SBVH build + BVH4 collapse on the host.
//
// Reproduces, decision for decision, crates/crust-rt/src/bvh.rs:300-327 (Bvh::new), :894-969
// (best_object_split), :974-1054 (best_spatial_split), :1059-1196 (build_subtree), :1224-1237
// (partition_by_bin), :1256-1291 (push_leaf), :1299-1375 (collapse), with triangle.rs:367-429
// (clip_triangle_aabb) and aabb.rs:47-58 (triangle_aabb). Leaf and lane order are load-bearing: ties in
// traversal resolve by them (bvh.rs:537-544), so the tree must match the reference's, not merely be good.
//
// Structure: subtrees are built as self-contained vectors with local indices and spliced under their
// parent (the reference's Subtree/merge, bvh.rs:294-297, :846-872); subtrees above PARALLEL_THRESHOLD
// references build on separate threads (bvh.rs:1156-1162). Threads change when, never what.
//
// Compiled with -ffp-contract=off: split costs and clipped bounds are compared exactly.
#include <cmath>
#include <cstring>
#include <atomic>
#include <future>
#include <system_error>
#include <thread>
#include <limits>
#include <optional>

#include "crt_internal.h"

namespace crt {

namespace {

constexpr size_t kMaxDepth = 60;           // bvh.rs:142
constexpr size_t kMinLeaf = 2;             // bvh.rs:144
constexpr size_t kMinLeafPacked = 4;       // bvh.rs:153
constexpr size_t kMaxLeaf = 8;             // bvh.rs:156
constexpr int kBins = 12;                  // bvh.rs:158
constexpr size_t kParallelThreshold = 4096;  // bvh.rs:160
constexpr float kSbvhAlpha = 1e-5f;        // bvh.rs:164
constexpr size_t kSbvhMaxDepth = 32;       // bvh.rs:167
constexpr uint32_t kEmptyLane = 0xFFFFFFFFu;
constexpr float kInf = std::numeric_limits<float>::infinity();

struct PrimRef { Aabb bbox; uint32_t idx; };
struct BNode { Aabb bbox; uint32_t first_or_right; uint32_t count; };
struct Subtree { std::vector<BNode> nodes; std::vector<uint32_t> indices; };

inline Aabb unite(const Aabb &a, const Aabb &b) { return {vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }
inline F3 centroid(const PrimRef &r) { return (r.bbox.mn + r.bbox.mx) * 0.5f; }
inline float surface_area(const Aabb &b) {
  F3 d = b.mx - b.mn;
  return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
inline float rmax(float a, float b) { return (a > b || b != b) ? a : b; }  // f32::max
inline float rmin(float a, float b) { return (a < b || b != b) ? a : b; }  // f32::min

Aabb union_all(const std::vector<PrimRef> &refs) {
  Aabb acc = refs[0].bbox;
  for (size_t i = 1; i < refs.size(); i++) acc = unite(acc, refs[i].bbox);
  return acc;
}

std::optional<Aabb> overlap_of(const Aabb &a, const Aabb &b) {  // bvh.rs:823-831
  F3 lo = vmax(a.mn, b.mn), hi = vmin(a.mx, b.mx);
  if (lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z) return Aabb{lo, hi};
  return std::nullopt;
}

void pad_flat_axes(F3 &lo, F3 &hi) {
  constexpr float PAD = 1e-4f;
  for (int a = 0; a < 3; a++) {
    if (hi[a] - lo[a] < PAD) { lo.at(a) -= PAD; hi.at(a) += PAD; }
  }
}

Aabb triangle_aabb(F3 v0, F3 v1, F3 v2) {  // aabb.rs:47-58
  F3 lo = vmin(vmin(v0, v1), v2), hi = vmax(vmax(v0, v1), v2);
  pad_flat_axes(lo, hi);
  return {lo, hi};
}

// Sutherland-Hodgman against the slab's two planes (triangle.rs:367-429).
std::optional<Aabb> clip_triangle(F3 v0, F3 v1, F3 v2, int axis, float mn, float mx) {
  F3 poly[8] = {v0, v1, v2};
  int n = 3;
  for (int pass = 0; pass < 2; pass++) {
    const float bound = pass == 0 ? mn : mx;
    const bool keep_ge = pass == 0;
    F3 out[8];
    int m = 0;
    for (int i = 0; i < n; i++) {
      F3 a = poly[i], b = poly[(i + 1) % n];
      float da = keep_ge ? a[axis] - bound : bound - a[axis];
      float db = keep_ge ? b[axis] - bound : bound - b[axis];
      if (da >= 0.0f) out[m++] = a;
      if ((da > 0.0f) != (db > 0.0f) && da != db) {
        float t = da / (da - db);
        out[m++] = a + (b - a) * t;
      }
    }
    std::memcpy(poly, out, sizeof out);
    n = m;
    if (n == 0) return std::nullopt;
  }
  F3 lo = poly[0], hi = poly[0];
  for (int i = 1; i < n; i++) { lo = vmin(lo, poly[i]); hi = vmax(hi, poly[i]); }
  lo.at(axis) = rmax(lo[axis], mn);
  hi.at(axis) = rmin(hi[axis], mx);
  pad_flat_axes(lo, hi);
  return Aabb{lo, hi};
}

std::optional<Aabb> clipped_aabb(const Prim &p, int axis, float mn, float mx) {
  if (p.kind == PRIM_TRI) return clip_triangle(p.v0, p.v1, p.v2, axis, mn, mx);  // prim.rs:116-118
  Aabb b = prim_bbox(p);                                                         // prim.rs:39-48
  if (b.mn[axis] > mx || b.mx[axis] < mn) return std::nullopt;
  b.mn.at(axis) = rmax(b.mn[axis], mn);
  b.mx.at(axis) = rmin(b.mx[axis], mx);
  return b;
}

// Rust `f32 as usize`: truncation, saturating, NaN -> 0.
inline size_t f32_as_usize(float x) {
  if (!(x > 0.0f)) return 0;
  if (x >= 1.8446744e19f) return ~size_t(0);
  return static_cast<size_t>(x);
}

struct ObjSplit { int axis; float cmin, scale; int split_bin; float cost; Aabb lb, rb; };
struct SpatSplit { int axis; float pos, cost; };

inline int object_bin(const PrimRef &r, int axis, float cmin, float scale) {  // bvh.rs:916 / :1225
  size_t b = f32_as_usize((centroid(r)[axis] - cmin) * scale);
  return static_cast<int>(b < size_t(kBins - 1) ? b : size_t(kBins - 1));
}

inline int longest_axis(F3 e) { return (e.x >= e.y && e.x >= e.z) ? 0 : (e.y >= e.z ? 1 : 2); }

// Prefix/suffix sweep over the bins exactly as written (bvh.rs:932-966): each candidate re-accumulates
// its sides from scratch in bin order, so the union association is ((b0 u b1) u b2)...
template <class Count>
void sweep(const Count *left_counts, const Count *right_counts, const std::optional<Aabb> *bounds, int split,
           size_t &lc, size_t &rc, std::optional<Aabb> &lb, std::optional<Aabb> &rb) {
  lc = rc = 0;
  lb.reset();
  rb.reset();
  for (int b = 0; b <= split; b++) {
    lc += left_counts[b];
    if (bounds[b]) lb = lb ? unite(*lb, *bounds[b]) : *bounds[b];
  }
  for (int b = split + 1; b < kBins; b++) {
    rc += right_counts[b];
    if (bounds[b]) rb = rb ? unite(*rb, *bounds[b]) : *bounds[b];
  }
}

std::optional<ObjSplit> best_object_split(const std::vector<PrimRef> &refs) {  // bvh.rs:894-969
  F3 cmin = centroid(refs[0]), cmax = cmin;
  for (size_t i = 1; i < refs.size(); i++) {
    F3 c = centroid(refs[i]);
    cmin = vmin(cmin, c);
    cmax = vmax(cmax, c);
  }
  F3 extent = cmax - cmin;
  const int axis = longest_axis(extent);
  if (extent[axis] <= 1e-6f) return std::nullopt;
  const float scale = float(kBins) / extent[axis];
  size_t counts[kBins] = {};
  std::optional<Aabb> bounds[kBins];
  for (const PrimRef &r : refs) {
    int b = object_bin(r, axis, cmin[axis], scale);
    counts[b]++;
    bounds[b] = bounds[b] ? unite(*bounds[b], r.bbox) : r.bbox;
  }
  std::optional<ObjSplit> best;
  for (int split = 0; split < kBins - 1; split++) {
    size_t lc, rc;
    std::optional<Aabb> lb, rb;
    sweep(counts, counts, bounds, split, lc, rc, lb, rb);
    if (lc == 0 || rc == 0) continue;
    float cost = surface_area(*lb) * float(lc) + surface_area(*rb) * float(rc);
    if (!best || cost < best->cost) best = ObjSplit{axis, cmin[axis], scale, split, cost, *lb, *rb};
  }
  return best;
}

std::optional<SpatSplit> best_spatial_split(const std::vector<Prim> &prims, const std::vector<PrimRef> &refs,
                                            const Aabb &bbox) {  // bvh.rs:974-1054
  F3 extent = bbox.mx - bbox.mn;
  const int axis = longest_axis(extent);
  if (extent[axis] <= 1e-6f) return std::nullopt;
  const float lo = bbox.mn[axis];
  const float width = extent[axis] / float(kBins);
  auto bin_of = [&](float x) {
    size_t b = f32_as_usize((x - lo) / width);
    return static_cast<int>(b > size_t(kBins - 1) ? size_t(kBins - 1) : b);
  };
  size_t entry[kBins] = {}, exit_[kBins] = {};
  std::optional<Aabb> bounds[kBins];
  auto add = [&](int b, const Aabb &box) { bounds[b] = bounds[b] ? unite(*bounds[b], box) : box; };
  for (const PrimRef &r : refs) {
    int b0 = bin_of(r.bbox.mn[axis]), b1 = bin_of(r.bbox.mx[axis]);
    entry[b0]++;
    exit_[b1]++;
    if (b0 == b1) { add(b0, r.bbox); continue; }
    for (int b = b0; b <= b1; b++) {
      float bin_lo = lo + float(b) * width, bin_hi = lo + float(b + 1) * width;
      if (auto c = clipped_aabb(prims[r.idx], axis, bin_lo, bin_hi))
        if (auto ci = overlap_of(*c, r.bbox)) add(b, *ci);
    }
  }
  std::optional<SpatSplit> best;
  for (int split = 0; split < kBins - 1; split++) {
    size_t lc, rc;
    std::optional<Aabb> lb, rb;
    sweep(entry, exit_, bounds, split, lc, rc, lb, rb);
    if (lc == 0 || rc == 0) continue;
    if (!lb || !rb) std::abort();  // the reference `expect`s here (bvh.rs:1043-1044)
    float cost = surface_area(*lb) * float(lc) + surface_area(*rb) * float(rc);
    if (!best || cost < best->cost) best = SpatSplit{axis, lo + float(split + 1) * width, cost};
  }
  return best;
}

size_t min_leaf_for(const std::vector<Prim> &prims, const std::vector<PrimRef> &refs) {  // bvh.rs:1201-1210
  for (const PrimRef &r : refs)
    if (prims[r.idx].kind != PRIM_TRI) return kMinLeaf;
  return kMinLeafPacked;
}

Subtree make_leaf(const Aabb &bbox, const std::vector<PrimRef> &refs) {  // bvh.rs:833-842
  Subtree s;
  s.nodes.push_back(BNode{bbox, 0, uint32_t(refs.size())});
  s.indices.reserve(refs.size());
  for (const PrimRef &r : refs) s.indices.push_back(r.idx);
  return s;
}

Subtree merge(const Aabb &bbox, Subtree &&left, Subtree &&right) {  // bvh.rs:846-872
  Subtree out;
  out.nodes.reserve(1 + left.nodes.size() + right.nodes.size());
  const uint32_t right_offset = 1 + uint32_t(left.nodes.size());
  out.nodes.push_back(BNode{bbox, right_offset, 0});
  for (BNode n : left.nodes) {
    if (n.count == 0) n.first_or_right += 1;
    out.nodes.push_back(n);
  }
  const uint32_t leaf_offset = uint32_t(left.indices.size());
  for (BNode n : right.nodes) {
    n.first_or_right += (n.count == 0) ? right_offset : leaf_offset;
    out.nodes.push_back(n);
  }
  out.indices = std::move(left.indices);
  out.indices.insert(out.indices.end(), right.indices.begin(), right.indices.end());
  return out;
}

void partition_by_bin(std::vector<PrimRef> &&refs, const ObjSplit &o, std::vector<PrimRef> &left,
                      std::vector<PrimRef> &right) {  // bvh.rs:1224-1237
  size_t n_left = 0;
  for (const PrimRef &r : refs) n_left += object_bin(r, o.axis, o.cmin, o.scale) <= o.split_bin;
  left.reserve(n_left);
  right.reserve(refs.size() - n_left);
  for (const PrimRef &r : refs) (object_bin(r, o.axis, o.cmin, o.scale) <= o.split_bin ? left : right).push_back(r);
}

Subtree build_subtree(const std::vector<Prim> &prims, std::vector<PrimRef> &&refs, size_t depth, float root_area);

// The reference forks with rayon::join on a bounded pool (bvh.rs:1156-1162). Here a fork takes one of a fixed
// number of helper-thread tokens (hardware threads - 1) or, when none is free or the thread cannot be created,
// builds both sides inline: at most that many OS threads are ever alive, however large the input. The tree is the
// same either way (the build is deterministic, bvh.rs:22-24; tests/test_build_parity.py).
std::atomic<int> g_build_helpers{0};

int max_build_helpers() {
  static const int n = [] {
    unsigned hc = std::thread::hardware_concurrency();
    return int(hc > 1 ? (hc > 64 ? 64 : hc) - 1 : 0);
  }();
  return n;
}

Subtree build_children(const std::vector<Prim> &prims, const Aabb &bbox, std::vector<PrimRef> &&l,
                       std::vector<PrimRef> &&r, size_t depth, float root_area, bool allow_parallel) {
  if (allow_parallel && std::max(l.size(), r.size()) > kParallelThreshold) {
    bool token = g_build_helpers.fetch_add(1) < max_build_helpers();
    if (!token) g_build_helpers.fetch_sub(1);
    if (token) {
      std::future<Subtree> fut;
      bool spawned = true;
      try {
        fut = std::async(std::launch::async, [&prims, &l, depth, root_area]() {
          // the token goes back on every way out of the helper, an exception (bad_alloc -> crt_commit returns NULL)
          // included: a failed commit must not shrink the pool for the rest of the process
          struct TokenReturn { ~TokenReturn() { g_build_helpers.fetch_sub(1); } } token_return;
          return build_subtree(prims, std::move(l), depth + 1, root_area);
        });
      } catch (const std::system_error &) {  // thread limit reached: nothing was moved from, build inline
        g_build_helpers.fetch_sub(1);
        spawned = false;
      }
      if (spawned) {
        Subtree right = build_subtree(prims, std::move(r), depth + 1, root_area);
        Subtree left = fut.get();
        return merge(bbox, std::move(left), std::move(right));
      }
    }
  }
  Subtree left = build_subtree(prims, std::move(l), depth + 1, root_area);
  Subtree right = build_subtree(prims, std::move(r), depth + 1, root_area);
  return merge(bbox, std::move(left), std::move(right));
}

Subtree object_partition_or_leaf(const std::vector<Prim> &prims, std::vector<PrimRef> &&refs, const Aabb &bbox,
                                 const std::optional<ObjSplit> &object, size_t depth, float root_area) {
  // bvh.rs:1173-1196
  if (object && refs.size() > min_leaf_for(prims, refs)) {
    std::vector<PrimRef> l, r;
    partition_by_bin(std::move(refs), *object, l, r);
    if (l.empty() || r.empty()) {
      l.insert(l.end(), r.begin(), r.end());
      return make_leaf(bbox, l);
    }
    return build_children(prims, bbox, std::move(l), std::move(r), depth, root_area, false);
  }
  return make_leaf(bbox, refs);
}

Subtree build_subtree(const std::vector<Prim> &prims, std::vector<PrimRef> &&refs, size_t depth, float root_area) {
  // bvh.rs:1059-1169
  const Aabb bbox = union_all(refs);
  const size_t count = refs.size();
  if (count <= min_leaf_for(prims, refs) || depth >= kMaxDepth) return make_leaf(bbox, refs);

  const std::optional<ObjSplit> object = best_object_split(refs);
  std::optional<SpatSplit> spatial;
  if (object && depth < kSbvhMaxDepth) {
    auto ov = overlap_of(object->lb, object->rb);
    float overlap = ov ? surface_area(*ov) : 0.0f;
    if (overlap / root_area > kSbvhAlpha) {
      auto s = best_spatial_split(prims, refs, bbox);
      if (s && s->cost < object->cost) spatial = s;
    }
  }

  std::vector<PrimRef> left, right;
  if (spatial) {
    const SpatSplit &s = *spatial;
    for (const PrimRef &r : refs) {
      if (r.bbox.mx[s.axis] <= s.pos) left.push_back(r);
      else if (r.bbox.mn[s.axis] >= s.pos) right.push_back(r);
      else {
        const Prim &prim = prims[r.idx];
        if (auto c = clipped_aabb(prim, s.axis, -kInf, s.pos))
          if (auto ci = overlap_of(*c, r.bbox)) left.push_back(PrimRef{*ci, r.idx});
        if (auto c = clipped_aabb(prim, s.axis, s.pos, kInf))
          if (auto ci = overlap_of(*c, r.bbox)) right.push_back(PrimRef{*ci, r.idx});
      }
    }
    if (left.empty() || right.empty()) {
      left.insert(left.end(), right.begin(), right.end());
      return object_partition_or_leaf(prims, std::move(left), bbox, object, depth, root_area);
    }
  } else if (object) {
    if (count <= kMaxLeaf && object->cost >= surface_area(bbox) * float(count)) return make_leaf(bbox, refs);
    partition_by_bin(std::move(refs), *object, left, right);
  } else {
    const size_t mid = count / 2;  // median split by input order (bvh.rs:1148-1152)
    left.assign(refs.begin(), refs.begin() + mid);
    right.assign(refs.begin() + mid, refs.end());
  }
  return build_children(prims, bbox, std::move(left), std::move(right), depth, root_area, true);
}

// ---- collapse to BVH4 (bvh.rs:1299-1375) ----
struct Collapsed {
  std::vector<WideNode> wide;
  std::vector<Leaf> leaves;
  std::vector<Tri4> packets;
  std::vector<uint32_t> indices;
};

WideNode empty_node() {  // bvh.rs:230-241
  WideNode w;
  std::memset(&w, 0, sizeof w);
  for (int a = 0; a < 3; a++)
    for (int l = 0; l < 4; l++) w.bmin[a][l] = w.bmax[a][l] = kInf;
  for (int l = 0; l < 4; l++) w.child[l] = kEmptyLane;
  return w;
}

void set_lane(WideNode &w, int lane, const Aabb &b) {  // bvh.rs:243-251
  for (int a = 0; a < 3; a++) { w.bmin[a][lane] = b.mn[a]; w.bmax[a][lane] = b.mx[a]; }
  w.flags |= 1u << lane;
}

Tri4 pack(const Prim *const *tris, const uint32_t *pi, int n) {  // triangle.rs:217-253
  Tri4 p;
  std::memset(&p, 0, sizeof p);
  p.mask_and = 0xFFFFFFFFu;
  for (int lane = 0; lane < 4; lane++) {
    const int src = lane < n - 1 ? lane : n - 1;  // tail lanes duplicate the last real triangle
    const Prim &t = *tris[src];
    for (int a = 0; a < 3; a++) {
      p.v[0][a][lane] = t.v0[a];
      p.v[1][a][lane] = t.v1[a];
      p.v[2][a][lane] = t.v2[a];
    }
    p.prim[lane] = 0xFFFFFFFFu;
    if (lane < n) {
      p.prim[lane] = pi[lane];
      p.masks[lane] = t.mask;
      p.active |= 1u << lane;
      p.mask_and &= t.mask;
      p.mask_or |= t.mask;
      F3 nrm = cross(t.v1 - t.v0, t.v2 - t.v0);
      if (t.has_normals || !(nrm.x == 0.0f && nrm.y == 0.0f && nrm.z == 0.0f)) p.normal_ok |= 1u << lane;
    }
  }
  return p;
}

uint32_t push_leaf(Collapsed &c, const uint32_t *range, size_t n, const std::vector<Prim> &prims) {
  // bvh.rs:1256-1290
  Leaf lf{uint32_t(c.packets.size()), 0, uint32_t(c.indices.size()), 0};
  const Prim *batch[4];
  uint32_t batch_pi[4];
  int nb = 0;
  for (size_t i = 0; i < n; i++) {
    const uint32_t pi = range[i];
    if (prims[pi].kind == PRIM_TRI) {
      batch[nb] = &prims[pi];
      batch_pi[nb++] = pi;
      if (nb == 4) { c.packets.push_back(pack(batch, batch_pi, 4)); nb = 0; }
    } else {
      c.indices.push_back(pi);
      lf.idx_count++;
    }
  }
  if (nb > 0) c.packets.push_back(pack(batch, batch_pi, nb));
  lf.pkt_count = uint32_t(c.packets.size()) - lf.pkt_first;
  c.leaves.push_back(lf);
  return uint32_t(c.leaves.size()) - 1;
}

uint32_t collapse_node(Collapsed &c, const Subtree &bin, const std::vector<Prim> &prims, uint32_t b_idx) {
  // bvh.rs:1328-1375
  const size_t slot = c.wide.size();
  c.wide.push_back(empty_node());
  uint32_t kids[4] = {b_idx + 1, bin.nodes[b_idx].first_or_right, 0, 0};
  int n_kids = 2;
  while (n_kids < 4) {
    int best = -1;
    float best_area = 0.0f;
    for (int i = 0; i < n_kids; i++) {
      const BNode &k = bin.nodes[kids[i]];
      if (k.count == 0) {
        float a = surface_area(k.bbox);
        if (best < 0 || a > best_area) { best = i; best_area = a; }  // ties resolve to the first
      }
    }
    if (best < 0) break;
    const uint32_t k = kids[best];
    kids[best] = k + 1;
    kids[n_kids++] = bin.nodes[k].first_or_right;
  }
  for (int lane = 0; lane < n_kids; lane++) {
    const BNode &k = bin.nodes[kids[lane]];
    set_lane(c.wide[slot], lane, k.bbox);
    if (k.count > 0) {
      uint32_t li = push_leaf(c, bin.indices.data() + k.first_or_right, k.count, prims);
      c.wide[slot].child[lane] = li;
      c.wide[slot].flags |= 1u << (4 + lane);
    } else {
      uint32_t ci = collapse_node(c, bin, prims, kids[lane]);
      c.wide[slot].child[lane] = ci;
    }
  }
  return uint32_t(slot);
}

}  // namespace

Aabb prim_bbox(const Prim &p) {
  switch (p.kind) {
    case PRIM_TRI: return triangle_aabb(p.v0, p.v1, p.v2);  // prim.rs:112-114
    case PRIM_SPHERE: {                                      // prim.rs:163-168
      F3 r = f3(p.radius, p.radius, p.radius);
      return {p.center - r, p.center + r};
    }
    default: return p.bounds;  // prim.rs:380-382
  }
}

void build_bvh(Bvh &out, std::vector<Prim> &&prims) {  // bvh.rs:300-327
  out = Bvh{};
  out.prims = std::move(prims);
  if (out.prims.empty()) return;
  std::vector<PrimRef> refs(out.prims.size());
  for (size_t i = 0; i < out.prims.size(); i++) refs[i] = PrimRef{prim_bbox(out.prims[i]), uint32_t(i)};
  const Aabb root = union_all(refs);
  Subtree tree = build_subtree(out.prims, std::move(refs), 0, surface_area(root));
  Collapsed c;
  c.wide.reserve(tree.nodes.size() / 2 + 1);
  if (tree.nodes[0].count > 0) {  // single-leaf tree (bvh.rs:1309-1317)
    WideNode w = empty_node();
    set_lane(w, 0, tree.nodes[0].bbox);
    w.child[0] = push_leaf(c, tree.indices.data() + tree.nodes[0].first_or_right, tree.nodes[0].count, out.prims);
    w.flags |= 1u << 4;
    c.wide.push_back(w);
  } else {
    collapse_node(c, tree, out.prims, 0);
  }
  out.wide = std::move(c.wide);
  out.leaves = std::move(c.leaves);
  out.packets = std::move(c.packets);
  out.indices = std::move(c.indices);
  out.has_bbox = true;
  out.root_bbox = root;
}

}  // namespace crt
