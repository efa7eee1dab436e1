// bvh_build.cpp -- see bvh_build.h.  Compiled with -ffp-contract=off: the triangle
// records must be bit-identical to what the parity oracle derives from the same mesh.
#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>

namespace rmclhip {
namespace {

struct Box {
  float mn[3], mx[3];
  void reset() {
    for (int k = 0; k < 3; ++k) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; }
  }
  void grow(const Box& o) {
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], o.mn[k]); mx[k] = std::max(mx[k], o.mx[k]); }
  }
  void grow(const float* p) {
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
  }
  float area() const {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
  }
};

struct Prim { Box b; float c[3]; };

struct Node2 {
  Box b;
  int32_t left = -1, right = -1;  // children (inner)
  uint32_t first = 0, count = 0;  // prims of the whole subtree (a contiguous range of `order`)
  bool leaf() const { return left < 0; }
};

constexpr int kMaxBins = 64;
// SAH bins per axis.  16 is the product's choice; RMCLHIP_BINS (2..64) is a STUDY knob of tools/ (tree quality vs build time), read
// ONCE per process -- never written afterwards, so concurrent builds (rmclhip_*_sharded_create) see one constant
static int bins_from_env() {
  const char* e = std::getenv("RMCLHIP_BINS");
  return e ? std::max(2, std::min(kMaxBins, std::atoi(e))) : 16;
}
static const int kBins = bins_from_env();

struct Builder {
  const std::vector<Prim>& prims;
  std::vector<uint32_t>& order;
  std::vector<Node2> nodes;

  uint32_t max_leaf;

  Builder(const std::vector<Prim>& p, std::vector<uint32_t>& o, uint32_t ml) : prims(p), order(o), max_leaf(ml) {}

  int bin_of(float c, float cmin, float scale) const {
    int b = static_cast<int>((c - cmin) * scale);
    return std::min(kBins - 1, std::max(0, b));
  }

  // iterative top-down build
  void build() {
    struct Task { int32_t node; uint32_t first, count; };
    nodes.reserve(order.size() * 2);
    nodes.emplace_back();
    std::vector<Task> stack;
    stack.push_back({0, 0, static_cast<uint32_t>(order.size())});
    while (!stack.empty()) {
      const Task t = stack.back();
      stack.pop_back();
      Box nb, cb;
      nb.reset(); cb.reset();
      for (uint32_t i = t.first; i < t.first + t.count; ++i) {
        const Prim& p = prims[order[i]];
        nb.grow(p.b);
        cb.grow(p.c);
      }
      nodes[t.node].b = nb;
      nodes[t.node].first = t.first;
      nodes[t.node].count = t.count;
      if (t.count <= max_leaf) continue;
      int best_axis = -1, best_bin = -1;
      float best_cost = FLT_MAX;
      for (int axis = 0; axis < 3; ++axis) {
        const float ext = cb.mx[axis] - cb.mn[axis];
        if (!(ext > 0.f)) continue;
        Box bb[kMaxBins];
        uint32_t bc[kMaxBins];
        for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
        const float scale = static_cast<float>(kBins) / ext;
        for (uint32_t i = t.first; i < t.first + t.count; ++i) {
          const Prim& p = prims[order[i]];
          const int b = bin_of(p.c[axis], cb.mn[axis], scale);
          bc[b]++;
          bb[b].grow(p.b);
        }
        float la[kMaxBins - 1], ra[kMaxBins - 1];
        uint32_t lc[kMaxBins - 1], rc[kMaxBins - 1];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for (int b = 0; b < kBins - 1; ++b) { acc.grow(bb[b]); c += bc[b]; la[b] = acc.area(); lc[b] = c; }
        acc.reset();
        c = 0;
        for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += bc[b]; ra[b - 1] = acc.area(); rc[b - 1] = c; }
        for (int b = 0; b < kBins - 1; ++b) {
          if (lc[b] == 0 || rc[b] == 0) continue;
          const float cost = la[b] * static_cast<float>(lc[b]) + ra[b] * static_cast<float>(rc[b]);
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
      }
      uint32_t mid;
      if (best_axis < 0) {
        mid = t.first + t.count / 2;
      } else {
        const float ext = cb.mx[best_axis] - cb.mn[best_axis];
        const float scale = static_cast<float>(kBins) / ext;
        const float cmin = cb.mn[best_axis];
        auto it = std::partition(order.begin() + t.first, order.begin() + t.first + t.count, [&](uint32_t id) {
          return bin_of(prims[id].c[best_axis], cmin, scale) <= best_bin;
        });
        mid = static_cast<uint32_t>(it - order.begin());
        if (mid == t.first || mid == t.first + t.count) mid = t.first + t.count / 2;
      }
      const int32_t l = static_cast<int32_t>(nodes.size());
      nodes.emplace_back();
      nodes.emplace_back();
      nodes[t.node].left = l;
      nodes[t.node].right = l + 1;
      stack.push_back({l + 1, mid, t.first + t.count - mid});
      stack.push_back({l, t.first, mid - t.first});
    }
  }
};

inline void cross_fma(const float* a, const float* b, float* r) {
  r[0] = std::fmaf(a[1], b[2], -(a[2] * b[1]));
  r[1] = std::fmaf(a[2], b[0], -(a[0] * b[2]));
  r[2] = std::fmaf(a[0], b[1], -(a[1] * b[0]));
}

// collapse of the BVH2 into BVH4 nodes, breadth-first emission.  A BVH2 node whose subtree holds <= leaf_limit
// primitives is a leaf of the emitted tree (the builder splits down to kPfLeafTris, so trees of different leaf sizes are
// cuts through the SAME BVH2 and share the leaf-ordered record array).
struct Collapsed {
  std::vector<Node4> nodes;
  uint32_t max_depth = 0, stack_need = 0;
};

Collapsed collapse(const std::vector<Node2>& n2, uint32_t leaf_limit, float pad) {
  Collapsed out;
  auto is_leaf = [&](int32_t id) { return n2[id].leaf() || n2[id].count <= leaf_limit; };
  struct QItem { int32_t n2; uint32_t depth; uint32_t stack_before; };
  std::deque<QItem> queue;
  // the root is always an inner Node4, even for tiny meshes
  queue.push_back({0, 1, 0});
  std::vector<std::pair<uint32_t, int>> patch;  // (node4 index, slot) -> child n2 id to resolve later

  // first pass: assign Node4 ids in BFS order
  while (!queue.empty()) {
    const QItem it = queue.front();
    queue.pop_front();
    const uint32_t my = static_cast<uint32_t>(out.nodes.size());
    out.nodes.emplace_back();
    Node4& nd = out.nodes.back();
    std::memset(&nd, 0, sizeof(nd));
    out.max_depth = std::max(out.max_depth, it.depth);

    int32_t kids[4];
    int nk = 0;
    if (is_leaf(it.n2)) {
      kids[nk++] = it.n2;
    } else {
      kids[nk++] = n2[it.n2].left;
      kids[nk++] = n2[it.n2].right;
      while (nk < 4) {
        int best = -1;
        float best_area = -1.f;
        for (int i = 0; i < nk; ++i) {
          if (is_leaf(kids[i])) continue;
          const float a = n2[kids[i]].b.area();
          if (a > best_area) { best_area = a; best = i; }
        }
        if (best < 0) break;
        const int32_t k = kids[best];
        kids[best] = n2[k].left;
        kids[nk++] = n2[k].right;
      }
    }
    const uint32_t stack_here = it.stack_before + static_cast<uint32_t>(nk - 1);
    out.stack_need = std::max(out.stack_need, stack_here);
    nd.n_children = static_cast<uint32_t>(nk);
    for (int s = 0; s < 4; ++s) {
      if (s < nk) {
        const Node2& ch = n2[kids[s]];
        nd.x[s] = ch.b.mn[0] - pad; nd.x[4 + s] = ch.b.mx[0] + pad;
        nd.y[s] = ch.b.mn[1] - pad; nd.y[4 + s] = ch.b.mx[1] + pad;
        nd.z[s] = ch.b.mn[2] - pad; nd.z[4 + s] = ch.b.mx[2] + pad;
        if (is_leaf(kids[s])) {
          nd.child[s] = make_leaf_ref(ch.first, ch.count);
        } else {
          // id resolved when the child is dequeued: BFS => ids are assigned in queue order
          nd.child[s] = 0;
          patch.emplace_back(my, s);
          queue.push_back({kids[s], it.depth + 1, stack_here});
        }
      } else {
        // unused slot: unreachable point box + a harmless leaf reference (layout.h)
        nd.x[s] = nd.x[4 + s] = kFarPoint[0];
        nd.y[s] = nd.y[4 + s] = kFarPoint[1];
        nd.z[s] = nd.z[4 + s] = kFarPoint[2];
        nd.child[s] = make_leaf_ref(0, 1);
      }
    }
  }
  // BFS: the i-th pushed inner child receives Node4 id (i+1)
  for (size_t i = 0; i < patch.size(); ++i) out.nodes[patch[i].first].child[patch[i].second] = static_cast<uint32_t>(i + 1);
  out.stack_need += 1;
  return out;
}

// quantised twins (layout.h): per node, corner + step of an 8-bit grid that covers all (padded) child boxes; lower
// planes round down, upper planes up, checked in the arithmetic the decode uses
void quantise(const std::vector<Node4>& nodes, std::vector<Node4Q>& qnodes) {
  qnodes.resize(nodes.size());
  for (size_t i = 0; i < nodes.size(); ++i) {
    const Node4& nd = nodes[i];
    Node4Q& q = qnodes[i];
    std::memset(&q, 0, sizeof(q));
    const float* lohi[3] = {nd.x, nd.y, nd.z};
    uint32_t* qlo[3] = {&q.qx_lo, &q.qy_lo, &q.qz_lo};
    uint32_t* qhi[3] = {&q.qx_hi, &q.qy_hi, &q.qz_hi};
    for (int a = 0; a < 3; ++a) {
      float mn = std::numeric_limits<float>::infinity(), mx = -std::numeric_limits<float>::infinity();
      for (uint32_t c = 0; c < nd.n_children; ++c) { mn = std::min(mn, lohi[a][c]); mx = std::max(mx, lohi[a][4 + c]); }
      float sc = (mx - mn) / 255.0f;
      if (!(sc > 1e-30f)) sc = 1e-30f;
      while (mn + 255.0f * sc < mx) sc = std::nextafter(sc, std::numeric_limits<float>::infinity());
      q.origin[a] = mn;
      q.scale[a] = sc;
      uint32_t lo_word = 0, hi_word = 0;
      for (uint32_t c = 0; c < 4; ++c) {
        uint32_t l = 255u, h = 0u;  // unused slot: inverted
        if (c < nd.n_children) {
          const double dl = (static_cast<double>(lohi[a][c]) - mn) / sc, dh = (static_cast<double>(lohi[a][4 + c]) - mn) / sc;
          int il = static_cast<int>(std::floor(dl)), ih = static_cast<int>(std::ceil(dh));
          il = std::max(0, std::min(255, il));
          ih = std::max(0, std::min(255, ih));
          while (il > 0 && mn + static_cast<float>(il) * sc > lohi[a][c]) --il;
          while (ih < 255 && mn + static_cast<float>(ih) * sc < lohi[a][4 + c]) ++ih;
          l = static_cast<uint32_t>(il);
          h = static_cast<uint32_t>(ih);
        }
        lo_word |= l << (8u * c);
        hi_word |= h << (8u * c);
      }
      *qlo[a] = lo_word;
      *qhi[a] = hi_word;
    }
    for (int c = 0; c < 4; ++c) q.child[c] = nd.child[c];
  }
}

}  // namespace

std::string build_bvh(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, BvhHost& out, uint32_t max_leaf) {
  if (!verts || !faces) return "null mesh pointers";
  if (nf == 0 || nv == 0) return "empty mesh";
  if (nf > 0x0FFFFFFFu) return "too many faces (max 2^28-1)";
  if (max_leaf < 1 || max_leaf > kMaxLeafTris) return "max_leaf must be 1..4";
  for (uint32_t i = 0; i < 3u * nf; ++i)
    if (faces[i] >= nv) return "face index out of range";
  for (size_t i = 0; i < 3 * static_cast<size_t>(nv); ++i)
    if (!std::isfinite(verts[i])) return "non-finite vertex coordinate";

  // triangle records (by face id) + primitive boxes
  std::vector<TriRec> recs(nf);
  std::vector<Prim> prims(nf);
  Box scene;
  scene.reset();
  for (uint32_t f = 0; f < nf; ++f) {
    const float* a = verts + 3 * static_cast<size_t>(faces[3 * f + 0]);
    const float* b = verts + 3 * static_cast<size_t>(faces[3 * f + 1]);
    const float* c = verts + 3 * static_cast<size_t>(faces[3 * f + 2]);
    TriRec& r = recs[f];
    for (int k = 0; k < 3; ++k) {
      r.v0[k] = a[k];
      r.e1[k] = a[k] - b[k];
      r.e2[k] = c[k] - a[k];
    }
    cross_fma(r.e2, r.e1, r.Ng);
    const float d = std::sqrt((r.Ng[0] * r.Ng[0] + r.Ng[1] * r.Ng[1]) + r.Ng[2] * r.Ng[2]);
    for (int k = 0; k < 3; ++k) r.n[k] = (d > 0.f) ? r.Ng[k] / d : 0.f;
    r.face_id = f;
    Prim& p = prims[f];
    p.b.reset();
    p.b.grow(a);
    p.b.grow(b);
    p.b.grow(c);
    for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * (p.b.mn[k] + p.b.mx[k]);
    scene.grow(p.b);
  }

  // ONE BVH2, split down to the smallest leaf size any tree of the map uses
  std::vector<uint32_t> order(nf);
  for (uint32_t f = 0; f < nf; ++f) order[f] = f;
  Builder bld(prims, order, std::min(max_leaf, kPfLeafTris));
  bld.build();
  const std::vector<Node2>& n2 = bld.nodes;

  // conservative padding of every stored box: the slab test runs in fp32 with fused ops and must
  // never cull a box whose triangle the (exact-spec) intersector would accept.
  float diag = 0.f, amax = 0.f;
  for (int k = 0; k < 3; ++k) {
    diag = std::max(diag, scene.mx[k] - scene.mn[k]);
    amax = std::max(amax, std::max(std::fabs(scene.mn[k]), std::fabs(scene.mx[k])));
  }
  const float pad = 1e-4f * std::max(diag, amax) + 1e-6f;

  out.tris.resize(nf);
  for (uint32_t i = 0; i < nf; ++i) out.tris[i] = recs[order[i]];

  // the map's tree: leaves of <= max_leaf records
  Collapsed main_tree = collapse(n2, max_leaf, pad);
  out.nodes = std::move(main_tree.nodes);
  quantise(out.nodes, out.qnodes);

  // the particle filter's tree (quantised nodes only): leaves of <= kPfLeafTris records, same record array.  The
  // filter's rays are incoherent and its kernel is bound by instruction issue with the lanes of a wave taking turns
  // through the triangle loop: shorter leaves trade a few more (cheap, quantised) node steps for fewer loop trips.
  if (max_leaf > kPfLeafTris) {
    Collapsed pf_tree = collapse(n2, kPfLeafTris, pad);
    quantise(pf_tree.nodes, out.qnodes_pf);
    out.info.n_nodes_pf = static_cast<uint32_t>(pf_tree.nodes.size());
    out.info.max_depth_pf = pf_tree.max_depth;
    out.info.stack_need_pf = pf_tree.stack_need;
    out.nodes_pf = std::move(pf_tree.nodes);
  } else {
    out.qnodes_pf = out.qnodes;
    out.nodes_pf = out.nodes;
    out.info.n_nodes_pf = static_cast<uint32_t>(out.nodes.size());
    out.info.max_depth_pf = main_tree.max_depth;
    out.info.stack_need_pf = main_tree.stack_need;
  }

  // frontier of the map's tree (layout.h: kFrontierDepth): every child reference found at BFS depth kFrontierDepth, and every
  // leaf reference above it, with the (padded) box its parent stores for it.  A child's stored box lies inside its parent's,
  // so "the ray hits this entry's box" is the exact condition under which the traversal from the root would reach it.
  auto frontier_of = [](const std::vector<Node4>& tree, std::vector<Node4C::Child>& table) {
    table.clear();
    struct It { uint32_t node; uint32_t depth; };
    std::vector<It> todo{{0u, 0u}};
    while (!todo.empty()) {
      const It it = todo.back();
      todo.pop_back();
      const Node4& nd = tree[it.node];
      for (uint32_t c = 0; c < nd.n_children; ++c) {
        const uint32_t ref = nd.child[c];
        if (!(ref & kLeafBit) && it.depth + 1u < kFrontierDepth) { todo.push_back({ref, it.depth + 1u}); continue; }
        Node4C::Child e;
        e.lo[0] = nd.x[c]; e.lo[1] = nd.y[c]; e.lo[2] = nd.z[c];
        e.hix = nd.x[4 + c]; e.hiy = nd.y[4 + c]; e.hiz = nd.z[4 + c];
        e.ref = ref; e.pad = 0;
        table.push_back(e);
      }
    }
  };
  frontier_of(out.nodes, out.frontier);
  // ... and of the filter's tree: the batch traversal of find (kind 24) walks that tree (round 3: its two-triangle leaves make the
  // per-lane triangle loop 6-10 % cheaper for pose batches), whose node indices differ from the map tree's below the top levels
  frontier_of(out.nodes_pf, out.frontier_pf);

  out.cnodes.resize(out.nodes.size());
  for (size_t i = 0; i < out.nodes.size(); ++i) {
    const Node4& nd = out.nodes[i];
    Node4C& cn = out.cnodes[i];
    for (int c = 0; c < 4; ++c) {
      cn.c[c].lo[0] = nd.x[c]; cn.c[c].lo[1] = nd.y[c]; cn.c[c].lo[2] = nd.z[c];
      cn.c[c].hix = nd.x[4 + c]; cn.c[c].hiy = nd.y[4 + c]; cn.c[c].hiz = nd.z[4 + c];
      cn.c[c].ref = nd.child[c];
      cn.c[c].pad = 0;
    }
  }

  out.info.n_faces = nf;
  out.info.n_vertices = nv;
  out.info.n_nodes = static_cast<uint32_t>(out.nodes.size());
  out.info.max_depth = main_tree.max_depth;
  out.info.stack_need = main_tree.stack_need;
  out.info.pad = pad;
  for (int k = 0; k < 3; ++k) { out.info.bbox_min[k] = scene.mn[k]; out.info.bbox_max[k] = scene.mx[k]; }
  return std::string();
}

}  // namespace rmclhip
