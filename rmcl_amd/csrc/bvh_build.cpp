// bvh_build.cpp -- see bvh_build.h.  Compiled with -ffp-contract=off: the triangle
// records must be bit-identical to what the parity oracle derives from the same mesh.
#include "bvh_build.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <exception>
#include <system_error>
#include <thread>

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
  int32_t left = -1, right = -1;  // children (inner); a child's index is always larger than its parent's
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

// Height budget of the BVH2 (round 5; VERDICT r4 "never refuse a valid mesh").  A node at depth d may only carry a subtree of
// height <= kMaxHeight2 - d; a SAH split that would leave a child with more primitives than a balanced subtree of the remaining
// height can hold is replaced by an object-median split (which halves the count, so the budget holds by induction).  With the
// height-first collapse below, a BVH2 of height h becomes a BVH4 whose traversal stack needs <= 3 * ceil(h / 2) + 1 entries:
// 42 -> 64 = kStackEntries of the kernels (traverse.hip.h), for ANY mesh of <= 2^28 - 1 faces.
constexpr uint32_t kMaxHeight2 = 42;
constexpr uint32_t kStackEntries = 64;

// threads of one build: the CPUs this process may run on (affinity mask, cgroup quota), at most 16; RMCLHIP_BUILD_THREADS overrides.
// The tree does not depend on it: every split is a function of the SET of primitives of its node, and partitions are stable.
static int build_threads() {
  if (const char* e = std::getenv("RMCLHIP_BUILD_THREADS")) return std::max(1, std::min(64, std::atoi(e)));
  int n = static_cast<int>(std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1 << 20, CPU_COUNT(&set));
  if (FILE* fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = 0, period = 0;
    if (std::fscanf(fh, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
      n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    std::fclose(fh);
  }
  return std::max(1, std::min(16, n));
}

// RMCLHIP_BUILD_TRACE=1: phase times of every build on stderr (tools/, never set by the product)
struct PhaseTimer {
  bool on = std::getenv("RMCLHIP_BUILD_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[bvh_build] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// f(thread, begin, end) over [0, n) in `nt` contiguous chunks (chunk k belongs to thread k: deterministic ownership)
// Nothing may escape a std::thread or leave joinable threads behind (either ends the process, and this code runs under extern-"C"
// entry points): a worker hands its exception over, a thread that cannot be created (EAGAIN under a pid limit) has its chunk run by the
// caller, and everything started is joined before the first exception continues to build_bvh, which turns it into an error string.
template <class F>
static void parallel_chunks(size_t n, int nt, F&& f) {
  if (nt <= 1 || n < 2) { f(0, size_t{0}, n); return; }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  std::vector<std::exception_ptr> err(static_cast<size_t>(nt));
  auto run = [&](int k) noexcept {
    try { f(k, n * k / nt, n * (k + 1) / nt); } catch (...) { err[static_cast<size_t>(k)] = std::current_exception(); }
  };
  int started = 1;
  for (int k = 1; k < nt; ++k) {
    try { th.emplace_back(run, k); } catch (const std::system_error&) { break; }
    ++started;
  }
  run(0);
  for (int k = started; k < nt; ++k) run(k);   // chunks whose thread could not be created
  for (auto& t : th) t.join();
  for (const auto& e : err) if (e) std::rethrow_exception(e);
}

struct Bins {
  uint32_t cnt[3][kMaxBins];
  Box box[3][kMaxBins];   // primitive boxes per bin
  Box cen[3][kMaxBins];   // centroid boxes per bin (the children's centroid bounds come out of the same pass)
  void reset(int nb) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < nb; ++b) { cnt[a][b] = 0; box[a][b].reset(); cen[a][b].reset(); }
  }
  void merge(const Bins& o, int nb) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < nb; ++b) { cnt[a][b] += o.cnt[a][b]; box[a][b].grow(o.box[a][b]); cen[a][b].grow(o.cen[a][b]); }
  }
};

struct Task {
  int32_t node;
  uint32_t first, count, depth;
  bool have_bounds;
  Box nb, cb;   // bounds of the primitives / of their centroids, when the parent's bins already hold them
};

// Binned-SAH top-down builder.  One pass bins a node's primitives on all three axes at once (boxes AND centroid boxes per bin, so the
// children's bounds need no pass of their own), one stable partition moves them: `order` stays sorted by face id inside every node,
// whatever the number of threads.  Large nodes are processed one at a time with the passes spread over the threads; subtrees below
// `grain` primitives are handed out whole.
struct Builder {
  const std::vector<Prim>& prims;
  std::vector<uint32_t>& order;
  std::vector<Node2> nodes;
  uint32_t max_leaf;
  int nthreads;
  uint32_t height_fallbacks = 0;   // median splits forced by the height budget (0 on every mesh of the test-suite's benchmarks)

  Builder(const std::vector<Prim>& p, std::vector<uint32_t>& o, uint32_t ml, int nt) : prims(p), order(o), max_leaf(ml), nthreads(nt) {}

  static int bin_of(float c, float cmin, float scale) {
    int b = static_cast<int>((c - cmin) * scale);
    return std::min(kBins - 1, std::max(0, b));
  }
  // primitives a subtree of height h can hold
  uint64_t cap(uint32_t h) const { return h >= 40 ? ~uint64_t{0} : static_cast<uint64_t>(max_leaf) << h; }

  void bounds_of(uint32_t first, uint32_t count, Box& nb, Box& cb, int nt) const {
    std::vector<Box> pn(nt), pc(nt);
    parallel_chunks(count, nt, [&](int k, size_t a, size_t b) {
      Box n, c;
      n.reset(); c.reset();
      for (size_t i = first + a; i < first + b; ++i) { const Prim& p = prims[order[i]]; n.grow(p.b); c.grow(p.c); }
      pn[k] = n; pc[k] = c;
    });
    nb = pn[0]; cb = pc[0];
    for (int k = 1; k < nt; ++k) { nb.grow(pn[k]); cb.grow(pc[k]); }
  }

  // splits one node: fills nodes[t.node], partitions order[first, first+count), returns false for a leaf, else the two child tasks
  // (their node indices are NOT assigned here).  `tmp` is scratch of >= count entries, `nt` the threads this call may use.
  bool split(const Task& t_in, Task& lt, Task& rt, uint32_t* tmp, int nt, Node2& node, uint32_t& fallbacks) {
    Task t = t_in;
    if (!t.have_bounds) bounds_of(t.first, t.count, t.nb, t.cb, nt);
    node.b = t.nb;
    node.first = t.first;
    node.count = t.count;
    if (t.count <= max_leaf) return false;
    const Box& cb = t.cb;
    float scale[3];
    bool use[3];
    for (int a = 0; a < 3; ++a) {
      const float ext = cb.mx[a] - cb.mn[a];
      use[a] = ext > 0.f;
      scale[a] = use[a] ? static_cast<float>(kBins) / ext : 0.f;
    }
    int best_axis = -1, best_bin = -1;
    Bins bins;
    if (use[0] || use[1] || use[2]) {
      std::vector<Bins> part(nt > 1 ? nt : 0);
      bins.reset(kBins);
      auto bin_range = [&](Bins& B, size_t a, size_t b) {
        for (size_t i = t.first + a; i < t.first + b; ++i) {
          const Prim& p = prims[order[i]];
          for (int ax = 0; ax < 3; ++ax) {
            if (!use[ax]) continue;
            const int k = bin_of(p.c[ax], cb.mn[ax], scale[ax]);
            B.cnt[ax][k]++;
            B.box[ax][k].grow(p.b);
            B.cen[ax][k].grow(p.c);
          }
        }
      };
      if (nt > 1) {
        parallel_chunks(t.count, nt, [&](int k, size_t a, size_t b) { part[k].reset(kBins); bin_range(part[k], a, b); });
        for (int k = 0; k < nt; ++k) bins.merge(part[k], kBins);
      } else {
        bin_range(bins, 0, t.count);
      }
      float best_cost = FLT_MAX;
      for (int axis = 0; axis < 3; ++axis) {
        if (!use[axis]) continue;
        float la[kMaxBins - 1], ra[kMaxBins - 1];
        uint32_t lc[kMaxBins - 1], rc[kMaxBins - 1];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for (int b = 0; b < kBins - 1; ++b) { acc.grow(bins.box[axis][b]); c += bins.cnt[axis][b]; la[b] = acc.area(); lc[b] = c; }
        acc.reset();
        c = 0;
        for (int b = kBins - 1; b > 0; --b) { acc.grow(bins.box[axis][b]); c += bins.cnt[axis][b]; ra[b - 1] = acc.area(); rc[b - 1] = c; }
        for (int b = 0; b < kBins - 1; ++b) {
          if (lc[b] == 0 || rc[b] == 0) continue;
          const float cost = la[b] * static_cast<float>(lc[b]) + ra[b] * static_cast<float>(rc[b]);
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
      }
    }
    const uint32_t height_left = kMaxHeight2 - t.depth;   // >= 1 here: count > max_leaf and count <= cap(height_left)
    uint32_t nl = 0;
    if (best_axis >= 0) {
      for (int b = 0; b <= best_bin; ++b) nl += bins.cnt[best_axis][b];
      const uint64_t child_cap = cap(height_left - 1);
      if (nl > child_cap || t.count - nl > child_cap) { best_axis = -1; nl = 0; ++fallbacks; }   // height budget: median instead
      else if (nl == 0 || nl == t.count) { best_axis = -1; nl = 0; }
    }
    lt.have_bounds = rt.have_bounds = false;
    if (best_axis >= 0) {
      const float cmin = cb.mn[best_axis], sc = scale[best_axis];
      const int ax = best_axis, bb = best_bin;
      uint32_t* o = order.data() + t.first;
      if (nt > 1) {
        std::vector<uint32_t> lcount(nt + 1, 0);
        parallel_chunks(t.count, nt, [&](int k, size_t a, size_t b) {
          uint32_t c = 0;
          for (size_t i = a; i < b; ++i) c += bin_of(prims[o[i]].c[ax], cmin, sc) <= bb;
          lcount[k + 1] = c;
        });
        for (int k = 0; k < nt; ++k) lcount[k + 1] += lcount[k];
        parallel_chunks(t.count, nt, [&](int k, size_t a, size_t b) {
          uint32_t li = lcount[k], ri = nl + static_cast<uint32_t>(a) - lcount[k];
          for (size_t i = a; i < b; ++i) {
            const uint32_t id = o[i];
            if (bin_of(prims[id].c[ax], cmin, sc) <= bb) tmp[li++] = id; else tmp[ri++] = id;
          }
        });
        parallel_chunks(t.count, nt, [&](int, size_t a, size_t b) { std::memcpy(o + a, tmp + a, (b - a) * sizeof(uint32_t)); });
      } else {
        uint32_t li = 0, ri = 0;
        for (uint32_t i = 0; i < t.count; ++i) {
          const uint32_t id = o[i];
          if (bin_of(prims[id].c[ax], cmin, sc) <= bb) o[li++] = id; else tmp[ri++] = id;
        }
        std::memcpy(o + li, tmp, ri * sizeof(uint32_t));
      }
      lt.nb.reset(); lt.cb.reset(); rt.nb.reset(); rt.cb.reset();
      for (int b = 0; b < kBins; ++b) {
        Task& side = b <= best_bin ? lt : rt;
        if (bins.cnt[ax][b] == 0) continue;
        side.nb.grow(bins.box[ax][b]);
        side.cb.grow(bins.cen[ax][b]);
      }
      lt.have_bounds = rt.have_bounds = true;
    } else {
      // no SAH split (all centroids coincide: halve the face-id order) or the height budget refused it (object median on the
      // widest centroid axis, ties by face id: the set of the lower half is unique, so the split does not depend on the input order)
      nl = t.count - t.count / 2;
      int ax = 0;
      float ext = -1.f;
      for (int a = 0; a < 3; ++a) if (cb.mx[a] - cb.mn[a] > ext) { ext = cb.mx[a] - cb.mn[a]; ax = a; }
      if (ext > 0.f) {
        uint32_t* o = order.data() + t.first;
        auto less = [&](uint32_t x, uint32_t y) {
          const float cx = prims[x].c[ax], cy = prims[y].c[ax];
          return cx < cy || (cx == cy && x < y);
        };
        std::nth_element(o, o + nl, o + t.count, less);
        std::sort(o, o + nl);               // back to face-id order inside each half
        std::sort(o + nl, o + t.count);
      } else {
        nl = t.count / 2;                    // (the split rule of rounds 1-4 for coincident centroids)
      }
    }
    lt.first = t.first; lt.count = nl; lt.depth = t.depth + 1;
    rt.first = t.first + nl; rt.count = t.count - nl; rt.depth = t.depth + 1;
    return true;
  }

  void build() {
    const uint32_t n = static_cast<uint32_t>(order.size());
    const uint32_t grain = nthreads > 1 ? std::max<uint32_t>(4096u, n / (static_cast<uint32_t>(nthreads) * 16u)) : 0xFFFFFFFFu;
    std::vector<uint32_t> tmp(nthreads > 1 ? n : 0);
    nodes.clear();
    nodes.emplace_back();
    Task root{};
    root.node = 0; root.first = 0; root.count = n; root.depth = 0; root.have_bounds = false;
    PhaseTimer pa;
    // phase A: nodes above the grain, one at a time, every pass spread over the threads
    std::vector<Task> big{root}, small;
    if (n <= grain) { small.swap(big); }
    while (!big.empty()) {
      const Task t = big.back();
      big.pop_back();
      Task lt, rt;
      Node2 nd;
      if (!split(t, lt, rt, tmp.data(), nthreads, nd, height_fallbacks)) { nodes[t.node] = nd; continue; }
      const int32_t l = static_cast<int32_t>(nodes.size());
      nd.left = l; nd.right = l + 1;
      nodes[t.node] = nd;
      nodes.emplace_back();
      nodes.emplace_back();
      lt.node = l; rt.node = l + 1;
      (rt.count > grain ? big : small).push_back(rt);
      (lt.count > grain ? big : small).push_back(lt);
    }
    pa.mark("  phase A (large nodes)");
    PhaseTimer pt;
    // phase B: whole subtrees, one thread each, into local node arrays (local index 0 = the subtree's root, which already has a slot)
    std::sort(small.begin(), small.end(), [](const Task& a, const Task& b) { return a.count != b.count ? a.count > b.count : a.first < b.first; });
    std::vector<std::vector<Node2>> local(small.size());
    std::vector<uint32_t> fb(small.size(), 0);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
      std::vector<uint32_t> scratch;
      std::vector<Task> stack;
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= small.size()) break;
        std::vector<Node2>& ln = local[k];
        ln.reserve(2 * static_cast<size_t>(small[k].count) / std::max(1u, max_leaf) + 2);
        if (scratch.size() < small[k].count) scratch.resize(small[k].count);
        ln.emplace_back();
        Task r = small[k];
        r.node = 0;
        stack.assign(1, r);
        while (!stack.empty()) {
          const Task t = stack.back();
          stack.pop_back();
          Task lt, rt;
          Node2 nd;
          if (!split(t, lt, rt, scratch.data(), 1, nd, fb[k])) { ln[t.node] = nd; continue; }
          const int32_t l = static_cast<int32_t>(ln.size());
          nd.left = l; nd.right = l + 1;
          ln[t.node] = nd;
          ln.emplace_back();
          ln.emplace_back();
          lt.node = l; rt.node = l + 1;
          stack.push_back(rt);
          stack.push_back(lt);
        }
      }
    };
    // (the workers pull subtrees from a shared counter: any number of them, down to the caller alone, completes the list)
    parallel_chunks(static_cast<size_t>(std::max<size_t>(std::min<size_t>(nthreads, small.size()), 1)),
                    static_cast<int>(std::min<size_t>(nthreads, small.size())), [&](int, size_t, size_t) { worker(); });
    pt.mark("  phase B (subtrees)");
    // splice: subtree k's local nodes 1.. go to [off[k], ...) of the global array
    std::vector<size_t> off(small.size() + 1, nodes.size());
    for (size_t k = 0; k < small.size(); ++k) off[k + 1] = off[k] + local[k].size() - 1;
    nodes.resize(off[small.size()]);
    parallel_chunks(small.size(), nthreads, [&](int, size_t a, size_t b) {
      for (size_t k = a; k < b; ++k) {
        const std::vector<Node2>& ln = local[k];
        const int32_t shift = static_cast<int32_t>(off[k]) - 1;
        for (size_t i = 0; i < ln.size(); ++i) {
          Node2 nd = ln[i];
          if (!nd.leaf()) { nd.left += shift; nd.right += shift; }
          nodes[i == 0 ? static_cast<size_t>(small[k].node) : off[k] + i - 1] = nd;
        }
      }
    });
    for (uint32_t f : fb) height_fallbacks += f;
    pt.mark("  splice");
  }
};

// RMCLHIP_SBVH_ALPHA (study knob, read once per process): extra references the spatial splits may create, as a multiple of the face
// count; 0 = object splits only (the builder of rounds 1-5)
static double sbvh_alpha_from_env() {
  const char* e = std::getenv("RMCLHIP_SBVH_ALPHA");
  return e ? std::max(0.0, std::min(64.0, std::atof(e))) : 1.0;
}
static double sbvh_alpha() {
  static const double a = sbvh_alpha_from_env();
  return a;
}
constexpr uint32_t kSbvhMaxFaces = 4000000u;   // larger maps: the in-place builder, object splits only
constexpr int64_t kSbvhFloor = 0;              // (an absolute floor of extra references for small maps was tried: a 20 k sliver fan took 37 x its records for 1.8 x; RMCLHIP_SBVH_ALPHA is the knob for such maps)

// ---------------------------------------------------------------------------------------------
// SBVH (round 6): the same top-down builder with SPATIAL splits (Stich, Friedrich, Dietrich: "Spatial splits in bounding volume
// hierarchies", HPG 2009).  An object split puts every primitive whole into one child: triangles that are long against their
// neighbours (CAD walls among scanned detail, slivers) give children whose boxes overlap, and a ray through the overlap walks both.
// A spatial split cuts the node's box by a plane and REFERENCES a straddling triangle from both children, each with the box of its
// part on that side: tighter boxes for more references.  Per node:
//   * the object split of Builder::split, decision for decision (same bins over the centroid bounds, same cost loop, same fallbacks),
//     so that a node which takes no spatial split has the children the plain builder gives it;
//   * if that split's children overlap by more than kSbvhOverlap of the root's area: kBins spatial bins per axis over the NODE's box;
//     every reference is chopped into the bins it spans (the triangle clipped to the bin's slab, intersected with the reference's
//     own box), entering the bin it starts in and leaving the bin it ends in; the cheapest plane by the same SAH;
//   * the spatial split is taken when it is cheaper than kSbvhGain x the object split, fits the node's share of the reference budget
//     and makes progress; a straddling reference is then split, or -- "unsplitting" -- put whole into one child when that is cheaper.
// Reference budget: alpha x n_faces extra references for the whole tree (at least kSbvhFloor: small maps may afford what their worst
// triangles need), handed down the tree in proportion to the children's reference counts, so that the result does not depend on the
// order in which subtrees are built (threads).  Leaves reference RECORDS: a face that is referenced k times has k identical 64-B
// records (the kernels' tie rule -- min t, then min face id -- makes duplicates invisible).
// ---------------------------------------------------------------------------------------------
struct Ref { Box b; float c[3]; uint32_t face; };

constexpr float kSbvhOverlap = 1.0e-5f;   // Stich et al.'s alpha: spatial splits only where the object split's children overlap noticeably
constexpr float kSbvhGain = 0.90f;        // ... and only when they are at least 10 % cheaper: regular meshes keep the plain builder's tree

struct SplitTask {
  std::vector<Ref> refs;
  int32_t node = 0;
  uint32_t depth = 0;
  int64_t budget = 0;       // extra references this subtree may still create
  bool have_bounds = false;
  Box nb, cb;
};

struct SplitBuilder {
  const float* verts;
  const uint32_t* faces;
  uint32_t max_leaf;
  int nthreads;
  float root_area = 0.f;
  std::vector<Node2> nodes;                       // leaves: first = index into `leaves` until finish() assigns the record ranges
  std::vector<std::vector<uint32_t>> leaves;      // faces of every leaf, in face-id order
  uint32_t height_fallbacks = 0;
  uint64_t spatial_splits = 0, duplicates = 0;

  SplitBuilder(const float* v, const uint32_t* f, uint32_t ml, int nt) : verts(v), faces(f), max_leaf(ml), nthreads(nt) {}

  static int bin_of(float c, float cmin, float scale) {
    int b = static_cast<int>((c - cmin) * scale);
    return std::min(kBins - 1, std::max(0, b));
  }
  uint64_t cap(uint32_t h) const { return h >= 40 ? ~uint64_t{0} : static_cast<uint64_t>(max_leaf) << h; }

  // bounds of the part of triangle `face` between the planes x_axis = lo and x_axis = hi (Sutherland-Hodgman on the two planes),
  // intersected with `within`; empty (mn > mx) when nothing of the triangle lies there
  Box clipped(uint32_t face, int axis, float lo, float hi, const Box& within) const {
    float poly[8][3], tmp[8][3];
    int n = 3;
    for (int k = 0; k < 3; ++k) {
      const float* p = verts + 3 * static_cast<size_t>(faces[3 * static_cast<size_t>(face) + k]);
      poly[k][0] = p[0]; poly[k][1] = p[1]; poly[k][2] = p[2];
    }
    for (int side = 0; side < 2 && n > 0; ++side) {
      const float plane = side == 0 ? lo : hi;
      const float sgn = side == 0 ? 1.f : -1.f;      // keep sgn * (x - plane) >= 0
      int m = 0;
      for (int i = 0; i < n; ++i) {
        const float* a = poly[i];
        const float* b = poly[(i + 1) % n];
        const float da = sgn * (a[axis] - plane), db = sgn * (b[axis] - plane);
        if (da >= 0.f) { tmp[m][0] = a[0]; tmp[m][1] = a[1]; tmp[m][2] = a[2]; ++m; }
        if ((da > 0.f && db < 0.f) || (da < 0.f && db > 0.f)) {
          const float t = da / (da - db);
          for (int k = 0; k < 3; ++k) tmp[m][k] = a[k] + t * (b[k] - a[k]);
          tmp[m][axis] = plane;
          ++m;
        }
      }
      n = m;
      for (int i = 0; i < n; ++i) { poly[i][0] = tmp[i][0]; poly[i][1] = tmp[i][1]; poly[i][2] = tmp[i][2]; }
    }
    Box r;
    r.reset();
    for (int i = 0; i < n; ++i) r.grow(poly[i]);
    if (n > 0) {
      // the interpolated vertices are rounded: widen by a few units in the last place of the coordinates involved before the
      // intersection (the stored boxes get the scene's padding on top, bvh_build_impl); the cut itself stays exact
      for (int k = 0; k < 3; ++k) {
        if (k == axis) continue;
        const float e = 4.f * FLT_EPSILON * std::max(std::fabs(r.mn[k]), std::fabs(r.mx[k]));
        r.mn[k] -= e; r.mx[k] += e;
      }
      for (int k = 0; k < 3; ++k) { r.mn[k] = std::max(r.mn[k], within.mn[k]); r.mx[k] = std::min(r.mx[k], within.mx[k]); }
      r.mn[axis] = std::max(r.mn[axis], std::max(lo, within.mn[axis]));
      r.mx[axis] = std::min(r.mx[axis], std::min(hi, within.mx[axis]));
    }
    return r;
  }
  static bool empty(const Box& b) { return b.mn[0] > b.mx[0] || b.mn[1] > b.mx[1] || b.mn[2] > b.mx[2]; }
  static void set_centre(Ref& r) { for (int k = 0; k < 3; ++k) r.c[k] = 0.5f * (r.b.mn[k] + r.b.mx[k]); }

  struct Leaf { std::vector<uint32_t> faces; };

  // splits one node: fills `node` (box), returns false for a leaf (its faces in leaf_out), else fills lt / rt (refs, bounds, budgets)
  bool split(SplitTask& t, SplitTask& lt, SplitTask& rt, int nt, Node2& node, std::vector<uint32_t>& leaf_out, uint32_t& fallbacks,
             uint64_t& n_spatial, uint64_t& n_dup) {
    std::vector<Ref>& R = t.refs;
    const uint32_t count = static_cast<uint32_t>(R.size());
    if (!t.have_bounds) {
      t.nb.reset(); t.cb.reset();
      for (const Ref& r : R) { t.nb.grow(r.b); t.cb.grow(r.c); }
    }
    node.b = t.nb;
    node.count = count;
    if (count <= max_leaf) {
      leaf_out.resize(count);
      for (uint32_t i = 0; i < count; ++i) leaf_out[i] = R[i].face;
      return false;
    }
    // ---- the object split (Builder::split) ----
    const Box& cb = t.cb;
    float scale[3];
    bool use[3];
    for (int a = 0; a < 3; ++a) {
      const float ext = cb.mx[a] - cb.mn[a];
      use[a] = ext > 0.f;
      scale[a] = use[a] ? static_cast<float>(kBins) / ext : 0.f;
    }
    int best_axis = -1, best_bin = -1;
    float best_cost = FLT_MAX;
    Bins bins;
    if (use[0] || use[1] || use[2]) {
      bins.reset(kBins);
      auto bin_range = [&](Bins& B, size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
          const Ref& p = R[i];
          for (int ax = 0; ax < 3; ++ax) {
            if (!use[ax]) continue;
            const int k = bin_of(p.c[ax], cb.mn[ax], scale[ax]);
            B.cnt[ax][k]++;
            B.box[ax][k].grow(p.b);
            B.cen[ax][k].grow(p.c);
          }
        }
      };
      if (nt > 1) {
        std::vector<Bins> part(nt);
        parallel_chunks(count, nt, [&](int k, size_t a, size_t b) { part[k].reset(kBins); bin_range(part[k], a, b); });
        for (int k = 0; k < nt; ++k) bins.merge(part[k], kBins);
      } else {
        bin_range(bins, 0, count);
      }
      for (int axis = 0; axis < 3; ++axis) {
        if (!use[axis]) continue;
        float la[kMaxBins - 1], ra[kMaxBins - 1];
        uint32_t lc[kMaxBins - 1], rc[kMaxBins - 1];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for (int b = 0; b < kBins - 1; ++b) { acc.grow(bins.box[axis][b]); c += bins.cnt[axis][b]; la[b] = acc.area(); lc[b] = c; }
        acc.reset();
        c = 0;
        for (int b = kBins - 1; b > 0; --b) { acc.grow(bins.box[axis][b]); c += bins.cnt[axis][b]; ra[b - 1] = acc.area(); rc[b - 1] = c; }
        for (int b = 0; b < kBins - 1; ++b) {
          if (lc[b] == 0 || rc[b] == 0) continue;
          const float cost = la[b] * static_cast<float>(lc[b]) + ra[b] * static_cast<float>(rc[b]);
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
      }
    }
    const uint32_t height_left = kMaxHeight2 - t.depth;
    const uint64_t child_cap = cap(height_left - 1);
    uint32_t nl = 0;
    bool height_refused = false;
    if (best_axis >= 0) {
      for (int b = 0; b <= best_bin; ++b) nl += bins.cnt[best_axis][b];
      if (nl > child_cap || count - nl > child_cap) { best_axis = -1; nl = 0; ++fallbacks; height_refused = true; }
      else if (nl == 0 || nl == count) { best_axis = -1; nl = 0; }
    }
    // ---- a spatial split, where the object split's children overlap (or there is no object split at all) ----
    bool spatial = false;
    int s_axis = -1;
    float s_plane = 0.f;
    if (!height_refused && t.budget > 0 && root_area > 0.f) {
      float overlap = FLT_MAX;
      if (best_axis >= 0) {
        Box lb, rb;
        lb.reset(); rb.reset();
        for (int b = 0; b < kBins; ++b) { if (bins.cnt[best_axis][b]) (b <= best_bin ? lb : rb).grow(bins.box[best_axis][b]); }
        Box ov;
        for (int k = 0; k < 3; ++k) { ov.mn[k] = std::max(lb.mn[k], rb.mn[k]); ov.mx[k] = std::min(lb.mx[k], rb.mx[k]); }
        overlap = ov.area();
      }
      if (overlap > kSbvhOverlap * root_area) {
        float sp_cost = FLT_MAX;
        uint32_t sp_nl = 0, sp_nr = 0;
        for (int axis = 0; axis < 3; ++axis) {
          const float lo = t.nb.mn[axis], ext = t.nb.mx[axis] - lo;
          if (!(ext > 0.f)) continue;
          const float sc = static_cast<float>(kBins) / ext, width = ext / static_cast<float>(kBins);
          struct SBins { Box bb[kMaxBins]; uint32_t en[kMaxBins], ex[kMaxBins]; };
          auto sbin_range = [&](SBins& S, size_t i0, size_t i1) {
            for (int b = 0; b < kBins; ++b) { S.bb[b].reset(); S.en[b] = S.ex[b] = 0; }
            for (size_t i = i0; i < i1; ++i) {
              const Ref& r = R[i];
              const int b0 = bin_of(r.b.mn[axis], lo, sc), b1 = bin_of(r.b.mx[axis], lo, sc);
              S.en[b0]++; S.ex[b1]++;
              if (b0 == b1) { S.bb[b0].grow(r.b); continue; }
              for (int b = b0; b <= b1; ++b) {
                const float p0 = (b == 0) ? -FLT_MAX : lo + width * static_cast<float>(b);
                const float p1 = (b == kBins - 1) ? FLT_MAX : lo + width * static_cast<float>(b + 1);
                const Box part = clipped(r.face, axis, p0, p1, r.b);
                if (!empty(part)) S.bb[b].grow(part);
              }
            }
          };
          SBins tot;
          if (nt > 1) {
            std::vector<SBins> part(nt);
            parallel_chunks(count, nt, [&](int k, size_t a, size_t b) { sbin_range(part[k], a, b); });
            tot = part[0];
            for (int k = 1; k < nt; ++k)
              for (int b = 0; b < kBins; ++b) { tot.bb[b].grow(part[k].bb[b]); tot.en[b] += part[k].en[b]; tot.ex[b] += part[k].ex[b]; }
          } else {
            sbin_range(tot, 0, count);
          }
          const Box* bb = tot.bb;
          const uint32_t* en = tot.en;
          const uint32_t* ex = tot.ex;
          float la[kMaxBins - 1], ra[kMaxBins - 1];
          uint32_t lc[kMaxBins - 1], rc[kMaxBins - 1];
          Box acc;
          acc.reset();
          uint32_t c = 0;
          for (int b = 0; b < kBins - 1; ++b) { acc.grow(bb[b]); c += en[b]; la[b] = acc.area(); lc[b] = c; }
          acc.reset();
          c = 0;
          for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += ex[b]; ra[b - 1] = acc.area(); rc[b - 1] = c; }
          for (int b = 0; b < kBins - 1; ++b) {
            if (lc[b] == 0 || rc[b] == 0) continue;
            const float cost = la[b] * static_cast<float>(lc[b]) + ra[b] * static_cast<float>(rc[b]);
            if (cost < sp_cost) { sp_cost = cost; s_axis = axis; s_plane = lo + width * static_cast<float>(b + 1); sp_nl = lc[b]; sp_nr = rc[b]; }
          }
        }
        const uint64_t dup = (s_axis >= 0 && static_cast<uint64_t>(sp_nl) + sp_nr > count) ? static_cast<uint64_t>(sp_nl) + sp_nr - count : 0u;
        // (a child may be as large as its parent -- every reference straddling the plane: what chops a bundle of slivers --; the split
        // then costs `count` references of the budget, which is what ends such a chain)
        spatial = s_axis >= 0 && (best_axis < 0 || sp_cost < kSbvhGain * best_cost) && static_cast<int64_t>(dup) <= t.budget &&
                  (dup > 0u || (sp_nl < count && sp_nr < count)) && sp_nl <= child_cap && sp_nr <= child_cap;
      }
    }
    lt.have_bounds = rt.have_bounds = false;
    lt.depth = rt.depth = t.depth + 1;
    int64_t budget_left = t.budget;
    if (spatial) {
      // ---- perform it: references wholly on one side go there; a straddler is split -- or kept whole in one child when that is cheaper
      Box lb, rb;
      lb.reset(); rb.reset();
      uint32_t n1 = 0, n2 = 0;
      std::vector<uint8_t> where(count);   // 0 left, 1 right, 2 straddles
      for (uint32_t i = 0; i < count; ++i) {
        const Ref& r = R[i];
        if (r.b.mx[s_axis] <= s_plane) { where[i] = 0; lb.grow(r.b); ++n1; }
        else if (r.b.mn[s_axis] >= s_plane) { where[i] = 1; rb.grow(r.b); ++n2; }
        else { where[i] = 2; ++n1; ++n2; }
      }
      // bounds of the halves of the straddlers first (their boxes enter both sides' bounds unless unsplit below)
      std::vector<Box> lpart(count), rpart(count);
      for (uint32_t i = 0; i < count; ++i) {
        if (where[i] != 2) continue;
        lpart[i] = clipped(R[i].face, s_axis, -FLT_MAX, s_plane, R[i].b);
        rpart[i] = clipped(R[i].face, s_axis, s_plane, FLT_MAX, R[i].b);
        if (empty(lpart[i]) && empty(rpart[i])) { lpart[i] = R[i].b; lpart[i].mx[s_axis] = s_plane; rpart[i] = R[i].b; rpart[i].mn[s_axis] = s_plane; }
        if (empty(lpart[i])) { where[i] = 1; --n1; rb.grow(R[i].b); continue; }
        if (empty(rpart[i])) { where[i] = 0; --n2; lb.grow(R[i].b); continue; }
        lb.grow(lpart[i]); rb.grow(rpart[i]);
      }
      lt.refs.clear(); rt.refs.clear();
      lt.refs.reserve(n1); rt.refs.reserve(n2);
      for (uint32_t i = 0; i < count; ++i) {
        const Ref& r = R[i];
        if (where[i] == 0) { lt.refs.push_back(r); continue; }
        if (where[i] == 1) { rt.refs.push_back(r); continue; }
        // unsplitting (Stich et al., section 4.4), in list order: whole left / whole right / split
        Box lu = lb, ru = rb;
        lu.grow(r.b); ru.grow(r.b);
        const float c_split = lb.area() * static_cast<float>(n1) + rb.area() * static_cast<float>(n2);
        const float c_left = lu.area() * static_cast<float>(n1) + rb.area() * static_cast<float>(n2 - 1u);
        const float c_right = lb.area() * static_cast<float>(n1 - 1u) + ru.area() * static_cast<float>(n2);
        if (c_left < c_split && c_left <= c_right && n2 > 1u) { lb = lu; --n2; lt.refs.push_back(r); continue; }
        if (c_right < c_split && n1 > 1u) { rb = ru; --n1; rt.refs.push_back(r); continue; }
        Ref a = r, b = r;
        a.b = lpart[i]; b.b = rpart[i];
        set_centre(a); set_centre(b);
        lt.refs.push_back(a); rt.refs.push_back(b);
      }
      const uint32_t a1 = static_cast<uint32_t>(lt.refs.size()), a2 = static_cast<uint32_t>(rt.refs.size());
      const uint64_t dup_now = static_cast<uint64_t>(a1) + a2 > count ? static_cast<uint64_t>(a1) + a2 - count : 0u;
      if (a1 == 0 || a2 == 0 || (dup_now == 0u && (a1 >= count || a2 >= count)) || static_cast<int64_t>(dup_now) > t.budget) {
        spatial = false;   // no progress (unsplitting emptied a side): the object split instead
      } else {
        const uint64_t dup = static_cast<uint64_t>(a1) + a2 > count ? static_cast<uint64_t>(a1) + a2 - count : 0u;
        budget_left -= static_cast<int64_t>(dup);
        n_dup += dup;
        ++n_spatial;
      }
    }
    if (!spatial && best_axis >= 0) {
      const float cmin = cb.mn[best_axis], sc = scale[best_axis];
      lt.refs.clear(); rt.refs.clear();
      lt.refs.reserve(nl); rt.refs.reserve(count - nl);
      for (const Ref& r : R) (bin_of(r.c[best_axis], cmin, sc) <= best_bin ? lt.refs : rt.refs).push_back(r);
      lt.nb.reset(); lt.cb.reset(); rt.nb.reset(); rt.cb.reset();
      for (int b = 0; b < kBins; ++b) {
        SplitTask& side = b <= best_bin ? lt : rt;
        if (bins.cnt[best_axis][b] == 0) continue;
        side.nb.grow(bins.box[best_axis][b]);
        side.cb.grow(bins.cen[best_axis][b]);
      }
      lt.have_bounds = rt.have_bounds = true;
    } else if (!spatial) {
      // no SAH split at all, or the height budget refused it: object median (Builder::split's rule)
      nl = count - count / 2;
      int ax = 0;
      float ext = -1.f;
      for (int a = 0; a < 3; ++a) if (cb.mx[a] - cb.mn[a] > ext) { ext = cb.mx[a] - cb.mn[a]; ax = a; }
      if (ext > 0.f) {
        auto less = [&](const Ref& x, const Ref& y) { return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.face < y.face); };
        std::nth_element(R.begin(), R.begin() + nl, R.end(), less);
        auto by_face = [](const Ref& x, const Ref& y) { return x.face < y.face; };
        std::sort(R.begin(), R.begin() + nl, by_face);
        std::sort(R.begin() + nl, R.end(), by_face);
      } else {
        nl = count / 2;
      }
      lt.refs.assign(R.begin(), R.begin() + nl);
      rt.refs.assign(R.begin() + nl, R.end());
    }
    std::vector<Ref>().swap(R);
    // the rest of the budget goes down in proportion to the children's reference counts
    {
      const uint64_t a1 = lt.refs.size(), a2 = rt.refs.size();
      const int64_t bl = (a1 + a2) ? static_cast<int64_t>(static_cast<unsigned __int128>(static_cast<uint64_t>(std::max<int64_t>(budget_left, 0))) * a1 / (a1 + a2)) : 0;
      lt.budget = bl;
      rt.budget = std::max<int64_t>(budget_left, 0) - bl;
    }
    return true;
  }

  void build(std::vector<Ref>&& refs, int64_t budget) {
    const uint32_t n = static_cast<uint32_t>(refs.size());
    const uint32_t grain = nthreads > 1 ? std::max<uint32_t>(4096u, n / (static_cast<uint32_t>(nthreads) * 16u)) : 0xFFFFFFFFu;
    nodes.clear();
    nodes.emplace_back();
    leaves.clear();
    {
      Box nb;
      nb.reset();
      for (const Ref& r : refs) nb.grow(r.b);
      root_area = nb.area();
    }
    std::vector<SplitTask> big, small;
    {
      SplitTask root;
      root.refs = std::move(refs);
      root.node = 0; root.depth = 0; root.budget = budget;
      (n > grain ? big : small).push_back(std::move(root));
    }
    auto store_leaf = [](std::vector<std::vector<uint32_t>>& store, Node2& nd, std::vector<uint32_t>& lf) {
      nd.left = nd.right = -1;
      nd.first = static_cast<uint32_t>(store.size());
      store.push_back(std::move(lf));
    };
    // phase A: nodes above the grain, one at a time (the binning pass spread over the threads)
    while (!big.empty()) {
      SplitTask t = std::move(big.back());
      big.pop_back();
      SplitTask lt, rt;
      Node2 nd;
      std::vector<uint32_t> lf;
      if (!split(t, lt, rt, nthreads, nd, lf, height_fallbacks, spatial_splits, duplicates)) { store_leaf(leaves, nd, lf); nodes[t.node] = nd; continue; }
      const int32_t l = static_cast<int32_t>(nodes.size());
      nd.left = l; nd.right = l + 1;
      nodes[t.node] = nd;
      nodes.emplace_back();
      nodes.emplace_back();
      lt.node = l; rt.node = l + 1;
      const bool rbig = rt.refs.size() > grain, lbig = lt.refs.size() > grain;
      (rbig ? big : small).push_back(std::move(rt));
      (lbig ? big : small).push_back(std::move(lt));
    }
    // phase B: whole subtrees, one thread each, into local node arrays and local leaf stores
    struct Local { std::vector<Node2> nodes; std::vector<std::vector<uint32_t>> leaves; uint32_t fb = 0; uint64_t sp = 0, dup = 0; };
    std::vector<Local> local(small.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= small.size()) break;
        Local& L = local[k];
        L.nodes.emplace_back();
        std::vector<SplitTask> stack;
        small[k].node = 0;
        stack.push_back(std::move(small[k]));
        while (!stack.empty()) {
          SplitTask t = std::move(stack.back());
          stack.pop_back();
          SplitTask lt, rt;
          Node2 nd;
          std::vector<uint32_t> lf;
          const int32_t me = t.node;
          if (!split(t, lt, rt, 1, nd, lf, L.fb, L.sp, L.dup)) { store_leaf(L.leaves, nd, lf); L.nodes[me] = nd; continue; }
          const int32_t l = static_cast<int32_t>(L.nodes.size());
          nd.left = l; nd.right = l + 1;
          L.nodes[me] = nd;
          L.nodes.emplace_back();
          L.nodes.emplace_back();
          lt.node = l; rt.node = l + 1;
          stack.push_back(std::move(rt));
          stack.push_back(std::move(lt));
        }
      }
    };
    std::vector<int32_t> small_root(small.size());
    for (size_t k = 0; k < small.size(); ++k) small_root[k] = small[k].node;
    parallel_chunks(static_cast<size_t>(std::max<size_t>(std::min<size_t>(nthreads, small.size()), 1)),
                    static_cast<int>(std::min<size_t>(nthreads, small.size())), [&](int, size_t, size_t) { worker(); });
    // splice (serial: node and leaf indices are offset per subtree)
    for (size_t k = 0; k < small.size(); ++k) {
      const Local& L = local[k];
      const int32_t shift = static_cast<int32_t>(nodes.size()) - 1;
      const uint32_t lshift = static_cast<uint32_t>(leaves.size());
      for (size_t i = 0; i < L.nodes.size(); ++i) {
        Node2 nd = L.nodes[i];
        if (nd.leaf()) nd.first += lshift;
        else { nd.left += shift; nd.right += shift; }
        if (i == 0) nodes[small_root[k]] = nd; else nodes.push_back(nd);
      }
      for (const auto& lf : L.leaves) leaves.push_back(lf);
      height_fallbacks += L.fb; spatial_splits += L.sp; duplicates += L.dup;
    }
  }

  // record order: the leaves left to right; every node's (first, count) = the contiguous record range of its subtree
  void finish(std::vector<uint32_t>& record_face) {
    record_face.clear();
    struct It { int32_t node; bool done; };
    std::vector<It> st{{0, false}};
    while (!st.empty()) {
      It it = st.back();
      st.pop_back();
      Node2& nd = nodes[it.node];
      if (nd.leaf()) {
        const std::vector<uint32_t>& lf = leaves[nd.first];
        nd.first = static_cast<uint32_t>(record_face.size());
        nd.count = static_cast<uint32_t>(lf.size());
        record_face.insert(record_face.end(), lf.begin(), lf.end());
        continue;
      }
      if (!it.done) {
        st.push_back({it.node, true});
        st.push_back({nd.right, false});
        st.push_back({nd.left, false});
      } else {
        nd.first = nodes[nd.left].first;
        nd.count = nodes[nd.left].count + nodes[nd.right].count;
      }
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
  uint32_t guarded = 0;   // nodes expanded tallest-first to keep stack_need <= kStackEntries
};

// `guard` (collapse_guard below) is null for the plain collapse: children are expanded largest-area first, the rule of rounds 1-4.
// With a guard, a node whose area-first subtree would need more than kStackEntries stack entries expands tallest-first instead:
// both children of a BVH2 node of height h have height <= h-1, and expanding the taller one first leaves four grandchildren of
// height <= h-2, so from a node that holds `s` entries the descent needs <= s + 3 * ceil(h / 2).
struct CollapseGuard {
  std::vector<uint8_t> height;    // of every BVH2 node in THIS cut (0 = leaf of the cut)
  std::vector<uint8_t> area_need; // stack entries below a BVH4 node rooted here under area-first expansion (saturating at 255)
};

// `before(a, b)`: inner child a is expanded rather than b (a strict order; among equals the first slot wins)
template <class IsLeaf, class Before>
static int expand_children(const std::vector<Node2>& n2, int32_t root, IsLeaf is_leaf, Before before, int32_t kids[4]) {
  int nk = 0;
  if (is_leaf(root)) { kids[nk++] = root; return nk; }
  kids[nk++] = n2[root].left;
  kids[nk++] = n2[root].right;
  while (nk < 4) {
    int best = -1;
    for (int i = 0; i < nk; ++i) {
      if (is_leaf(kids[i])) continue;
      if (best < 0 || before(kids[i], kids[best])) best = i;
    }
    if (best < 0) break;
    const int32_t k = kids[best];
    kids[best] = n2[k].left;
    kids[nk++] = n2[k].right;
  }
  return nk;
}

static CollapseGuard collapse_guard(const std::vector<Node2>& n2, uint32_t leaf_limit) {
  CollapseGuard g;
  g.height.assign(n2.size(), 0);
  g.area_need.assign(n2.size(), 0);
  auto is_leaf = [&](int32_t id) { return n2[id].leaf() || n2[id].count <= leaf_limit; };
  auto area = [&](int32_t a, int32_t b) { return n2[a].b.area() > n2[b].b.area(); };
  for (size_t i = n2.size(); i-- > 0;) {   // children have larger indices than their parents
    const int32_t id = static_cast<int32_t>(i);
    if (is_leaf(id)) continue;
    g.height[i] = static_cast<uint8_t>(1 + std::max(g.height[n2[i].left], g.height[n2[i].right]));
    int32_t kids[4];
    const int nk = expand_children(n2, id, is_leaf, area, kids);
    uint32_t below = 0;
    for (int k = 0; k < nk; ++k) below = std::max<uint32_t>(below, g.area_need[kids[k]]);
    g.area_need[i] = static_cast<uint8_t>(std::min<uint32_t>(255u, below + static_cast<uint32_t>(nk - 1)));
  }
  return g;
}

Collapsed collapse(const std::vector<Node2>& n2, uint32_t leaf_limit, float pad, const CollapseGuard* guard = nullptr) {
  Collapsed out;
  auto is_leaf = [&](int32_t id) { return n2[id].leaf() || n2[id].count <= leaf_limit; };
  auto area = [&](int32_t a, int32_t b) { return n2[a].b.area() > n2[b].b.area(); };
  struct QItem { int32_t n2; uint32_t depth; uint32_t stack_before; };
  std::deque<QItem> queue;
  // the root is always an inner Node4, even for tiny meshes
  queue.push_back({0, 1, 0});
  std::vector<std::pair<uint32_t, int>> patch;  // (node4 index, slot) -> child n2 id to resolve later
  // every Node4 but a leaf-only root absorbs at least one inner BVH2 node: no reallocation while the tree grows
  out.nodes.reserve(n2.size() / 2 + 2);
  patch.reserve(n2.size() / 2 + 2);

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
    int nk;
    if (guard && it.stack_before + guard->area_need[it.n2] > kStackEntries - 1) {
      // tallest first; equal heights by area, so the choice stays a function of the tree alone
      nk = expand_children(n2, it.n2, is_leaf, [&](int32_t a, int32_t b) {
        return guard->height[a] != guard->height[b] ? guard->height[a] > guard->height[b] : area(a, b);
      }, kids);
      ++out.guarded;
    } else {
      nk = expand_children(n2, it.n2, is_leaf, area, kids);
    }
    // Slot order (round 5): the children sorted by box centre along the axis on which their centres spread most (ties: expansion
    // order).  A ray whose direction is positive on that axis meets the slots roughly near-to-far in slot order, a negative one in
    // reverse: the particle filter's walk takes that order instead of sorting the entry distances (traverse.hip.h node_step_so).
    // Every other traversal orders the children itself, so only their node numbering moves.  axis -> reserved[0] and, for the
    // quantised twin, the low two mantissa bits of its x scale (quantise()).
    {
      float c[4][3], lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (int i = 0; i < nk; ++i)
        for (int a = 0; a < 3; ++a) {
          c[i][a] = 0.5f * (n2[kids[i]].b.mn[a] + n2[kids[i]].b.mx[a]);
          lo[a] = std::min(lo[a], c[i][a]);
          hi[a] = std::max(hi[a], c[i][a]);
        }
      int ax = 0;
      for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
      int idx[4] = {0, 1, 2, 3};
      std::stable_sort(idx, idx + nk, [&](int x, int y) { return c[x][ax] < c[y][ax]; });
      int32_t sorted[4];
      for (int i = 0; i < nk; ++i) sorted[i] = kids[idx[i]];
      for (int i = 0; i < nk; ++i) kids[i] = sorted[i];
      nd.reserved[0] = static_cast<uint32_t>(ax);
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
void quantise(const std::vector<Node4>& nodes, std::vector<Node4Q>& qnodes, int nt) {
  qnodes.resize(nodes.size());
  parallel_chunks(nodes.size(), nt, [&](int, size_t lo_i, size_t hi_i) {
  for (size_t i = lo_i; i < hi_i; ++i) {
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
      if (a == 0) {
        // the node's slot-order axis rides in the low two mantissa bits of the x scale: the next float >= sc that ends in those bits
        // (a larger step still covers the boxes; the bytes below are computed with the final value)
        uint32_t bits;
        std::memcpy(&bits, &sc, 4);
        const uint32_t want = nd.reserved[0] & 3u;
        bits += (want - (bits & 3u)) & 3u;
        std::memcpy(&sc, &bits, 4);
      }
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
  });
}

}  // namespace

static std::string build_bvh_impl(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, BvhHost& out, uint32_t max_leaf);

// the boundary of the builder: an allocation that fails on a 10 M-face build, or a thread the system refuses, becomes an error string
// for the C entry points (which never throw) instead of crossing them
std::string build_bvh(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, BvhHost& out, uint32_t max_leaf) {
  try {
    return build_bvh_impl(verts, nv, faces, nf, out, max_leaf);
  } catch (const std::bad_alloc&) {
    return "BVH build: out of host memory";
  } catch (const std::exception& e) {
    return std::string("BVH build: ") + e.what();
  } catch (...) {
    return "BVH build: unknown failure";
  }
}

static std::string build_bvh_impl(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, BvhHost& out, uint32_t max_leaf) {
  if (!verts || !faces) return "null mesh pointers";
  if (nf == 0 || nv == 0) return "empty mesh";
  if (nf > 0x0FFFFFFFu) return "too many faces (max 2^28-1)";
  if (max_leaf < 1 || max_leaf > kMaxLeafTris) return "max_leaf must be 1..4";
  const int nt = build_threads();
  PhaseTimer timer;
  {
    std::atomic<int> bad{0};
    parallel_chunks(3 * static_cast<size_t>(nf), nt, [&](int, size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i)
        if (faces[i] >= nv) { bad.fetch_or(1); return; }
    });
    parallel_chunks(3 * static_cast<size_t>(nv), nt, [&](int, size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i)
        if (!std::isfinite(verts[i])) { bad.fetch_or(2); return; }
    });
    if (bad.load() & 1) return "face index out of range";
    if (bad.load() & 2) return "non-finite vertex coordinate";
  }

  // primitive boxes (by face id)
  std::vector<Prim> prims(nf);
  Box scene;
  scene.reset();
  {
    std::vector<Box> part(nt);
    parallel_chunks(nf, nt, [&](int k, size_t lo, size_t hi) {
      Box sb;
      sb.reset();
      for (size_t f = lo; f < hi; ++f) {
        Prim& p = prims[f];
        p.b.reset();
        for (int c = 0; c < 3; ++c) p.b.grow(verts + 3 * static_cast<size_t>(faces[3 * f + c]));
        for (int c = 0; c < 3; ++c) p.c[c] = 0.5f * (p.b.mn[c] + p.b.mx[c]);
        sb.grow(p.b);
      }
      part[k] = sb;
    });
    for (int k = 0; k < nt; ++k) scene.grow(part[k]);
  }

  timer.mark("validate + primitive boxes");
  // ONE BVH2, split down to the smallest leaf size any tree of the map uses.  `order`: the face of every record, in leaf order.
  // Maps of up to kSbvhMaxFaces faces are built with spatial splits (SplitBuilder above; RMCLHIP_SBVH_ALPHA = 0 switches them off);
  // larger ones by the in-place builder (object splits only: its passes are spread over the threads, 10 M faces in ~2 s).
  std::vector<uint32_t> order;
  std::vector<Node2> n2_store;
  uint32_t height_fallbacks = 0;
  const double alpha = sbvh_alpha();
  if (alpha > 0.0 && nf <= kSbvhMaxFaces) {
    std::vector<Ref> refs(nf);
    parallel_chunks(nf, nt, [&](int, size_t lo, size_t hi) {
      for (size_t f = lo; f < hi; ++f) {
        refs[f].b = prims[f].b;
        for (int c = 0; c < 3; ++c) refs[f].c[c] = prims[f].c[c];
        refs[f].face = static_cast<uint32_t>(f);
      }
    });
    std::vector<Prim>().swap(prims);
    SplitBuilder sb(verts, faces, std::min(max_leaf, kPfLeafTris), nt);
    const int64_t budget = std::min<int64_t>(std::max<int64_t>(static_cast<int64_t>(alpha * static_cast<double>(nf)), kSbvhFloor),
                                             static_cast<int64_t>(0x0FFFFFFF) - static_cast<int64_t>(nf));
    sb.build(std::move(refs), budget);
    sb.finish(order);
    n2_store = std::move(sb.nodes);
    height_fallbacks = sb.height_fallbacks;
    out.info.spatial_splits = static_cast<uint32_t>(std::min<uint64_t>(sb.spatial_splits, 0xFFFFFFFFu));
  } else {
    order.resize(nf);
    for (uint32_t f = 0; f < nf; ++f) order[f] = f;
    Builder bld(prims, order, std::min(max_leaf, kPfLeafTris), nt);
    bld.build();
    n2_store = std::move(bld.nodes);
    height_fallbacks = bld.height_fallbacks;
    std::vector<Prim>().swap(prims);
  }
  const std::vector<Node2>& n2 = n2_store;
  const size_t n_records = order.size();
  timer.mark("BVH2 (binned SAH)");

  // conservative padding of every stored box: the slab test runs in fp32 with fused ops and must
  // never cull a box whose triangle the (exact-spec) intersector would accept.
  float diag = 0.f, amax = 0.f;
  for (int k = 0; k < 3; ++k) {
    diag = std::max(diag, scene.mx[k] - scene.mn[k]);
    amax = std::max(amax, std::max(std::fabs(scene.mn[k]), std::fabs(scene.mx[k])));
  }
  const float pad = 1e-4f * std::max(diag, amax) + 1e-6f;

  // triangle records, written in leaf order
  out.tris.resize(n_records);
  parallel_chunks(n_records, nt, [&](int, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      const uint32_t f = order[i];
      const float* a = verts + 3 * static_cast<size_t>(faces[3 * static_cast<size_t>(f) + 0]);
      const float* b = verts + 3 * static_cast<size_t>(faces[3 * static_cast<size_t>(f) + 1]);
      const float* c = verts + 3 * static_cast<size_t>(faces[3 * static_cast<size_t>(f) + 2]);
      TriRec& r = out.tris[i];
      for (int k = 0; k < 3; ++k) {
        r.v0[k] = a[k];
        r.e1[k] = a[k] - b[k];
        r.e2[k] = c[k] - a[k];
      }
      cross_fma(r.e2, r.e1, r.Ng);
      const float d = std::sqrt((r.Ng[0] * r.Ng[0] + r.Ng[1] * r.Ng[1]) + r.Ng[2] * r.Ng[2]);
      for (int k = 0; k < 3; ++k) r.n[k] = (d > 0.f) ? r.Ng[k] / d : 0.f;
      r.face_id = f;
    }
  });

  timer.mark("triangle records");
  // The two cuts of the BVH2 (collapsed concurrently): the map's tree, leaves of <= max_leaf records, and the particle filter's tree
  // (quantised nodes only), leaves of <= kPfLeafTris records over the same record array.  The filter's rays are incoherent and its
  // kernel is bound by instruction issue with the lanes of a wave taking turns through the triangle loop: shorter leaves trade a
  // few more (cheap, quantised) node steps for fewer loop trips.
  // A cut whose area-first collapse would need more than kStackEntries stack entries is collapsed again under the guard (tallest
  // child first where the budget is short): stack_need <= kStackEntries holds for every mesh, map_upload only asserts it.
  auto collapse_bounded = [&](uint32_t leaf_limit) {
    Collapsed t = collapse(n2, leaf_limit, pad);
    if (t.stack_need > kStackEntries) {
      const CollapseGuard g = collapse_guard(n2, leaf_limit);
      t = collapse(n2, leaf_limit, pad, &g);
    }
    return t;
  };
  Collapsed main_tree, pf_tree;
  const bool own_pf_tree = max_leaf > kPfLeafTris;
  {
    // the two cuts side by side; chunk 0 (the caller) takes the map's, chunk 1 the filter's (parallel_chunks: exception- and EAGAIN-safe)
    if (own_pf_tree && nt > 1) {
      parallel_chunks(2, 2, [&](int k, size_t, size_t) {
        if (k == 0) main_tree = collapse_bounded(max_leaf);
        else pf_tree = collapse_bounded(kPfLeafTris);
      });
    } else {
      main_tree = collapse_bounded(max_leaf);
      if (own_pf_tree) pf_tree = collapse_bounded(kPfLeafTris);
    }
  }
  timer.mark("collapse (both cuts)");
  out.nodes = std::move(main_tree.nodes);
  quantise(out.nodes, out.qnodes, nt);
  out.info.height_fallbacks = height_fallbacks;
  out.info.guarded_nodes = main_tree.guarded + pf_tree.guarded;
  if (own_pf_tree) {
    quantise(pf_tree.nodes, out.qnodes_pf, nt);
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
  timer.mark("quantise");
  frontier_of(out.nodes, out.frontier);
  // ... and of the filter's tree: the batch traversal of find (kind 24) walks that tree (round 3: its two-triangle leaves make the
  // per-lane triangle loop 6-10 % cheaper for pose batches), whose node indices differ from the map tree's below the top levels
  frontier_of(out.nodes_pf, out.frontier_pf);

  out.cnodes.resize(out.nodes.size());
  parallel_chunks(out.nodes.size(), nt, [&](int, size_t lo_i, size_t hi_i) {
  for (size_t i = lo_i; i < hi_i; ++i) {
    const Node4& nd = out.nodes[i];
    Node4C& cn = out.cnodes[i];
    for (int c = 0; c < 4; ++c) {
      cn.c[c].lo[0] = nd.x[c]; cn.c[c].lo[1] = nd.y[c]; cn.c[c].lo[2] = nd.z[c];
      cn.c[c].hix = nd.x[4 + c]; cn.c[c].hiy = nd.y[4 + c]; cn.c[c].hiz = nd.z[4 + c];
      cn.c[c].ref = nd.child[c];
      cn.c[c].pad = 0;
    }
  }
  });

  timer.mark("frontier, child-major twins");
  out.info.n_faces = nf;
  out.info.n_records = static_cast<uint32_t>(n_records);
  out.info.n_vertices = nv;
  out.info.n_nodes = static_cast<uint32_t>(out.nodes.size());
  out.info.max_depth = main_tree.max_depth;
  out.info.stack_need = main_tree.stack_need;
  out.info.pad = pad;
  for (int k = 0; k < 3; ++k) { out.info.bbox_min[k] = scene.mn[k]; out.info.bbox_max[k] = scene.mx[k]; }
  return std::string();
}

}  // namespace rmclhip
