// find_kernel.hip.h -- k_find<model, traversal kind, clocks>: ray-casting correspondences, rm::*Simulator*::simulate as called by
// RCC*::find (rmcl/src/rmcl/registration/RCCEmbree.cpp:26-36,89-99).  One template, two homes: kernels.hip instantiates the kinds
// the product can select (0 packet, 2 quad, 23 / 24 one lane per ray with the frontier start) WITHOUT clocks; kernels_lab.hip instantiates
// the measured-and-rejected kinds and the clocked variants for tools/wave_timeline.py.
#pragma once
#include "traverse.hip.h"

namespace rmclhip {
namespace {

// ---------------------------------------------------------------------------------------------
// find
// ---------------------------------------------------------------------------------------------
// kTrav: 0 = wave packet, 1 = one lane per ray (while-while), 4 = the same on the quantised 64-B nodes, 5 = one lane
// per ray with the tail of every wave handed to quads,
// 2 = four lanes per ray (quad-cooperative; the block
// of 256 threads then covers ONE 64-ray tile instead of four)
// kTrav 5..10 share the tail traversal: 6 / 7 add the LDS-resident top of the tree (85 / 341 nodes = levels 0-3 / 0-4 of a
// full BVH4), 8 adds the one-round-trip leaf, 9 / 10 both
constexpr int find_top_nodes(int trav) { return (trav == 6 || trav == 9) ? 85 : ((trav == 7 || trav == 10) ? 341 : 0); }
constexpr bool find_leaf_batch(int trav) { return trav >= 8 && trav <= 10; }
constexpr int kFindBfRows = 24;  // LDS stack rows per lane (sentinel included) of the branch-free lane traversal in k_find
constexpr uint32_t kFindTailLdsDwords = 16u * 256u + kQuadStackEntries * 64u + 4u * kTailRays * kTailXferDwords;

// kinds 19..22: kind 17 + leaving the node phase when 32 / 24 / 16 / 8 lanes hold a leaf
constexpr int find_leaf_trigger(int trav) { return (trav == 19 || trav == 20 || trav == 23 || (trav >= 26 && trav <= 30)) ? 10 : 0; }   // leave at <= 4/10 of the round's rays
// kinds 23 / 24: kinds 19 / 22 whose rays start at the map's frontier (traverse.hip.h frontier_start) instead of the root
constexpr bool find_frontier(int trav) { return trav == 23 || trav == 24 || (trav >= 25 && trav <= 30); }   // 26: 23 on the quantised nodes; 27: 23 + record prefetch; 28: 23 with the pipelined node step; 29 / 30: 23 with four / three of the five ordering steps
// kind 25: kind 2 (four lanes per ray) with the frontier start
constexpr bool find_quad(int trav) { return trav == 2 || trav == 25; }

// the ray of image position (cv, ch) in the SENSOR frame (loc = cv * W + ch)
template <uint32_t kModel>
__device__ __forceinline__ void find_ray_s(const FindParams& p, uint32_t cv, uint32_t ch, uint32_t loc, f3& dir_s, f3& orig_s) {
  if (kModel == kModelSpherical) {
    // rmagine SphericalModel::getDirection (convention pinned by rmcl_ros/src/util/conversions.cpp:174-188);
    // the four trig tables hold the host libm values of cos/sin(phi_v), cos/sin(theta_h)
    const float cp = p.model_tab[cv], sp = p.model_tab[p.H + cv];
    const float ct = p.model_tab[2u * p.H + ch], st = p.model_tab[2u * p.H + p.W + ch];
    dir_s = mk3(cp * ct, cp * st, sp);
  } else if (kModel == kModelPinhole) {
    dir_s = pinhole_direction(p.pin_f[0], p.pin_f[1], p.pin_c[0], p.pin_c[1], cv, ch);
  } else if (kModel == kModelOnDn) {
    const float* og = p.model_tab + 3u * static_cast<size_t>(loc);
    const float* dr = p.model_tab + 3u * (static_cast<size_t>(p.W) * p.H + loc);
    orig_s = mk3(og[0], og[1], og[2]);
    dir_s = mk3(dr[0], dr[1], dr[2]);
  } else {
    dir_s = mk3(p.model_tab[3u * loc], p.model_tab[3u * loc + 1u], p.model_tab[3u * loc + 2u]);
  }
}

// The plane table of the frontier start (traverse.hip.h: tile_planes_wave): one wave per tile of the scan image, 16 floats per
// tile, in the sensor frame.  Run once per (model, tiling), not per find.  grid = ceil(ntiles / 4) blocks of 256.
template <uint32_t kModel>
__global__ void __launch_bounds__(256) k_tile_planes(const FindParams p, float* __restrict__ planes) {
  const uint32_t lane = threadIdx.x & 63u, tile = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (tile >= p.tiles_x * p.tiles_y) return;
  const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const uint32_t twl = p.tile_w_log2;
  const uint32_t lx = lane & ((1u << twl) - 1u), ly = lane >> twl;
  const uint32_t vid = (ty << (6u - twl)) + ly, hid = (tx << twl) + lx;
  const bool valid = (vid < p.H) && (hid < p.W);
  const uint32_t cv = valid ? vid : 0u, ch = valid ? hid : 0u;
  f3 dir_s, orig_s = p.orig_s;
  find_ray_s<kModel>(p, cv, ch, cv * p.W + ch, dir_s, orig_s);
  const bool finite = (dir_s.x == dir_s.x) && (dir_s.y == dir_s.y) && (dir_s.z == dir_s.z);
  float out[16];
  tile_planes_wave(dir_s, valid && finite, twl, out);
  if (lane < 16u) {
    float v = out[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) v = (lane == static_cast<uint32_t>(k)) ? out[k] : v;
    planes[static_cast<size_t>(tile) * 16u + lane] = v;
  }
}

// kClock: entry / traversal / store clocks of every wave go to p.wave_clock (tools/wave_timeline.py); the production
// instantiations are built with kClock = false and contain no s_memtime
template <uint32_t kModel, int kTrav, bool kClock = false>
__global__ void __launch_bounds__(256) k_find(const FindParams p) {
  extern __shared__ uint32_t lds_dyn[];
  constexpr bool kPacket = (kTrav == 0);
  constexpr bool kQuad = find_quad(kTrav);
  constexpr int kTop = find_top_nodes(kTrav);
  const uint32_t lane = kQuad ? (threadIdx.x >> 2) : (threadIdx.x & 63u), wave = threadIdx.x >> 6;
  const uint32_t sub = threadIdx.x & 3u;  // quad mode: child slot / triangle slot / output role of this lane
  uint32_t clk_begin = 0, clk_real = 0;
  if (kClock && p.wave_clock != nullptr) {  // diagnostics (tools/wave_timeline.py)
    uint64_t t, r;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) : : "memory");
    clk_begin = static_cast<uint32_t>(t);
    clk_real = static_cast<uint32_t>(r);
  }
  if constexpr (kTop > 0) {
    // the block's copy of the top of the tree: coalesced 16-B pieces, all requested before the first LDS write
    uint4* dst = reinterpret_cast<uint4*>(lds_dyn + kFindTailLdsDwords);
    const uint4* src = reinterpret_cast<const uint4*>(p.nodes);
    const uint32_t n16 = min(static_cast<uint32_t>(kTop), p.n_nodes) * 8u;
    constexpr int kRounds = (kTop > 0) ? (kTop * 8 + 255) / 256 : 1;
    uint4 v[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const uint32_t i = static_cast<uint32_t>(r) * 256u + threadIdx.x;
      if (i < n16) v[r] = src[i];
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const uint32_t i = static_cast<uint32_t>(r) * 256u + threadIdx.x;
      if (i < n16) dst[i] = v[r];
    }
    __syncthreads();
  }
  // XCD-aware remap: the dispatcher places block b on XCD b%8; give every XCD a contiguous range of
  // tiles so neighbouring tiles (which walk the same subtrees) share one L2.  gridDim.x % 8 == 0.
  const uint32_t chunk = gridDim.x >> 3;
  const uint32_t vb = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
  const uint32_t tile = (kQuad ? vb : (vb * 4u + wave));
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  if (tile >= ntiles) return;
  const uint32_t pose = blockIdx.y;
  const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const uint32_t twl = p.tile_w_log2;
  const uint32_t lx = lane & ((1u << twl) - 1u), ly = lane >> twl;
  const uint32_t vid = (ty << (6u - twl)) + ly, hid = (tx << twl) + lx;
  const bool valid = (vid < p.H) && (hid < p.W);
  const uint32_t cv = valid ? vid : 0u, ch = valid ? hid : 0u;
  const uint32_t loc = cv * p.W + ch;

  xform Tsm, Tms;
  if (p.Tsm_arr != nullptr) { Tsm = p.Tsm_arr[pose]; Tms = p.Tms_arr[pose]; }
  else { Tsm = p.Tsm; Tms = p.Tms; }

  f3 dir_s, orig_s = p.orig_s;
  find_ray_s<kModel>(p, cv, ch, loc, dir_s, orig_s);
  const f3 org_m = (kModel == kModelSpherical || kModel == kModelPinhole) ? Tsm.t : xapply(Tsm, orig_s);
  const f3 dir_m = qrot(Tsm.R, dir_s);
  const bool finite = (dir_m.x == dir_m.x) && (dir_m.y == dir_m.y) && (dir_m.z == dir_m.z);
  const float ray_tfar = (valid && finite) ? p.tfar : -1.0f;

  uint32_t clk_trace0 = 0, clk_trace1 = 0, clk_visits = 0, clk_dbg[3] = {0u, 0u, 0u};
  if (kClock && p.wave_clock != nullptr) {
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    clk_trace0 = static_cast<uint32_t>(t);
  }
  RayHit h;
  uint4 pre_nrec = uint4{0u, 0u, 0u, 0u};   // kind 27: the best record's last 16 B, requested during the traversal
  uint32_t pre_rec = kNone;
  if (kPacket) {
    trace_packet((cu32p)(p.nodes), (cu32p)(p.tris), org_m, dir_m, ray_tfar, lane, h);
  } else if (kQuad) {
    bool started = false;
    if constexpr (kTrav == 25 && kModel != kModelOnDn) {
      if (p.tile_planes != nullptr) {
        // the block's 64-ray tile has ONE pyramid; each wave (16 rays x 4 lanes) culls the frontier against it with its 64 lanes
        // and every lane filters the survivors with its ray (the four lanes of a ray agree and store the same rows)
        const float* planes = p.tile_planes + static_cast<size_t>(__builtin_amdgcn_readfirstlane(tile)) * 16u;
        const TraceStart st = frontier_start<static_cast<int>(kQuadStackEntries), 1>(
            p.frontier, p.n_frontier, p.scene_center, p.scene_half_diag, planes, Tsm.R, p.tfar, org_m, dir_m, ray_tfar, threadIdx.x & 63u,
            lds_dyn + lane, 64u);
        QuadResume rsm;
        rsm.cur = st.cur; rsm.n_stack = st.sp - 1u; rsm.best_t = ray_tfar; rsm.best_face = kInvalidFace; rsm.best_rec = 0u;
        trace_quad<true>(p.cnodes, p.tris, org_m, dir_m, ray_tfar, sub, lane, lds_dyn, h, &rsm);
        started = true;
      }
    }
    if (!started) trace_quad(p.cnodes, p.tris, org_m, dir_m, ray_tfar, sub, lane, lds_dyn, h);
  } else {
    // kind 4 serves launches that fill the chip (pose batches): throughput, not the slowest wave's chain, is what counts there, and
    // the branchy step with its partial sort and 16 LDS rows (more resident waves) is 11 % faster than the branch-free one
    // where the ray starts: the root -- or, for the frontier kinds, what frontier_start returns.  ALWAYS a struct passed by
    // address, never "null or &start": a pointer chosen at run time forces the struct through scratch memory (8 B per ray
    // written and read back on the critical path of every wave; seen as 1 MiB of extra HBM writes per C2 launch)
    constexpr bool kWwStack = (kTrav == 4 || kTrav == 22 || kTrav == 24);   // trace_lane_ww: first stack row 0; branch-free forms: row 1
    TraceStart start;
    start.cur = (ray_tfar >= 0.0f) ? 0u : 0x7FFFFFFFu;
    start.sp = kWwStack ? 0u : 1u;
    const TraceStart* sp0 = &start;
    if constexpr (find_frontier(kTrav) && kModel != kModelOnDn) {   // (OnDn: one origin per ray, no common pyramid)
      if (p.tile_planes != nullptr) {     // (no table: the rays start at the root)
        const float* planes = p.tile_planes + static_cast<size_t>(__builtin_amdgcn_readfirstlane(tile)) * 16u;
        if (kTrav == 23 || (kTrav >= 26 && kTrav <= 30))
          start = frontier_start<kFindBfRows, 1>(p.frontier, p.n_frontier, p.scene_center, p.scene_half_diag, planes, Tsm.R, p.tfar, org_m,
                                                 dir_m, ray_tfar, lane, lds_dyn + threadIdx.x, kBfStride);
        else
          start = frontier_start<16, 0>(p.frontier, p.n_frontier, p.scene_center, p.scene_half_diag, planes, Tsm.R, p.tfar, org_m, dir_m,
                                        ray_tfar, lane, lds_dyn + threadIdx.x, blockDim.x);
      }
    }
    if (kTrav == 4) trace_lane_ww<16, true>(p.qnodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, blockDim.x, h);
    else if (kTrav == 22 || kTrav == 24)
      trace_lane_ww<16, true, false, true>(p.qnodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, blockDim.x, h, sp0);
    else if (kTrav == 21)
      trace_lane_ww_tail<16, 0, false, true>(
          p.nodes, p.cnodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, blockDim.x, lds_dyn + 16u * 256u,
          lds_dyn + 16u * 256u + kQuadStackEntries * 64u + (threadIdx.x >> 6) * (kTailRays * kTailXferDwords), h,
          lds_dyn + kFindTailLdsDwords);
    else if (kTrav == 1) trace_lane_bf<kFindBfRows>(p.nodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, h);
    else if (kTrav == 12) trace_lane_bf<kFindBfRows, false, true>(p.nodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, h);
    else if (kTrav == 16 || kTrav == 17 || kTrav == 19 || kTrav == 20 || kTrav == 23 || (kTrav >= 26 && kTrav <= 30))
      trace_lane_bf_tail<kFindBfRows, kTrav != 16 && kTrav != 20, find_leaf_trigger(kTrav), kTrav == 26, kTrav == 27, kTrav == 28, (kTrav == 29 ? 4 : (kTrav == 30 ? 3 : 5))>(kTrav == 26 ? p.qnodes : p.nodes, p.cnodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x,
                                                   lds_dyn + kFindBfRows * 256u,
                                                   lds_dyn + kFindBfRows * 256u + kQuadStackEntries * 64u + (threadIdx.x >> 6) * (kTailRays * kTailXferDwords), h,
                                                   kClock ? &clk_visits : nullptr, sp0, kClock ? clk_dbg : nullptr, &pre_nrec, &pre_rec);
    else if (kTrav == 13) trace_lane_bf<kFindBfRows, false, false, true>(p.nodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, h);
    else if (kTrav == 14) trace_lane_bf<kFindBfRows, false, true, true>(p.nodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, h);
    else if (kTrav >= 5 && kTrav <= 10)
      trace_lane_ww_tail<16, kTop, find_leaf_batch(kTrav)>(
          p.nodes, p.cnodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, blockDim.x, lds_dyn + 16u * 256u,
          lds_dyn + 16u * 256u + kQuadStackEntries * 64u + (threadIdx.x >> 6) * (kTailRays * kTailXferDwords), h,
          lds_dyn + kFindTailLdsDwords);
    else trace_lane_ww<16>(p.nodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x, blockDim.x, h);
  }

  if (kClock && p.wave_clock != nullptr) {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    clk_trace1 = static_cast<uint32_t>(t);
  }
  if (valid) {
  const size_t g = static_cast<size_t>(pose) * p.W * p.H + loc;
  const bool found = (h.rec != kNone);
  // quad mode: the four lanes of a ray hold the same result and share the stores (0: hits/ranges/face ids, 1: points,
  // 2: normals)
  const bool w0 = !kQuad || sub == 0u, w1 = !kQuad || sub == 1u, w2 = !kQuad || sub == 2u;
  if (found) {
    if (p.hits && w0) p.hits[g] = 1;
    if (p.ranges && w0) p.ranges[g] = h.t;
    if (p.points && w1) {
      f3 pt = scale3(dir_s, h.t);
      if (kModel == kModelO1Dn || kModel == kModelOnDn) pt = add3(pt, orig_s);
      p.points[3 * g] = pt.x; p.points[3 * g + 1] = pt.y; p.points[3 * g + 2] = pt.z;
    }
    // the record's last 16 B: unit normal + the ORIGINAL face id
    if ((p.normals && w2) || (p.face_ids && w0)) {
      uint4 nrec;
      if (kTrav == 27 && pre_rec == h.rec) nrec = pre_nrec;   // (rays finished by the quad tail fetch it here)
      else nrec = reinterpret_cast<const uint4*>(p.tris)[static_cast<size_t>(h.rec) * 4u + 3u];
      if (p.normals && w2) {
        f3 n = qrot(Tms.R, mk3(asf(nrec.x), asf(nrec.y), asf(nrec.z)));
        if (dot_plain(dir_s, n) > 0.0f) n = neg3(n);  // flip towards the sensor
        p.normals[3 * g] = n.x; p.normals[3 * g + 1] = n.y; p.normals[3 * g + 2] = n.z;
      }
      if (p.face_ids && w0) p.face_ids[g] = nrec.w;
    }
  } else {
    const float qn = __uint_as_float(0x7FC00000u);
    if (p.hits && w0) p.hits[g] = 0;
    if (p.ranges && w0) p.ranges[g] = p.tfar + 1.0f;
    if (p.points && w1) { p.points[3 * g] = qn; p.points[3 * g + 1] = qn; p.points[3 * g + 2] = qn; }
    if (p.normals && w2) { p.normals[3 * g] = qn; p.normals[3 * g + 1] = qn; p.normals[3 * g + 2] = qn; }
    if (p.face_ids && w0) p.face_ids[g] = kInvalidFace;
  }
  }  // valid
  if (kClock && p.wave_clock != nullptr) {
    uint64_t t;
    uint64_t t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) : : "memory");   // stores issued
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");  // ... and completed
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // diagnostics of the lane traversals with counters: hardest ray's node visits | rays handed to quads | steps on the general
    // (scratch-capable) path | cycles inside the quad tail / 64
    auto wmax = [](uint32_t v, int bits) { uint32_t m = 0; for (int b = bits - 1; b >= 0; --b) { const uint32_t c = m | (1u << b); if (__any(v >= c)) m = c; } return m; };
    const uint32_t clk_w7 = wmax(min(clk_visits, 63u), 6) | (wmax(min(clk_dbg[1], 31u), 5) << 6) | (wmax(min(clk_dbg[0], 511u), 9) << 11) |
                            (wmax(min(clk_dbg[2] >> 6, 4095u), 12) << 20);
    if ((threadIdx.x & 63u) == 0u) {
      uint32_t* w = p.wave_clock + 8u * ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + wave);
      w[0] = clk_begin; w[1] = static_cast<uint32_t>(t); w[2] = clk_real; w[3] = (tile & 0xFFFFFFu) | (xcc << 24);
      w[4] = clk_trace0; w[5] = clk_trace1; w[6] = static_cast<uint32_t>(t2);
      w[7] = clk_w7;
    }
  }
}

}  // namespace
}  // namespace rmclhip
