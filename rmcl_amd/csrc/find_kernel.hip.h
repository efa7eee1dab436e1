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
constexpr int find_leaf_trigger(int trav) { return (trav == 19 || trav == 20 || trav == 23 || (trav >= 26 && trav <= 32)) ? 10 : 0; }   // leave at <= 4/10 of the round's rays
// kinds 23 / 24: kinds 19 / 22 whose rays start at the map's frontier (traverse.hip.h frontier_start) instead of the root
// kinds 31 / 32 (round 6): the wave keeps descending cooperatively below the frontier (traverse.hip.h frontier_descent_start); 32 = the product's form (one bit per final leaf and ray), 31 = the first form (sorted hand-over), kept in the lab for A/B
constexpr bool find_frontier(int trav) { return trav == 23 || trav == 24 || (trav >= 25 && trav <= 32); }
// LDS of the one-lane-per-ray kinds with quad-finished tails (23, 31, ...), in dwords; kinds 31 and 32 append their waves' descent lists
constexpr uint32_t kFindBfTailLdsDwords = static_cast<uint32_t>(kFindBfRows) * 256u + kQuadStackEntries * 64u + 4u * kTailRays * kTailXferDwords;
// kinds 31 and 32 have no quad tail: its LDS is the lane stacks, then the four waves' descent lists
constexpr uint32_t kFind31ListsAt = static_cast<uint32_t>(kFindBfRows) * 256u;
constexpr uint32_t kFind31LdsDwords = kFind31ListsAt + 4u * kDescentWaveDwords;
static_assert(kFind31ListsAt % 4u == 0u, "the descent lists are read and written 16 B at a time");   // 26: 23 on the quantised nodes; 27: 23 + record prefetch; 28: 23 with the pipelined node step; 29 / 30: 23 with four / three of the five ordering steps
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

// ---------------------------------------------------------------------------------------------
// MICP moment epilogue of k_find (kMoments; launch_find_moments).  The gate-stable moment form (kernels.hip k_micp_moments) needs,
// per correspondence that is certainly gated in, the 82 sums  sum X_a Y_b  of the factor vectors
//   X = (N0N0, N0N1, N0N2, N1N1, N1N2, N2N2 | 1 | sN0, sN1, sN2)      s = N . I
//   Y = (1 | D0, D1, D2 | D0D0, D0D1, D0D2, D1D1, D1D2, D2D2)
// of (dataset point D, model point I, model normal N): that IS a dense product X^T Y (10 x 10, 18 entries unused) over the
// correspondences -- the one place on this path where the correspondence stack is tiled as a GEMM.  A wave holds 64 correspondences,
// one per lane, right after its traversal; it stages the factors of 32 of them at a time in its OWN stack columns (free once the
// traversal has returned; no other wave touches them) and feeds v_mfma_f64_16x16x4_f64: A[i][k] = X_i of correspondence k,
// B[k][j] = Y_j of correspondence k, 16 instructions for the wave's 64 correspondences, f64 accumulation.  The four waves of a
// workgroup add their tiles through a 3-KB array and write ONE partial row in k_micp_moments' layout; the loop kernel is unchanged
// except that the mask words of the undecided correspondences arrive in the find's tile order.
// ---------------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr uint32_t kMomRow = 96;          // == kMicpFastMoments

// The wave's staging area = its 24 x 64 dwords of stack columns: row r of the wave is 256 contiguous bytes = 32 doubles, one per staged
// correspondence.  Factor X_i of correspondence c sits in row i, factor Y_i in row 10 + i, both in column (c + 4 i) mod 32: the skew
// spreads the 16 rows an MFMA operand fetch touches (a row stride of 1 KB would put them all in one bank) and keeps every access one
// base register + an immediate: a row is 1024 B further, the Y row of the same i 10 240 B further, the next MFMA step 32 B.
__device__ __forceinline__ char* mom_stage_base(uint32_t* lds_dyn, uint32_t wave) {
  return reinterpret_cast<char*>(lds_dyn + wave * 64u);
}

// the workgroup's partial row: sum of the four waves' 16 x 16 tiles (every wave of the workgroup calls this exactly once); thread t
// writes row entry t, which is tile element (x, y) = the inverse of k_micp_moments' layout
constexpr uint32_t kMomTile = 256;
__device__ __forceinline__ void find_moments_block_sum(const FindParams& p, double (*s_red)[kMomTile], const uint32_t* s_piece = nullptr,
                                                       uint32_t tile_word = 0u) {
  __syncthreads();
  const uint32_t t = threadIdx.x;
  // quad kind: the tile's 64 rays are spread over the four waves (16 each): their 16-bit pieces make the tile's mask word
  if (s_piece != nullptr && t == 0u)
    p.mom_unc_mask[tile_word] = static_cast<unsigned long long>(s_piece[0]) | (static_cast<unsigned long long>(s_piece[1]) << 16) |
                                (static_cast<unsigned long long>(s_piece[2]) << 32) | (static_cast<unsigned long long>(s_piece[3]) << 48);
  if (t < kMomRow) {
    // n | D[3] | DD[6] | sN[3] | sND[9] | NN[6] | NND[18] | NNDD[36]   <->   X = (NN[6] | 1 | sN[3]), Y = (1 | D[3] | DD[6])
    uint32_t x = 16u, y = 0u;   // x = 16: an unused entry (82 .. 95)
    if (t < 10u) { x = 6u; y = t; }
    else if (t < 13u) { x = 7u + (t - 10u); y = 0u; }
    else if (t < 22u) { x = 7u + (t - 13u) / 3u; y = 1u + (t - 13u) % 3u; }
    else if (t < 28u) { x = t - 22u; y = 0u; }
    else if (t < 46u) { x = (t - 28u) / 3u; y = 1u + (t - 28u) % 3u; }
    else if (t < 82u) { x = (t - 46u) / 6u; y = 4u + (t - 46u) % 6u; }
    const uint32_t e = (x & 15u) * 16u + y;
    const double v = ((s_red[0][e] + s_red[1][e]) + s_red[2][e]) + s_red[3][e];
    p.mom_partials[static_cast<size_t>(blockIdx.x) * kMomRow + t] = (x < 16u) ? v : 0.0;
  }
}

// a wave without a tile (grid padding): no correspondences
template <bool kQuadRays>
__device__ __forceinline__ void find_moments_idle_wave(const FindParams& p, double (*s_red)[kMomTile], uint32_t* s_piece, uint32_t word_index,
                                                       uint32_t wave, uint32_t lane) {
#pragma unroll
  for (uint32_t v = 0; v < 4u; ++v) s_red[wave][v * 64u + lane] = 0.0;
  if (kQuadRays) {
    if (lane == 0u) s_piece[wave] = 0u;
    find_moments_block_sum(p, s_red, s_piece, word_index);
  } else {
    if (lane == 0u) p.mom_unc_mask[word_index] = 0ull;
    find_moments_block_sum(p, s_red);
  }
}

// the correspondence's dataset side, requested BEFORE the traversal (its latency would otherwise sit at the end of the slowest wave)
struct MomDataset { f3 D; bool ok; };
__device__ __forceinline__ MomDataset find_moments_dataset(const FindParams& p, bool valid, uint32_t loc) {
  MomDataset d;
  d.D = mk3(0.f, 0.f, 0.f);
  d.ok = valid && (loc < p.mom_n);
  if (d.ok) {
    d.ok = (p.mom_dataset_mask == nullptr) || (p.mom_dataset_mask[loc] > 0);
    const float* dp = p.mom_dataset_points + 3 * static_cast<size_t>(loc);
    d.D = mk3(dp[0], dp[1], dp[2]);
  }
  return d;
}

// kQuadRays: the quad traversal -- a ray is held by four lanes (the one with sub == 0 speaks for it: `have` is false in the others), a wave
// has 16 of the tile's 64 rays: the MFMA passes run over 64 lanes of which 16 contribute, the mask word is assembled per workgroup
template <bool kQuadRays>
__device__ __forceinline__ void find_moments_wave(const FindParams& p, uint32_t* lds_dyn, double (*s_red)[kMomTile], uint32_t* s_piece, bool have,
                                                  MomDataset ds, f3 Ii, f3 Ni, uint32_t word_index, uint32_t wave, uint32_t lane) {
  // classification: k_micp_moments' own arithmetic at the identity pre-transform
  const bool ok = have && ds.ok;
  const float spd0 = dot_plain(sub3(Ii, ds.D), Ni);
  const float nd = sqrtf(dot_plain(ds.D, ds.D));
  const int cls = micp_gate_class(spd0, nd, p.mom_gate_lo, p.mom_gate_hi, p.mom_rho_cap, p.mom_tau_cap);
  const bool uncertain = ok && cls == 2;
  unsigned long long word = __ballot(uncertain);
  if (kQuadRays) {
    // bits sit at lanes 4 r (sub == 0): lane r < 16 fetches lane 4 r's flag, the ballot of those is the wave's 16-bit piece
    const int u4 = __shfl(uncertain ? 1 : 0, static_cast<int>((lane & 15u) << 2), 64);
    word = __ballot(lane < 16u && u4 != 0);
    if (lane == 0u) s_piece[wave] = static_cast<uint32_t>(word);
  } else if (lane == 0u) {
    p.mom_unc_mask[word_index] = word;
  }
  const bool gate = ok && cls == 1;
  double X[10], Y[10];
  {
    // a gated-out lane contributes zeros (its inputs may be NaN: select, do not multiply)
    const f3 Dz = gate ? ds.D : mk3(0.f, 0.f, 0.f), Nz = gate ? Ni : mk3(0.f, 0.f, 0.f), Iz = gate ? Ii : mk3(0.f, 0.f, 0.f);
    const double g = gate ? 1.0 : 0.0;
    const double D0 = Dz.x, D1 = Dz.y, D2 = Dz.z, N0 = Nz.x, N1 = Nz.y, N2 = Nz.z, I0 = Iz.x, I1 = Iz.y, I2 = Iz.z;
    const double sI = (N0 * I0 + N1 * I1) + N2 * I2;
    X[0] = N0 * N0; X[1] = N0 * N1; X[2] = N0 * N2; X[3] = N1 * N1; X[4] = N1 * N2; X[5] = N2 * N2;
    X[6] = g; X[7] = sI * N0; X[8] = sI * N1; X[9] = sI * N2;
    Y[0] = g; Y[1] = D0; Y[2] = D1; Y[3] = D2;
    Y[4] = D0 * D0; Y[5] = D0 * D1; Y[6] = D0 * D2; Y[7] = D1 * D1; Y[8] = D1 * D2; Y[9] = D2 * D2;
  }
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  const uint32_t i16 = lane & 15u, k4 = lane >> 4;
  char* stage = mom_stage_base(lds_dyn, wave);
  constexpr uint32_t kRowBytes = kBfStride * 4u;   // 1024
  // operand lanes 10 .. 15 of an MFMA step (the tile is 16 wide, the factor vectors 10 long) read zeros: row 20 of the wave
  if (lane < 32u) *reinterpret_cast<double*>(stage + 20u * kRowBytes + lane * 8u) = 0.0;
  const bool opl = i16 < 10u;
  const char* row_a = stage + (opl ? i16 : 20u) * kRowBytes;
  const char* row_b = stage + (opl ? 10u + i16 : 20u) * kRowBytes;
  const uint32_t col0 = k4 + 4u * i16;
#pragma unroll
  for (uint32_t half = 0; half < 2u; ++half) {
    if ((lane >> 5) == half) {
      const uint32_t c = lane & 31u;
#pragma unroll
      for (uint32_t i = 0; i < 10u; ++i) {
        char* at = stage + i * kRowBytes + ((c + 4u * i) & 31u) * 8u;
        *reinterpret_cast<double*>(at) = X[i];
        *reinterpret_cast<double*>(at + 10u * kRowBytes) = Y[i];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (uint32_t s = 0; s < 8u; ++s) {
      const uint32_t cb = ((col0 + 4u * s) & 31u) * 8u;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(*reinterpret_cast<const double*>(row_a + cb), *reinterpret_cast<const double*>(row_b + cb), acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // v_mfma_f64_16x16x4_f64 leaves D[x = 4 * v + lane / 16][y = lane % 16] in acc[v] (checked against k_micp_moments' sums: rmclhip_debug_micp_moments)
#pragma unroll
  for (uint32_t v = 0; v < 4u; ++v) s_red[wave][(4u * v + k4) * 16u + i16] = acc[v];
  if (kQuadRays) find_moments_block_sum(p, s_red, s_piece, word_index);
  else find_moments_block_sum(p, s_red);
}

// kClock: entry / traversal / store clocks of every wave go to p.wave_clock (tools/wave_timeline.py); the production
// instantiations are built with kClock = false and contain no s_memtime
template <uint32_t kModel, int kTrav, bool kClock = false, bool kMoments = false>
__global__ void __launch_bounds__(256) k_find(const FindParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  static_assert(!kMoments || ((kTrav == 23 || kTrav == 31 || kTrav == 32 || kTrav == 2) && !kClock), "the moment epilogue is built for kinds 23, 31, 32 and 2");
  __shared__ double s_mom_red[kMoments ? 4 : 1][kMoments ? kMomTile : 1];
  __shared__ uint32_t s_mom_piece[4];
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
  // Which workgroup computes which tile.  The dispatcher places workgroup b on XCD b % 8 and deals an XCD's workgroups round-robin over
  // its 32 CUs.  Rounds 1-5 gave every XCD a contiguous range of tiles (neighbouring tiles walk the same subtrees: one L2) -- but a
  // single scan is bound by issue slots and by its slowest waves, not by L2 misses, and a scan's hard tiles are neighbours (a room's
  // grazing rows): contiguous ranges put them on ONE or two XCDs and, two tile rows apart, on the same SIMDs.  Round 6
  // (profiles/r06_tile_mapping_ab.txt): single scans are dealt as the hardware deals them -- workgroup b = tiles 4b .. 4b+3, so that
  // every XCD sees every eighth workgroup of every tile row (room-100k kind 23: 25.3 -> 22.2 us, sphere-100k 17.4 -> 16.4, 1 M faces
  // 22.3 -> 21.3).  p.xcd_mapping: 0 = that; 1 = contiguous ranges; 2 = the two workgroups of a CU from the two halves of the image
  // (rmclhip_rcc_autotune measures the three).  gridDim.x % 8 == 0.
  // Pose batches launched pose-major (gridDim.y > 1; world order off) keep the contiguous ranges, rotated by the pose: a model of a few
  // tiles fills only the first of the eight ranges, and without the rotation every pose's occupied range sat on the SAME XCD (round 4:
  // 2000 poses x 32x32 rays ran on an eighth of the chip, 0.80 ms; rotated 0.2x ms -- profiles/r04_v1_batch_breakdown.txt).
  const uint32_t chunk = gridDim.x >> 3;
  const uint32_t xr = (gridDim.y > 1u) ? ((blockIdx.x + blockIdx.y) & 7u) : (blockIdx.x & 7u);
  uint32_t vb = xr * chunk + (blockIdx.x >> 3);
  if (gridDim.y == 1u) {
    if (p.xcd_mapping == 0u) vb = blockIdx.x;
    else if (p.xcd_mapping == 2u && (chunk & 63u) == 0u) {
      const uint32_t li = blockIdx.x >> 3, band = (li >> 5) & 1u, in_band = (li >> 6) * 32u + (li & 31u);
      vb = band * (gridDim.x >> 1) + xr * (chunk >> 1) + in_band;
    }
  }
  uint32_t tile = (kQuad ? vb : (vb * 4u + wave));
  uint32_t pose = blockIdx.y;
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  if constexpr (!kMoments) {
    if (p.tile_order != nullptr) {   // a pose batch in world order: slot -> (pose, tile)
      // XCD x takes `granule` consecutive workgroups' worth of slots, then XCD x + 1 the next: what runs on an XCD at a time sees one
      // region of the map (its L2), and a hard region is shared by all eight (an eighth of the sorted list per XCD left the room
      // 7 % slower: its grazing tiles ended up on one XCD)
      const uint32_t G = p.tile_order_granule, li = blockIdx.x >> 3;
      const uint32_t sb = ((li / G) * 8u + (blockIdx.x & 7u)) * G + (li % G);
      if (sb >= p.n_tile_order) return;
      const uint32_t e = __builtin_amdgcn_readfirstlane(p.tile_order[sb]);   // one entry per workgroup: its pose and its (four) tiles
      pose = e >> 16;
      tile = kQuad ? (e & 0xFFFFu) : ((e & 0xFFFFu) * 4u + wave);
    }
  }
  if (tile >= ntiles) {
    if constexpr (kMoments) find_moments_idle_wave<kQuad>(p, s_mom_red, s_mom_piece, kQuad ? vb : (vb * 4u + wave), wave, threadIdx.x & 63u);
    return;
  }
  const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const uint32_t twl = p.tile_w_log2;
  const uint32_t lx = lane & ((1u << twl) - 1u), ly = lane >> twl;
  const uint32_t vid = (ty << (6u - twl)) + ly, hid = (tx << twl) + lx;
  const bool valid = (vid < p.H) && (hid < p.W);
  const uint32_t cv = valid ? vid : 0u, ch = valid ? hid : 0u;
  const uint32_t loc = cv * p.W + ch;

  MomDataset mom_ds = {mk3(0.f, 0.f, 0.f), false};
  if constexpr (kMoments) mom_ds = find_moments_dataset(p, valid && (!kQuad || sub == 0u), loc);
  xform Tsm, Tms;
  if (p.Tsm_arr != nullptr) { Tsm = p.Tsm_arr[pose]; Tms = p.Tms_arr[pose]; }
  else { Tsm = p.Tsm; Tms = p.Tms; }

  f3 dir_s, orig_s = p.orig_s;
  find_ray_s<kModel>(p, cv, ch, loc, dir_s, orig_s);
  const f3 org_m = (kModel == kModelSpherical || kModel == kModelPinhole) ? Tsm.t : xapply(Tsm, orig_s);
  const f3 dir_m = qrot(Tsm.R, dir_s);
  const bool finite = (dir_m.x == dir_m.x) && (dir_m.y == dir_m.y) && (dir_m.z == dir_m.z);
  const float ray_tfar = (valid && finite) ? p.tfar : -1.0f;

  uint32_t clk_trace0 = 0, clk_trace1 = 0, clk_visits = 0, clk_dbg[4] = {0u, 0u, 0u, 0u}, clk_descent = 0, clk_start = 0;
  uint32_t clk_stamps[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // kinds 31 and 32, clocked: frontier_descent_start's phase boundaries
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
            lds_dyn + lane, 64u, p.frontier_max_preload);
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
    RayHit coop_seed = {ray_tfar, kNone};   // kind 32: the closest hit among the leaves the wave tested together (frontier_descent_start)
    start.cur = (ray_tfar >= 0.0f) ? 0u : 0x7FFFFFFFu;
    start.sp = kWwStack ? 0u : 1u;
    const TraceStart* sp0 = &start;
    if constexpr (find_frontier(kTrav) && kModel != kModelOnDn) {   // (OnDn: one origin per ray, no common pyramid)
      if (p.tile_planes != nullptr) {     // (no table: the rays start at the root)
        const float* planes = p.tile_planes + static_cast<size_t>(__builtin_amdgcn_readfirstlane(tile)) * 16u;
        if (kTrav == 31 || kTrav == 32)
          start = frontier_descent_start<kFindBfRows, 1, kTrav == 32>(p.frontier, p.n_frontier, p.cnodes, p.cnodes16, p.scene_center, p.scene_half_diag, planes, Tsm.R, p.tfar,
                                                         org_m, dir_m, ray_tfar, lane, lds_dyn + threadIdx.x, kBfStride, p.frontier_max_preload,
                                                         lds_dyn + kFind31ListsAt + wave * kDescentWaveDwords, min(p.descent_final_cap, kDescentCap), p.descent_levels, kClock ? &clk_descent : nullptr,
                                                         kClock ? clk_stamps : nullptr, p.tris, &coop_seed);
        else if (kTrav == 23 || (kTrav >= 26 && kTrav <= 30))
          start = frontier_start<kFindBfRows, 1>(p.frontier, p.n_frontier, p.scene_center, p.scene_half_diag, planes, Tsm.R, p.tfar, org_m,
                                                 dir_m, ray_tfar, lane, lds_dyn + threadIdx.x, kBfStride, p.frontier_max_preload);
        else
          start = frontier_start<16, 0>(p.frontier, p.n_frontier, p.scene_center, p.scene_half_diag, planes, Tsm.R, p.tfar, org_m, dir_m,
                                        ray_tfar, lane, lds_dyn + threadIdx.x, blockDim.x, p.frontier_max_preload);
      }
    }
    if (kClock && p.wave_clock != nullptr) {   // the start (frontier cull / cooperative descent) ends here
      uint64_t t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
      clk_start = static_cast<uint32_t>(t);
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
    else if (kTrav == 16 || kTrav == 17 || kTrav == 19 || kTrav == 20 || kTrav == 23 || (kTrav >= 26 && kTrav <= 32))
      trace_lane_bf_tail<kFindBfRows, kTrav != 16 && kTrav != 20, find_leaf_trigger(kTrav), kTrav == 26, kTrav == 27, kTrav == 28, (kTrav == 29 ? 4 : (kTrav == 30 ? 3 : 5)), kTrav < 31>(kTrav == 26 ? p.qnodes : p.nodes, p.cnodes, p.tris, org_m, dir_m, ray_tfar, lds_dyn + threadIdx.x,
                                                   lds_dyn + kFindBfRows * 256u,
                                                   lds_dyn + kFindBfRows * 256u + kQuadStackEntries * 64u + (threadIdx.x >> 6) * (kTailRays * kTailXferDwords), h,
                                                   kClock ? &clk_visits : nullptr, sp0, kClock ? clk_dbg : nullptr, &pre_nrec, &pre_rec, (kTrav == 32) ? &coop_seed : nullptr);
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
  f3 mom_I = mk3(0.f, 0.f, 0.f), mom_N = mk3(0.f, 0.f, 0.f);   // kMoments: the correspondence as stored (sensor frame)
  if (valid) {
  const size_t g = static_cast<size_t>(pose) * p.W * p.H + loc;
  const bool found = (h.rec != kNone);
  // quad mode: the four lanes of a ray hold the same result and share the stores (0: hits/ranges/face ids, 1: points,
  // 2: normals)
  const bool w0 = !kQuad || sub == 0u, w1 = !kQuad || sub == 1u, w2 = !kQuad || sub == 2u;
  if (found) {
    if (p.hits && w0) p.hits[g] = 1;
    if (p.ranges && w0) p.ranges[g] = h.t;
    if ((p.points && w1) || kMoments) {
      f3 pt = scale3(dir_s, h.t);
      if (kModel == kModelO1Dn || kModel == kModelOnDn) pt = add3(pt, orig_s);
      if (p.points && w1) { p.points[3 * g] = pt.x; p.points[3 * g + 1] = pt.y; p.points[3 * g + 2] = pt.z; }
      mom_I = pt;
    }
    // the record's last 16 B: unit normal + the ORIGINAL face id
    if ((p.normals && w2) || (p.face_ids && w0) || kMoments) {
      uint4 nrec;
      if (kTrav == 27 && pre_rec == h.rec) nrec = pre_nrec;   // (rays finished by the quad tail fetch it here)
      else nrec = reinterpret_cast<const uint4*>(p.tris)[static_cast<size_t>(h.rec) * 4u + 3u];
      if ((p.normals && w2) || kMoments) {
        f3 n = qrot(Tms.R, mk3(asf(nrec.x), asf(nrec.y), asf(nrec.z)));
        if (dot_plain(dir_s, n) > 0.0f) n = neg3(n);  // flip towards the sensor
        if (p.normals && w2) { p.normals[3 * g] = n.x; p.normals[3 * g + 1] = n.y; p.normals[3 * g + 2] = n.z; }
        mom_N = n;
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
  if constexpr (kMoments)   // (after the stores have been issued: they complete under it)
    find_moments_wave<kQuad>(p, lds_dyn + (kQuad ? kQuadStackEntries * 64u : 0u), s_mom_red, s_mom_piece, valid && (h.rec != kNone) && (!kQuad || sub == 0u),
                             mom_ds, mom_I, mom_N, kQuad ? vb : (vb * 4u + wave), wave, threadIdx.x & 63u);
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
      // kinds 23 / 31: w[2] = cycles of the start (>> 4, 16 bits) | the wave's most leaf visits of a lane << 16 | most node visits << 24 instead of the realtime clock
      const uint32_t w2 = (kTrav == 23 || kTrav == 31 || kTrav == 32) ? (min((clk_start - clk_trace0) >> 4, 0xFFFFu) | (wmax(min(clk_dbg[3], 255u), 8) << 16) | (wmax(min(clk_visits, 255u), 8) << 24)) : clk_real;
      w[0] = clk_begin; w[1] = static_cast<uint32_t>(t); w[2] = w2; w[3] = (tile & 0xFFFFFFu) | (xcc << 24);
      w[4] = clk_trace0; w[5] = clk_trace1; w[6] = static_cast<uint32_t>(t2);
      w[7] = (kTrav == 31 || kTrav == 32) ? clk_descent : clk_w7;   // kinds 31 and 32: what its cooperative descent did (traverse.hip.h frontier_descent_start)
      if (kTrav == 31 || kTrav == 32) {   // ... and when: 16 more words per wave behind the table of all waves
        uint32_t* ws = p.wave_clock + 8u * (gridDim.y * gridDim.x * 4u) + 16u * ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + wave);
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) ws[i] = clk_stamps[i];
      }
    }
  }
}

}  // namespace
}  // namespace rmclhip
