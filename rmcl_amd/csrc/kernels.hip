// kernels.hip -- PRODUCTION kernels of librmclhip.so (gfx950, MI355X, CDNA4) for the RMCL / MICP-L hot path.
//
//  k_find            (find_kernel.hip.h) ray-casting correspondences, the kinds the product can select (0, 2, 23, 24)
//  k_cpc_find        (traverse.hip.h) closest-point correspondences, CPCEmbree::find
//  k_reduce_partials rm::statistics_p2l (CorrespondencesCPU.cpp:26-30; gate MICPSensorCPU.cpp:70-84)
//  k_micp_*          the inner iterations of MICPLocalizationNode::correctOnce
//                    (rmcl_ros/src/nodes/micp_localization.cpp:915-964) + rm::umeyama_transform
//  k_pf_update_v3    PCDSensorUpdater{Embree,Optix}::update, all beams fused
//                    (PCDSensorUpdaterEmbree.cpp:290-342, optix/BeamEvaluateProgram.cu:15-130)
//  k_pf_motion, k_gladiator_resample, k_likelihood_stats_*, k_pose_moments*: the rest of a filter cycle
//
// Experiments (rejected traversal kinds, probes, the round-2 particle-filter kernels) live in kernels_lab.hip and ship in
// librmclhip_lab.so; launch_find / launch_pf_update hand kinds they do not own to that library when it is loaded
// (lab_hooks.h), and report kLabMissing (kernels.h) otherwise.
#include "find_kernel.hip.h"
#include "lab_hooks.h"
#include "pf_common.hip.h"


namespace rmclhip {

namespace {

// Node16C twins (layout.h): thread = (node, entry)
__global__ void __launch_bounds__(256) k_build_cnodes16(const uint4* __restrict__ cn, uint32_t n_nodes, uint4* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t node = t >> 4, e = t & 15u, c = e >> 2, g = e & 3u;
  if (node >= n_nodes) return;
  const uint4 ca = cn[static_cast<size_t>(node) * 8u + 2u * c], cb = cn[static_cast<size_t>(node) * 8u + 2u * c + 1u];
  const uint4 far_a = uint4{__float_as_uint(kFarPoint[0]), __float_as_uint(kFarPoint[1]), __float_as_uint(kFarPoint[2]), __float_as_uint(kFarPoint[0])};
  const uint4 far_b = uint4{__float_as_uint(kFarPoint[1]), __float_as_uint(kFarPoint[2]), kLeafBit, 0u};
  uint4 oa = far_a, ob = far_b;
  const bool unused = __uint_as_float(ca.x) >= 1.0e29f;
  if (!unused) {
    if (cb.z & kLeafBit) {
      if (g == 0u) { oa = ca; ob = cb; }
    } else {
      oa = cn[static_cast<size_t>(cb.z) * 8u + 2u * g];
      ob = cn[static_cast<size_t>(cb.z) * 8u + 2u * g + 1u];
    }
  }
  out[static_cast<size_t>(node) * 32u + 2u * e] = oa;
  out[static_cast<size_t>(node) * 32u + 2u * e + 1u] = ob;
}

__global__ void k_compose_poses(const xform* __restrict__ Tbm, xform Tsb, xform* __restrict__ Tsm,
                                xform* __restrict__ Tms, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const xform T = xmul(Tbm[i], Tsb);
  Tsm[i] = T;
  Tms[i] = xinv(T);
}

// ---------------------------------------------------------------------------------------------
// statistics_p2l
// ---------------------------------------------------------------------------------------------
constexpr int kAcc = 16;  // sd[3] sm[3] smd[9] cnt

// wave64 sum of 16 doubles per lane through LDS instead of cross-lane shuffles: every lane stores its 16 values (row = lane),
// lane L adds column L & 15 over the 16 rows of slice L >> 4, two xor steps join the four slices.  The 17 shuffles of the
// halving butterfly are a chain of dependent ds_bpermute round trips (~3.4k clocks measured for a lone wave); here all stores
// and all loads are independent (~0.6k).  scratch: 64 x 17 doubles owned by this wave; LDS operations of one wave complete in
// order, so no barrier is needed.  Afterwards lanes 0..15 hold the totals of values 0..15.
__device__ __forceinline__ double wave_sum16_lds(const double (&v)[16], double* scratch, uint32_t lane) {
#pragma unroll
  for (int k = 0; k < 16; ++k) scratch[lane * 17u + static_cast<uint32_t>(k)] = v[k];
  const uint32_t col = lane & 15u, row0 = (lane >> 4) * 16u;
  double a[16];
#pragma unroll
  for (uint32_t r = 0; r < 16u; ++r) a[r] = scratch[(row0 + r) * 17u + col];
  double t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  t += ((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15]));
  t += __shfl_xor(t, 16, 64);
  t += __shfl_xor(t, 32, 64);
  return t;
}


// sum the per-block partials of one pose (one wave) and turn the raw moments into CrossStatistics
template <bool kAgentLoads = false>
__device__ __forceinline__ cstats finalize_pose(const double* partials, uint32_t nblocks) {
  // transposed reduction: lane = 16*g + k sums moment k over the blocks b = g, g+4, ... (16 lanes read one
  // 128-B partial: coalesced), then only TWO cross-lane steps (xor 16, 32) for one double per lane and 16
  // v_readlane broadcasts -- instead of 16 moments x 6 butterfly steps = 192 dependent ds_bpermute (measured
  // ~4.5 us of the 14 us k_micp_step)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t k0 = lane & 15u, g = lane >> 4;
  // 32 loads in flight per lane: every batch is one L2 round trip (~0.8 us) for this lone wave, so 256 partials
  // cost two round trips (8 in flight: 8 round trips, measured 6 us of the 11 us solve step; one load per
  // iteration serialised 64 round trips)
  double a = 0.0;
  uint32_t b = g;
  for (; b + 124u < nblocks; b += 128u) {
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const double* q = partials + static_cast<size_t>(b + 4u * u) * kAcc + k0;
      v[u] = kAgentLoads ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    }
#pragma unroll
    for (int u = 0; u < 32; u += 8) a += ((v[u] + v[u + 1]) + (v[u + 2] + v[u + 3])) + ((v[u + 4] + v[u + 5]) + (v[u + 6] + v[u + 7]));
  }
  for (; b + 28u < nblocks; b += 32u) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double* q = partials + static_cast<size_t>(b + 4u * u) * kAcc + k0;
      v[u] = kAgentLoads ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    }
    a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; b < nblocks; b += 4u) {
    const double* q = partials + static_cast<size_t>(b) * kAcc + k0;
    a += kAgentLoads ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
  }
  a += __shfl_xor(a, 16, 64);
  a += __shfl_xor(a, 32, 64);
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = __shfl(a, k, 64);
  cstats s = cs_identity();
  const double n = acc[15];
  if (n > 0.0) {
    const double md[3] = {acc[0] / n, acc[1] / n, acc[2] / n};
    const double mm[3] = {acc[3] / n, acc[4] / n, acc[5] / n};
    s.dataset_mean = mk3(static_cast<float>(md[0]), static_cast<float>(md[1]), static_cast<float>(md[2]));
    s.model_mean = mk3(static_cast<float>(mm[0]), static_cast<float>(mm[1]), static_cast<float>(mm[2]));
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) s.covariance[3 * r + c] = static_cast<float>(acc[6 + 3 * r + c] / n - mm[r] * md[c]);
    s.n_meas = static_cast<uint32_t>(n);
  }
  return s;
}


// one MICP inner iteration from the reduced statistics, executed by ONE lane
// micp_localization.cpp:915-964 for one sensor (merge_weight_multiplier == 1):
//   Cs_b = Tsb * stats_s (MICPSensor.hpp:182); Cs_o = Tbo * Cs_b (:931); Cmerged = Identity += Cs_o (:936)
//   T_inner = umeyama(Cmerged) (:952); T_onew_oold = T_onew_oold * T_inner (:963)
//   next T_bnew_bold = ~Tbo * T_onew_oold * Tbo (:926); T_snew_sold = ~Tsb * T_bnew_bold * Tsb (MICPSensor.hpp:178)
__device__ __forceinline__ void micp_advance(const cstats& stats_s, const xform& Tsb, const xform& Tbo, MicpState* st) {
  const cstats Cs_b = cs_transform(Tsb, stats_s);
  const cstats Cs_o = cs_transform(Tbo, Cs_b);
  const cstats Cmerged = cs_merge(cs_identity(), Cs_o);
  const xform T_inner = umeyama(Cmerged);
  const xform T_onew_oold = xmul(st->T_onew_oold, T_inner);
  const xform T_bnew_bold = xmul(xmul(xinv(Tbo), T_onew_oold), Tbo);
  st->T_onew_oold = T_onew_oold;
  st->T_snew_sold = xmul(xmul(xinv(Tsb), T_bnew_bold), Tsb);
  st->stats_o = Cmerged;
}

// The same loop for ONE sensor, written in the sensor frame.  umeyama is equivariant under a rigid change of frame
// (umeyama(T * C) = T o umeyama(C) o T^-1 for a transform T applied to both means and the covariance), so with
// Tso = Tbo * Tsb:  T_inner = Tso U Tso^-1 with U = umeyama(stats_s), and
//   T_snew_sold' = Tso^-1 (T_onew_oold T_inner) Tso = T_snew_sold * U.
// The loop therefore only needs U and one product per iteration; T_onew_oold = Tso T_snew_sold Tso^-1 and
// stats_o = Tbo * (Tsb * stats_s) are formed once, after the last iteration (micp_close_sensor).  Same mathematics as
// micp_advance, ~700 instead of ~2500 dependent operations per iteration for the lone lane that runs it; the rounding
// differs from the frame-by-frame order at the 1e-7 level (tests: 1e-5 against the oracle's frame-by-frame loop).
__device__ __forceinline__ void micp_advance_sensor(const cstats& stats_s, xform* T_snew_sold) {
  *T_snew_sold = xmul(*T_snew_sold, umeyama(stats_s));
}
__device__ __forceinline__ void micp_close_sensor(const cstats& stats_s_last, const xform& T_snew_sold, const xform& Tsb,
                                                  const xform& Tbo, MicpState* st) {
  const xform Tso = xmul(Tbo, Tsb);
  st->T_snew_sold = T_snew_sold;
  st->T_onew_oold = xmul(xmul(Tso, T_snew_sold), xinv(Tso));
  st->stats_o = cs_merge(cs_identity(), cs_transform(Tbo, cs_transform(Tsb, stats_s_last)));
}

// kTail == kTailNone keeps the streaming kernel lean (the solve code of the fused tails costs registers and
// scratch: with it compiled in, this kernel went from 96 to 192 VGPRs + 80 B scratch and 2.3x slower launches)
template <uint32_t kTail>
__global__ void __launch_bounds__(256) k_reduce_partials(const ReduceParams p) {
  __shared__ double red[4][kAcc];
  __shared__ double s_wsum[4][64 * 17];
  const uint32_t pose = blockIdx.y;
  const xform Tpre = (p.Tpre_dev != nullptr) ? p.Tpre_dev[pose] : p.Tpre;
  const float max_dist = (p.call != nullptr) ? p.call->max_dist : p.max_dist;
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
  const size_t mbase = static_cast<size_t>(pose) * p.n;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < p.n; i += gridDim.x * 256u) {
    const bool dok = (p.dataset_mask == nullptr) || (p.dataset_mask[i] > 0);
    if (dok && (p.model_mask == nullptr || p.model_mask[mbase + i] > 0)) {   // (null: rmclhip_statistics_p2l on a view without a mask)
      const float* dp = p.dataset_points + 3 * static_cast<size_t>(i);
      const float* mp = p.model_points + 3 * (mbase + i);
      const float* mn = p.model_normals + 3 * (mbase + i);
      const f3 Di = xapply(Tpre, mk3(dp[0], dp[1], dp[2]));
      const f3 Ii = mk3(mp[0], mp[1], mp[2]);
      const f3 Ni = mk3(mn[0], mn[1], mn[2]);
      const float spd = dot_plain(sub3(Ii, Di), Ni);
      if (fabsf(spd) < max_dist) {
        const f3 Mi = add3(Di, scale3(Ni, spd));
        const double d[3] = {Di.x, Di.y, Di.z}, m[3] = {Mi.x, Mi.y, Mi.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) { acc[k] += d[k]; acc[3 + k] += m[k]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[6 + 3 * r + c] += m[r] * d[c];
        acc[15] += 1.0;
      }
    }
  }
  // wave64 reduction of the 16 moments through LDS (wave_sum16_lds: round 1 used a halving butterfly of 17 dependent
  // double shuffles, ~3.4k clocks for the last wave of a block; the transposed sum has no dependent round trips)
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  {
    const double wsum = wave_sum16_lds(acc, &s_wsum[wave][0], lane);
    if (lane < 16u) red[wave][lane] = wsum;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    const double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    // agent-scope relaxed store = write-through (sc1) 8-B store: visible to the last arriver without an L2
    // write-back fence (MI355X_MICROARCH.md, hand-off forms)
    __hip_atomic_store(p.partials + (static_cast<size_t>(pose) * p.nblocks + blockIdx.x) * kAcc + threadIdx.x, v,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (kTail != kTailNone) {
  // ---- fused tail: the LAST block of this pose to arrive finalizes.  Hand-off without fences: write-through
  // (sc1) partial stores -> every wave s_waitcnt vmcnt(0) -> barrier -> relaxed agent-scope ticket; the last
  // arriver reads the partials with sc1 loads (L1-bypassing).  A release fence per block (256 x buffer_wbl2)
  // measured SLOWER than the kernel boundary it replaces (reduce 14.5 us vs 10.7 us for two launches).
  __shared__ uint32_t s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t ticket = __hip_atomic_fetch_add(p.tickets + pose, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (ticket == gridDim.x - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (s_last == 0u) return;
  if (threadIdx.x >= 64u) return;
  const cstats st = finalize_pose<true>(p.partials + static_cast<size_t>(pose) * p.nblocks * kAcc, p.nblocks);
  if (threadIdx.x == 0) {
    p.tickets[pose] = 0u;  // re-armed for the next launch on this stream
    if (kTail == kTailStats) {
      p.stats_out[pose] = st;
    } else if (kTail == kTailMicp) {
      micp_advance(st, p.call ? p.call->Tsb : p.Tsb, p.call ? p.call->Tbo : p.Tbo, p.state);
    } else {  // kTailBatchSolve: v1 corrector, Tdelta_b = Tsb * T_s * ~Tsb
      const xform Ts = umeyama(st);
      p.Tdelta_out[pose] = xmul(xmul(p.Tsb, Ts), xinv(p.Tsb));
      if (p.stats_out) p.stats_out[pose] = st;
    }
  }
  }  // kTail != kTailNone
}

// Completion tag of a launch chain whose results go to host-mapped memory and whose caller polls instead of waiting for the
// stream: ONE 8-byte store {seq, xor of every result word}, issued after the results and a system-scope fence.  The SEQUENCE NUMBER
// is what makes the hand-off sound: round 2's form polled a flag the host itself had cleared before the launch -- the same value
// every call -- and about 1 call in 10^4 took the previous call's results (tools/determinism2.py).  Round 5 isolated the mechanism
// (tools/ubench/tag_handoff.hip, tools/tag_retries.py, profiles/r05_tag_handoff.txt): with a per-call value the device's
// "results, __threadfence_system(), tag" order has never been seen violated (6 x 10^6 isolated hand-offs across allocations and
// pinning flags, 9 x 10^5 product calls, not one checksum rejection) -- the failure belonged to the reused flag, not to the store
// order.  The xor stays as a belt: the host accepts a result only when the tag carries this call's sequence number AND the words it
// reads add up to the tag's sum (capi_rcc.cpp wait_done), and keeps polling otherwise.
template <typename Tp>
__device__ __forceinline__ uint32_t xor_words(const Tp& v) {
  static_assert(sizeof(Tp) % 4 == 0, "word-sized results only");
  uint32_t w[sizeof(Tp) / 4];
  __builtin_memcpy(w, &v, sizeof(Tp));
  uint32_t x = 0;
#pragma unroll
  for (uint32_t i = 0; i < sizeof(Tp) / 4; ++i) x ^= w[i];
  return x;
}
__device__ __forceinline__ void publish_tag(unsigned long long* tag, uint32_t seq, uint32_t sum) {
  __threadfence_system();
  __hip_atomic_store(tag, (static_cast<unsigned long long>(sum) << 32) | static_cast<unsigned long long>(seq), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(64) k_reduce_finalize(const double* __restrict__ partials, uint32_t nblocks,
                                                       cstats* __restrict__ out, unsigned long long* done, uint32_t seq) {
  const uint32_t pose = blockIdx.x;
  const cstats s = finalize_pose(partials + static_cast<size_t>(pose) * nblocks * kAcc, nblocks);
  if (threadIdx.x == 0) {
    out[pose] = s;
    if (done) publish_tag(done, seq, xor_words(s));   // single-pose call with a host-mapped result (capi_rcc.cpp wait_done)
  }
}

// ---------------------------------------------------------------------------------------------
// persistent MICP loop: ALL optimization iterations of one correction in ONE launch.
// The per-iteration chain "reduce (grid) -> finalize + solve (one lane) -> next pre-transform" is bound by launch
// boundaries (two ~5 us launches per iteration for ~1 us of HBM/L2 streaming).  Here a small co-resident grid
// (gridDim.x <= number of CUs, 512 threads) keeps iterating: each block reduces its contiguous slice of the
// correspondences (which stay L2-resident), publishes a 128-B partial with write-through stores, meets the other
// blocks at a counter barrier, and then EVERY block finalizes and solves redundantly (same partials, same
// order, same code => identical state) -- one barrier per iteration and no broadcast step.
// ---------------------------------------------------------------------------------------------
struct MicpLoopParams {
  const float* dataset_points;
  const uint8_t* dataset_mask;  // nullable
  const float* model_points;
  const float* model_normals;
  const uint8_t* model_mask;
  uint32_t n, n_iter;
  const MicpCall* call;  // Tsb, Tbo, max_dist of this correction
  double* partials;      // [2][gridDim.x][16] (double-buffered across iterations)
  uint32_t* barrier;     // zero at launch (k_micp_init)
  MicpState* state;      // result, written by block 0
};

// kOneXcd: the grid is 8x larger and only the blocks the dispatcher places on XCD 0 (block id % 8 == 0: observed placement,
// used for speed only -- the barrier protocol is valid wherever the blocks land) take part, so that partials, counter and
// correspondences meet in ONE L2 instead of crossing the fabric between eight
template <bool kOneXcd>
__global__ void __launch_bounds__(512) k_micp_loop(const MicpLoopParams p) {
  __shared__ double red[8][kAcc];
  __shared__ MicpState s_state;
  __shared__ xform s_Tsb, s_Tbo;
  if (kOneXcd && (blockIdx.x & 7u) != 0u) return;
  const uint32_t G = kOneXcd ? (gridDim.x >> 3) : gridDim.x, b = kOneXcd ? (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    s_state.T_onew_oold = xidentity();
    s_state.T_snew_sold = xidentity();
    s_state.stats_o = cs_identity();
    s_Tsb = p.call->Tsb;
    s_Tbo = p.call->Tbo;
  }
  const float max_dist = p.call->max_dist;
  // contiguous slice of this block, a multiple of the block size
  const uint32_t per = ((p.n + G - 1u) / G + 511u) & ~511u;
  const uint32_t i0 = b * per, i1 = min(p.n, i0 + per);
  __syncthreads();
  for (uint32_t it = 0; it < p.n_iter; ++it) {
    const xform Tpre = s_state.T_snew_sold;
    double acc[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 512u) {
      const bool dok = (p.dataset_mask == nullptr) || (p.dataset_mask[i] > 0);
      if (dok && p.model_mask[i] > 0) {
        const float* dp = p.dataset_points + 3 * static_cast<size_t>(i);
        const float* mp = p.model_points + 3 * static_cast<size_t>(i);
        const float* mn = p.model_normals + 3 * static_cast<size_t>(i);
        const f3 Di = xapply(Tpre, mk3(dp[0], dp[1], dp[2]));
        const f3 Ii = mk3(mp[0], mp[1], mp[2]);
        const f3 Ni = mk3(mn[0], mn[1], mn[2]);
        const float spd = dot_plain(sub3(Ii, Di), Ni);
        if (fabsf(spd) < max_dist) {
          const f3 Mi = add3(Di, scale3(Ni, spd));
          const double d[3] = {Di.x, Di.y, Di.z}, m[3] = {Mi.x, Mi.y, Mi.z};
#pragma unroll
          for (int k = 0; k < 3; ++k) { acc[k] += d[k]; acc[3 + k] += m[k]; }
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[6 + 3 * r + c] += m[r] * d[c];
          acc[15] += 1.0;
        }
      }
    }
    // wave reduction (halving butterfly, see k_reduce_partials), then the 8 waves through LDS
#pragma unroll
    for (int half = 8, off = 32; half >= 1; half >>= 1, off >>= 1) {
      const bool hi = (lane & static_cast<uint32_t>(off)) != 0u;
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const double send = hi ? acc[j] : acc[j + half];
        const double keep = hi ? acc[j + half] : acc[j];
        acc[j] = keep + __shfl_xor(send, off, 64);
      }
    }
    acc[0] += __shfl_xor(acc[0], 2, 64);
    acc[0] += __shfl_xor(acc[0], 1, 64);
    if ((lane & 3u) == 0u) red[wave][lane >> 2] = acc[0];
    __syncthreads();
    double* part = p.partials + static_cast<size_t>(it & 1u) * G * kAcc;
    if (threadIdx.x < kAcc) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[w][threadIdx.x];
      __hip_atomic_store(part + static_cast<size_t>(b) * kAcc + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // grid barrier: write-through partial stores complete (vmcnt) -> arrive -> spin on the monotonic counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(p.barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t target = (it + 1u) * G;
      while (__hip_atomic_load(p.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (wave == 0u) {
      const cstats st = finalize_pose<true>(part, G);
      if (lane == 0u) micp_advance(st, s_Tsb, s_Tbo, &s_state);
    }
    __syncthreads();
  }
  if (b == 0u && threadIdx.x == 0u) *p.state = s_state;
}

// One MICP iteration per launch (instead of reduce + solve = two): the launch of iteration i first finishes
// iteration i-1 -- wave 0 of EVERY block sums the previous partials and solves redundantly (same inputs, same
// order => the same pre-transform in every block; block 0 records the advanced state) -- and then streams the
// correspondences with that pre-transform.  Partials and state ping-pong between two buffers so that no block
// reads what another block of the same launch writes.
struct MicpIterParams {
  const float* dataset_points;
  const uint8_t* dataset_mask;  // nullable
  const float* model_points;
  const float* model_normals;
  const uint8_t* model_mask;
  uint32_t n, nblocks;
  const MicpCall* call;
  const double* partials_prev;  // of the previous launch (unused when first)
  double* partials_out;
  const MicpState* state_in;    // state before finishing the previous iteration
  MicpState* state_out;
  uint32_t first;
};

__global__ void __launch_bounds__(256) k_micp_iter(const MicpIterParams p) {
  __shared__ double red[4][kAcc];
  __shared__ double s_wsum[4][64 * 17];
  __shared__ xform s_Tpre;
  // The correspondences of this thread do not depend on the pre-transform the prologue is about to compute: request the
  // first two elements (all a thread gets at reduce_num_blocks' 512 elements per block) BEFORE the prologue, so that
  // their load latency hides behind the finalize + solve of wave 0 instead of following it.
  constexpr int kPre = 2;
  float pd[kPre][3], pm[kPre][3], pn[kPre][3];
  bool pok[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x + static_cast<uint32_t>(u) * gridDim.x * 256u;
    pok[u] = false;
    if (i < p.n) {
      const bool dok = (p.dataset_mask == nullptr) || (p.dataset_mask[i] > 0);
      pok[u] = dok && p.model_mask[i] > 0;
      const float* dp = p.dataset_points + 3 * static_cast<size_t>(i);
      const float* mp = p.model_points + 3 * static_cast<size_t>(i);
      const float* mn = p.model_normals + 3 * static_cast<size_t>(i);
#pragma unroll
      for (int k = 0; k < 3; ++k) { pd[u][k] = dp[k]; pm[u][k] = mp[k]; pn[u][k] = mn[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) { pd[u][k] = 0.f; pm[u][k] = 0.f; pn[u][k] = 0.f; }
    }
  }
  if (threadIdx.x < 64u) {
    if (p.first) {
      if (threadIdx.x == 0) {
        s_Tpre = xidentity();
        if (blockIdx.x == 0) {  // the state before any iteration (replaces a separate init launch)
          MicpState init;
          init.T_onew_oold = xidentity();
          init.T_snew_sold = xidentity();
          init.stats_o = cs_identity();
          *p.state_out = init;
        }
      }
    } else {
      const cstats st = finalize_pose(p.partials_prev, p.nblocks);
      if (threadIdx.x == 0) {
        // sensor-frame form of the iteration (micp_advance_sensor): the odom-frame quantities are formed by the closing
        // launch (k_micp_close); between launches the state carries T_snew_sold only
        xform T_s = p.state_in->T_snew_sold;
        micp_advance_sensor(st, &T_s);
        s_Tpre = T_s;
        if (blockIdx.x == 0) p.state_out->T_snew_sold = T_s;
      }
    }
  }
  __syncthreads();
  const xform Tpre = s_Tpre;
  const float max_dist = p.call->max_dist;
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
#define RMCL_P2L_ACCUMULATE(DX, DY, DZ, IX, IY, IZ, NX, NY, NZ)                      \
  {                                                                                 \
    const f3 Di = xapply(Tpre, mk3(DX, DY, DZ));                                    \
    const f3 Ii = mk3(IX, IY, IZ);                                                  \
    const f3 Ni = mk3(NX, NY, NZ);                                                  \
    const float spd = dot_plain(sub3(Ii, Di), Ni);                                  \
    if (fabsf(spd) < max_dist) {                                                    \
      const f3 Mi = add3(Di, scale3(Ni, spd));                                      \
      const double d[3] = {Di.x, Di.y, Di.z}, m[3] = {Mi.x, Mi.y, Mi.z};            \
      _Pragma("unroll") for (int k = 0; k < 3; ++k) { acc[k] += d[k]; acc[3 + k] += m[k]; } \
      _Pragma("unroll") for (int r = 0; r < 3; ++r)                                 \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) acc[6 + 3 * r + c] += m[r] * d[c]; \
      acc[15] += 1.0;                                                               \
    }                                                                               \
  }
#pragma unroll
  for (int u = 0; u < kPre; ++u)
    if (pok[u]) RMCL_P2L_ACCUMULATE(pd[u][0], pd[u][1], pd[u][2], pm[u][0], pm[u][1], pm[u][2], pn[u][0], pn[u][1], pn[u][2])
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x + static_cast<uint32_t>(kPre) * gridDim.x * 256u; i < p.n; i += gridDim.x * 256u) {
    const bool dok = (p.dataset_mask == nullptr) || (p.dataset_mask[i] > 0);
    if (dok && p.model_mask[i] > 0) {
      const float* dp = p.dataset_points + 3 * static_cast<size_t>(i);
      const float* mp = p.model_points + 3 * static_cast<size_t>(i);
      const float* mn = p.model_normals + 3 * static_cast<size_t>(i);
      RMCL_P2L_ACCUMULATE(dp[0], dp[1], dp[2], mp[0], mp[1], mp[2], mn[0], mn[1], mn[2])
    }
  }
#undef RMCL_P2L_ACCUMULATE
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  {
    const double wsum = wave_sum16_lds(acc, &s_wsum[wave][0], lane);
    if (lane < 16u) red[wave][lane] = wsum;
  }
  __syncthreads();
  if (threadIdx.x < kAcc)
    p.partials_out[static_cast<size_t>(blockIdx.x) * kAcc + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void k_micp_init(MicpState* st, uint32_t* barrier) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->T_onew_oold = xidentity();
    st->T_snew_sold = xidentity();
    st->stats_o = cs_identity();
    st[1] = st[0];  // the state ping-pongs between two slots (k_micp_iter)
    if (barrier) *barrier = 0u;
  }
}

// unfused form of the MICP step (kept for A/B against the fused tail of k_reduce_partials)
// st_out may alias st (in place) or point to host-mapped memory: the closing step of a correction then delivers the
// result without a device-to-host copy node
// closing launch of the one-launch-per-iteration form: the last iteration's solve + the odom-frame results
__global__ void __launch_bounds__(64) k_micp_close(const double* __restrict__ partials, uint32_t nblocks, const MicpCall* call,
                                                   const MicpState* st, MicpState* st_out, unsigned long long* done) {
  const cstats stats_s = finalize_pose(partials, nblocks);
  if (threadIdx.x == 0) {
    xform T_s = st->T_snew_sold;
    micp_advance_sensor(stats_s, &T_s);
    MicpState out;
    micp_close_sensor(stats_s, T_s, call->Tsb, call->Tbo, &out);
    *st_out = out;
    if (done) publish_tag(done, call->seq, xor_words(out));   // the caller polls the tag instead of waiting for the stream's signal
  }
}

__global__ void __launch_bounds__(64) k_micp_step(const double* __restrict__ partials, uint32_t nblocks, xform Tsb,
                                                  xform Tbo, const MicpCall* call, const MicpState* st, MicpState* st_out) {
  const cstats stats_s = finalize_pose(partials, nblocks);
  if (threadIdx.x == 0) {
    MicpState local = *st;
    micp_advance(stats_s, call ? call->Tsb : Tsb, call ? call->Tbo : Tbo, &local);
    *st_out = local;
  }
}

// ---------------------------------------------------------------------------------------------
// Gate-stable moment form of the MICP-L inner loop, schedule (R) (micp_localization.cpp:900-964: ONE find, then n_iter x
// (statistics_p2l + umeyama) over the SAME correspondences).  Between iterations only the pre-transform T = (R, t) changes:
//   D' = R D + t,  dist = N.I - N.D',  M = D' + N dist,  gate |dist| < max_dist           (MICPSensorCPU.cpp:70-84)
// so for a FIXED set of gated-in correspondences the 16 raw sums of the reduction (sum D', sum M, sum M D'^T, n) are
// polynomials in (R, t) whose coefficients are 82 moments of (D, N, s = N.I):
//   sum D, sum D D^T, sum s N, sum s N D^T, sum N N^T, sum N_a N_b D_j, sum N_a N_b D_j D_k.
// The gate is the only non-polynomial part.  A correspondence whose |dist| at the first pre-transform (identity) is
// farther from max_dist than the farthest its point can move, |D' - D| <= rho |D| + tau (rho = 2 sin(theta/2), tau = |t|),
// keeps its gate decision for every iteration whose pre-transform stays within (rho_cap, tau_cap): its contribution comes
// from the moments.  The others ("uncertain", normally a few hundred) are re-evaluated every iteration with the reduction's
// own f32 arithmetic.  One streaming pass (k_micp_moments) + ONE single-workgroup launch for all iterations
// (k_micp_fast_loop) replace n_iter streaming launches.  If a pre-transform leaves the caps or more than
// kFastMaxUncertain correspondences are uncertain, the loop reports it and the caller runs the per-iteration form instead:
// the result never depends on the caps.
// ---------------------------------------------------------------------------------------------
constexpr int kMom = 96;  // 82 used: n | D[3] | DD[6] | sN[3] | sND[9] | NN[6] | NND[18] | NNDD[36]
constexpr uint32_t kFastMaxUncertain = 4096;
// Bound of the device-side polls below (fold flags of sibling workgroups, join flags of another stream's signal kernel): a poll is
// one L2 round trip (~1 us), so ~2 s.  A producer that never arrives ends the launch with status code 2 -- the host then takes the
// per-iteration form, which joins with stream events -- instead of a kernel the 20 ms host fallback could never get past.
constexpr uint32_t kDevicePollBound = 1u << 21;
constexpr uint32_t kFastThreads = 256;   // 1 wave per SIMD: the one-lane solve may use up to 512 VGPRs (no scratch)

// wave64 sum of 16 doubles per lane by the halving butterfly of k_reduce_partials: afterwards lane L holds the wave total of
// value L >> 2 in v[0]
__device__ __forceinline__ void wave_reduce16(double (&v)[16], uint32_t lane) {
#pragma unroll
  for (int half = 8, off = 32; half >= 1; half >>= 1, off >>= 1) {
    const bool hi = (lane & static_cast<uint32_t>(off)) != 0u;
#pragma unroll
    for (int j = 0; j < half; ++j) {
      const double send = hi ? v[j] : v[j + half];
      const double keep = hi ? v[j + half] : v[j];
      v[j] = keep + __shfl_xor(send, off, 64);
    }
  }
  v[0] += __shfl_xor(v[0], 2, 64);
  v[0] += __shfl_xor(v[0], 1, 64);
}

// raw sums of the reduction (sd[3] sm[3] smd[9] n) -> CrossStatistics, as finalize_pose does
// ---- cheaper reciprocals for the ONE lane that solves (moment-form loops): a lone lane retires one instruction per ~5-8 cycles,
// so the solve costs what its instruction count costs, and an IEEE f64 division is ~25 instructions, a square root + division
// ~50.  v_rcp_f64 / v_rsq_f64 with two Newton-Raphson refinements give 1/x and 1/sqrt(x) to ~1 ulp in 5 / 9 instructions.  Used
// where the consumer is itself an iteration (Newton's step of the quartic) or is rounded to f32 afterwards (quaternion
// normalisation, 1/n of the statistics); the generic umeyama() of devmath.h -- shared with the host and the oracle-facing
// entry points -- keeps IEEE divisions.
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double rsq_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x, y * y, 1.5);
  y = y * fma(-0.5 * x, y * y, 1.5);
  return y;
}

// horn_quaternion (devmath.h) with rcp_nr / rsq_nr; same formulas, same fall-back condition (false -> the caller takes umeyama())
__device__ __forceinline__ bool horn_quaternion_fast(const double* C, double* q) {
  const double Sxx = C[0], Sxy = C[3], Sxz = C[6], Syx = C[1], Syy = C[4], Syz = C[7], Szx = C[2], Szy = C[5], Szz = C[8];
  sym4 K;
  K.k00 = Sxx + Syy + Szz; K.k01 = Syz - Szy; K.k02 = Szx - Sxz; K.k03 = Sxy - Syx;
  K.k11 = Sxx - Syy - Szz; K.k12 = Sxy + Syx; K.k13 = Szx + Sxz;
  K.k22 = -Sxx + Syy - Szz; K.k23 = Syz + Szy;
  K.k33 = -Sxx - Syy + Szz;
  const double ss = ((Sxx * Sxx + Sxy * Sxy + Sxz * Sxz) + (Syx * Syx + Syy * Syy + Syz * Syz)) + (Szx * Szx + Szy * Szy + Szz * Szz);
  if (!(ss > 0.0)) return false;
  const double c2 = -2.0 * ss;
  const double c1 = -8.0 * det3(C);
  const double c0 = sym4_det(sym4_sub(K));
  const double lam0 = sqrt(3.0 * ss) * (1.0 + 1e-12);
  double lam = lam0;
  {
    const double a = K.k00, a2 = a * a;
    const double Pa = (a2 + c2) * a2 + c1 * a + c0, dPa = (4.0 * a2 + 2.0 * c2) * a + c1;
    if (a > 0.0 && 3.0 * a2 > ss && dPa > 0.0 && Pa <= 0.0) {
      const double a1 = a - Pa * rcp_nr(dPa);
      if (a1 < lam0) lam = a1;
    }
  }
  bool converged = false;
  for (int it = 0; it < 60; ++it) {
    const double l2 = lam * lam;
    const double P = (l2 + c2) * l2 + c1 * lam + c0;
    const double dP = (4.0 * l2 + 2.0 * c2) * lam + c1;
    if (!(dP > 0.0)) break;
    const double step = P * rcp_nr(dP);
    lam -= step;
    if (step <= 1e-14 * lam0) { converged = true; break; }
  }
  if (!converged) return false;
  K.k00 -= lam; K.k11 -= lam; K.k22 -= lam; K.k33 -= lam;
  const sym4_minors m = sym4_sub(K);
  const double a00 = K.k11 * m.c5 - K.k12 * m.c4 + K.k13 * m.c3;
  const double a11 = K.k00 * m.c5 - K.k02 * m.c2 + K.k03 * m.c1;
  const double a22 = K.k03 * m.s4 - K.k13 * m.s2 + K.k33 * m.s0;
  const double a33 = K.k02 * m.s3 - K.k12 * m.s1 + K.k22 * m.s0;
  const double a01 = -K.k01 * m.c5 + K.k02 * m.c4 - K.k03 * m.c3;
  const double a02 = K.k13 * m.s5 - K.k23 * m.s4 + K.k33 * m.s3;
  const double a03 = -K.k12 * m.s5 + K.k22 * m.s4 - K.k23 * m.s3;
  const double a12 = -K.k03 * m.s5 + K.k23 * m.s2 - K.k33 * m.s1;
  const double a13 = K.k02 * m.s5 - K.k22 * m.s2 + K.k23 * m.s1;
  const double a23 = -K.k02 * m.s4 + K.k12 * m.s2 - K.k23 * m.s0;
  double v0 = a00, v1 = a01, v2 = a02, v3 = a03, dbest = fabs(a00);
  if (fabs(a11) > dbest) { v0 = a01; v1 = a11; v2 = a12; v3 = a13; dbest = fabs(a11); }
  if (fabs(a22) > dbest) { v0 = a02; v1 = a12; v2 = a22; v3 = a23; dbest = fabs(a22); }
  if (fabs(a33) > dbest) { v0 = a03; v1 = a13; v2 = a23; v3 = a33; dbest = fabs(a33); }
  if (!(dbest > 1e-10 * lam0 * lam0 * lam0)) return false;  // repeated largest eigenvalue
  const double n2 = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
  if (!(n2 > 0.0)) return false;
  const double rn = rsq_nr(n2);
  double w = v0 * rn, x = v1 * rn, y = v2 * rn, z = v3 * rn;
  double lead = w;
  if (!(fabs(w) > 0.5)) {
    if (x * x > y * y && x * x > z * z) lead = x;
    else if (y * y > z * z) lead = y;
    else lead = z;
  }
  if (lead < 0.0) { w = -w; x = -x; y = -y; z = -z; }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
  return true;
}

// umeyama() for the moment-form loops' one solving lane; degenerate inputs take the generic path
__device__ __forceinline__ xform umeyama_fast(const cstats& s) {
  if (s.n_meas == 0) return xidentity();
  double C[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = static_cast<double>(s.covariance[i]);
  double q[4];
  if (!horn_quaternion_fast(C, q)) return umeyama(s);
  xform T = xidentity();
  T.R.x = static_cast<float>(q[0]); T.R.y = static_cast<float>(q[1]);
  T.R.z = static_cast<float>(q[2]); T.R.w = static_cast<float>(q[3]);
  T.t = sub3(s.model_mean, qrot(T.R, s.dataset_mean));
  return T;
}

__device__ __forceinline__ cstats cstats_from_sums(const double* acc) {
  cstats s = cs_identity();
  const double n = acc[15];
  if (n > 0.0) {
    const double rn = rcp_nr(n);   // (finalize_pose divides 15 times: same values to the last bit or two)
    const double md[3] = {acc[0] * rn, acc[1] * rn, acc[2] * rn};
    const double mm[3] = {acc[3] * rn, acc[4] * rn, acc[5] * rn};
    s.dataset_mean = mk3(static_cast<float>(md[0]), static_cast<float>(md[1]), static_cast<float>(md[2]));
    s.model_mean = mk3(static_cast<float>(mm[0]), static_cast<float>(mm[1]), static_cast<float>(mm[2]));
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) s.covariance[3 * r + c] = static_cast<float>(acc[6 + 3 * r + c] * rn - mm[r] * md[c]);
    s.n_meas = static_cast<uint32_t>(n);
  }
  return s;
}

struct MicpFastParams {
  const float* dataset_points;
  const uint8_t* dataset_mask;  // nullable
  const float* model_points;
  const float* model_normals;
  const uint8_t* model_mask;
  uint32_t n, nblocks;          // nblocks: grid of k_micp_moments
  const MicpCall* call;
  double* partials;             // [nblocks][kMom]
  unsigned long long* unc_mask; // [ceil(n / 64)]: bit i%64 of word i/64 = correspondence i is uncertain
  uint32_t n_iter;
  MicpState* state_out;         // may be host-mapped
  MicpFastStatus* status;       // may be host-mapped
  unsigned long long* done;     // host-mapped completion tag
  MicpCallLite cv;              // used when call == nullptr (direct launches)
  // mask words in the tile order of the find that produced them (launch_find_moments): word t = (virtual) tile t, bit l = lane l
  uint32_t mask_tiled, mask_W, mask_tiles_x, mask_tile_w_log2, mask_nwords;
  // k_micp_fast_loop launched as gridDim.x > 1 workgroups: workgroup b folds its share of the partial rows; b > 0 hands its 82 sums to
  // workgroup 0 through fold_rows[b] / fold_flags[b] (= this call's sequence number) and leaves
  double* fold_rows;            // [kMicpFoldBlocks][kMom]
  uint32_t* fold_flags;         // [kMicpFoldBlocks]
  MicpHostBlock* host_block;    // k_micp_publish: pinned, host-mapped
};
// correspondence index of bit `b` of mask word `w`
__device__ __forceinline__ uint32_t micp_mask_index(const MicpFastParams& p, uint32_t w, uint32_t b) {
  if (p.mask_tiled == 0u) return (w << 6) + b;
  const uint32_t ty = w / p.mask_tiles_x, tx = w - ty * p.mask_tiles_x, twl = p.mask_tile_w_log2;
  const uint32_t vid = (ty << (6u - twl)) + (b >> twl), hid = (tx << twl) + (b & ((1u << twl) - 1u));
  return vid * p.mask_W + hid;
}
#define RMCL_FCALL(p, field) ((p).call != nullptr ? (p).call->field : (p).cv.field)

// Status blocks live in pinned host memory: the block is written whole, then the completion tag (publish_tag) with the sum of
// the block and of whatever else this exit wrote for the host (`extra`: the xor of the state block, 0 for the early exits).
__device__ __forceinline__ void publish_status(MicpFastStatus* dst, const MicpFastStatus& st, unsigned long long* tag, uint32_t seq,
                                               uint32_t extra) {
  MicpFastStatus body = st;
  body.pad[2] = 0u;
  *dst = body;                      // two 16-B stores
  publish_tag(tag, seq, xor_words(body) ^ extra);
}
__device__ __forceinline__ void publish_status(MicpMultiFastStatus* dst, const MicpMultiFastStatus& st, unsigned long long* tag,
                                               uint32_t seq, uint32_t extra) {
  *dst = st;
  publish_tag(tag, seq, xor_words(st) ^ extra);
}

__global__ void __launch_bounds__(256) k_micp_moments(const MicpFastParams p) {
  __shared__ double red[4][kMom];
  __shared__ double s_scratch[4][2][64 * 17];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const float gate_lo = RMCL_FCALL(p, gate_lo), gate_hi = RMCL_FCALL(p, gate_hi), rho_cap = RMCL_FCALL(p, rho_cap), tau_cap = RMCL_FCALL(p, tau_cap);
  double m[kMom];
#pragma unroll
  for (int k = 0; k < kMom; ++k) m[k] = 0.0;
  // a wave takes 64 consecutive correspondences per step (one mask word); two steps are requested before the first is used
  const uint32_t stride = gridDim.x * 256u;
  const uint32_t nceil = (p.n + 63u) & ~63u;
  for (uint32_t base = (blockIdx.x * 4u + wave) * 64u; base < nceil; base += 2u * stride) {
    float d[2][3], q[2][3], nn[2][3];
    bool ok[2], live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t i = base + static_cast<uint32_t>(u) * stride + lane;
      live[u] = (base + static_cast<uint32_t>(u) * stride) < nceil;
      ok[u] = false;
#pragma unroll
      for (int k = 0; k < 3; ++k) { d[u][k] = 0.f; q[u][k] = 0.f; nn[u][k] = 0.f; }
      if (i < p.n) {
        const bool dok = (p.dataset_mask == nullptr) || (p.dataset_mask[i] > 0);
        ok[u] = dok && p.model_mask[i] > 0;
        const float* dp = p.dataset_points + 3 * static_cast<size_t>(i);
        const float* mp = p.model_points + 3 * static_cast<size_t>(i);
        const float* mn = p.model_normals + 3 * static_cast<size_t>(i);
#pragma unroll
        for (int k = 0; k < 3; ++k) { d[u][k] = dp[k]; q[u][k] = mp[k]; nn[u][k] = mn[k]; }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!live[u]) continue;   // wave-uniform
      const f3 Di = mk3(d[u][0], d[u][1], d[u][2]), Ii = mk3(q[u][0], q[u][1], q[u][2]), Ni = mk3(nn[u][0], nn[u][1], nn[u][2]);
      // the reduction's own gate value at the identity pre-transform
      const float spd0 = dot_plain(sub3(Ii, Di), Ni);
      const float nd = sqrtf(dot_plain(Di, Di));
      // (a NaN gate value -- a NaN / inf dataset point without a mask -- is NaN under every pre-transform: gated out for good)
      const int cls = micp_gate_class(spd0, nd, gate_lo, gate_hi, rho_cap, tau_cap);
      const bool uncertain = ok[u] && cls == 2;
      const unsigned long long word = __ballot(uncertain);
      if (lane == 0u) p.unc_mask[(base + static_cast<uint32_t>(u) * stride) >> 6] = word;
      if (ok[u] && cls == 1) {
        const double D[3] = {Di.x, Di.y, Di.z}, N[3] = {Ni.x, Ni.y, Ni.z};
        const double sI = (N[0] * static_cast<double>(Ii.x) + N[1] * static_cast<double>(Ii.y)) + N[2] * static_cast<double>(Ii.z);
        const double DD[6] = {D[0] * D[0], D[0] * D[1], D[0] * D[2], D[1] * D[1], D[1] * D[2], D[2] * D[2]};
        const double NN[6] = {N[0] * N[0], N[0] * N[1], N[0] * N[2], N[1] * N[1], N[1] * N[2], N[2] * N[2]};
        m[0] += 1.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) m[1 + j] += D[j];
#pragma unroll
        for (int k = 0; k < 6; ++k) m[4 + k] += DD[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double sn = sI * N[a];
          m[10 + a] += sn;
#pragma unroll
          for (int j = 0; j < 3; ++j) m[13 + 3 * a + j] += sn * D[j];
        }
#pragma unroll
        for (int pq = 0; pq < 6; ++pq) {
          m[22 + pq] += NN[pq];
#pragma unroll
          for (int j = 0; j < 3; ++j) m[28 + 3 * pq + j] += NN[pq] * D[j];
#pragma unroll
          for (int k = 0; k < 6; ++k) m[46 + 6 * pq + k] += NN[pq] * DD[k];
        }
      }
    }
  }
  // six chunks of 16 moments, each with its own scratch area so that the chunks overlap
#pragma unroll
  for (int c = 0; c < kMom / 16; ++c) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = m[16 * c + k];
    const double t = wave_sum16_lds(v, &s_scratch[wave][c & 1][0], lane);
    if (lane < 16u) red[wave][16 * c + lane] = t;
  }
  __syncthreads();
  if (threadIdx.x < kMom)
    p.partials[static_cast<size_t>(blockIdx.x) * kMom + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// The 16 raw sums of the reduction over the CERTAIN correspondences at pre-transform (R row-major, t) from the moments, by
// one wave.  With X ranging over the ten per-correspondence factors  N_aN_b (6, symmetric pairs) | 1 | s N_a (3)  the moments
// are W(X) = sum X, P(X)_j = sum X D_j, Q(X)_jk = sum X D_j D_k, and every sum below is one of
//   rP(X)_b = sum_j R_bj P(X)_j,   H(X)_bc = sum_jk R_bj R_ck Q(X)_jk   (X = "1": sum D'_b, sum D'_b D'_c up to the t terms).
// Stage A (63 lanes) G(X)_cj = sum_k Q(X)_jk R_ck; stage B (63 + 30 lanes) H and rP; stage C (16 lanes) the outputs.
// All operands live in LDS; LDS operations of one wave complete in order (no barrier between the stages).
__device__ __forceinline__ int sym3(int a, int b) {
  const int lo = min(a, b), hi = max(a, b);
  return lo == 0 ? hi : (lo == 1 ? hi + 2 : 5);
}
__device__ __forceinline__ const double* mom_Q(const double* mom, int x) { return x < 6 ? mom + 46 + 6 * x : mom + 4; }
__device__ __forceinline__ const double* mom_P(const double* mom, int x) { return x < 6 ? mom + 28 + 3 * x : (x == 6 ? mom + 1 : mom + 13 + 3 * (x - 7)); }
__device__ __forceinline__ double mom_W(const double* mom, int x) { return x < 6 ? mom[22 + x] : (x == 6 ? mom[0] : mom[10 + (x - 7)]); }

struct MomentScratch { double G[64], H[64], rP[32]; };

__device__ __forceinline__ void micp_moment_sums_wave(uint32_t lane, const double* mom, const double* R, const double* t,
                                                      MomentScratch* ws, double* tot) {
  // stage A: lane = 9 x + 3 c + j, x = 0..6
  if (lane < 63u) {
    const int x = static_cast<int>(lane) / 9, c = (static_cast<int>(lane) % 9) / 3, j = static_cast<int>(lane) % 3;
    const double* Q = mom_Q(mom, x);
    ws->G[lane] = (Q[sym3(j, 0)] * R[3 * c] + Q[sym3(j, 1)] * R[3 * c + 1]) + Q[sym3(j, 2)] * R[3 * c + 2];
  }
  // stage B: lane = 9 x + 3 b + c -> H(x)_bc; lanes 0..29 also rP(x)_b with lane = 3 x + b, x = 0..9
  if (lane < 63u) {
    const int x = static_cast<int>(lane) / 9, b = (static_cast<int>(lane) % 9) / 3, c = static_cast<int>(lane) % 3;
    const double* Gx = ws->G + 9 * x + 3 * c;
    ws->H[lane] = (R[3 * b] * Gx[0] + R[3 * b + 1] * Gx[1]) + R[3 * b + 2] * Gx[2];
  }
  if (lane < 30u) {
    const int x = static_cast<int>(lane) / 3, b = static_cast<int>(lane) % 3;
    const double* P = mom_P(mom, x);
    ws->rP[lane] = (R[3 * b] * P[0] + R[3 * b + 1] * P[1]) + R[3 * b + 2] * P[2];
  }
  // stage C
  if (lane < 16u) {
    const double n = mom[0];
    double out;
    if (lane == 15u) {
      out = n;
    } else if (lane < 3u) {
      const int c = static_cast<int>(lane);
      out = ws->rP[18 + c] + n * t[c];
    } else if (lane < 6u) {
      const int a = static_cast<int>(lane) - 3;
      double ndp = 0.0;   // sum N_a (N . D')
      for (int b = 0; b < 3; ++b) {
        const int x = sym3(a, b);
        ndp += ws->rP[3 * x + b] + t[b] * mom_W(mom, x);
      }
      out = (ws->rP[18 + a] + n * t[a]) + mom[10 + a] - ndp;
    } else {
      const int a = (static_cast<int>(lane) - 6) / 3, c = (static_cast<int>(lane) - 6) % 3;
      const double ddp = ws->H[54 + 3 * a + c] + ws->rP[18 + a] * t[c] + t[a] * ws->rP[18 + c] + n * t[a] * t[c];   // sum D'_a D'_c
      const double sndp = ws->rP[3 * (7 + a) + c] + mom[10 + a] * t[c];                                            // sum s N_a D'_c
      double nndd = 0.0;                                                                                           // sum N_a (N . D') D'_c
      for (int b = 0; b < 3; ++b) {
        const int x = sym3(a, b);
        nndd += ws->H[9 * x + 3 * b + c] + t[c] * ws->rP[3 * x + b] + t[b] * ws->rP[3 * x + c] + t[b] * t[c] * mom_W(mom, x);
      }
      out = ddp + sndp - nndd;
    }
    tot[lane] = out;
  }
}

// Sum of the per-block moment partials [nblocks][kMom] into s_part[kFoldGroups][kMom] (the caller adds the groups after a barrier):
// five groups of 48 threads, TWO moments (one 16-B load) per thread and row, 13 rows in flight -- 128 rows are two round trips
// (round 2: two groups of 96 threads, one moment each, 16 in flight: four round trips, ~2 us of the loop kernel's set-up).
constexpr uint32_t kMomUsed = 82;   // columns of a partial row that carry a moment
// Six groups of 41 lanes: a lane owns one 16-B column pair and every sixth row, 22 rows requested per round -- a fold is bound by what
// ONE compute unit can pull from L2 (64 B per clock), so only the 82 used columns are read and enough requests are in flight to
// keep that path busy: 128 rows (k_micp_moments) are one round, the 512 rows of a find with the moment epilogue four.
constexpr uint32_t kFoldGroups = 6, kFoldLanes = kMomUsed / 2, kFoldBatch = 22;
static_assert(kFoldGroups * kFoldLanes <= kFastThreads, "fold lanes");
__device__ __forceinline__ void fold_moment_partials(const double* __restrict__ partials, uint32_t nblocks, double (*s_part)[kMom], uint32_t tid) {
  const uint32_t k2 = tid % kFoldLanes, g = tid / kFoldLanes;
  if (g >= kFoldGroups) return;
  const double2* base = reinterpret_cast<const double2*>(partials) + k2;
  double a0 = 0.0, a1 = 0.0;
  for (uint32_t b0 = g; b0 < nblocks; b0 += kFoldBatch * kFoldGroups) {
    double2 v[kFoldBatch];
#pragma unroll
    for (uint32_t u = 0; u < kFoldBatch; ++u) {
      const uint32_t b = b0 + u * kFoldGroups;
      v[u] = (b < nblocks) ? base[static_cast<size_t>(b) * (kMom / 2)] : double2{0.0, 0.0};
    }
#pragma unroll
    for (uint32_t u = 0; u < kFoldBatch; ++u) { a0 += v[u].x; a1 += v[u].y; }
  }
  s_part[g][2u * k2] = a0;
  s_part[g][2u * k2 + 1u] = a1;
  if (k2 < (kMom - kMomUsed) / 2u) { s_part[g][kMomUsed + 2u * k2] = 0.0; s_part[g][kMomUsed + 2u * k2 + 1u] = 0.0; }
}

__global__ void __launch_bounds__(kFastThreads) k_micp_fast_loop(const MicpFastParams p) {
  constexpr uint32_t kGroups = kFoldGroups;
  __shared__ double s_mom[kMom];
  __shared__ double s_part[kGroups][kMom];
  __shared__ double s_rows[kFastThreads][17];   // per-thread raw sums of the uncertain correspondences (+1: bank spread)
  __shared__ double s_tot[16];
  __shared__ double s_R[9], s_t[3];
  __shared__ MomentScratch s_ws;
  __shared__ uint32_t s_list[kFastMaxUncertain];
  __shared__ uint32_t s_wave_cnt[kFastThreads / 64];
  __shared__ xform s_Tpre;
  __shared__ uint32_t s_flag;
  __shared__ uint32_t s_abort;   // a sibling workgroup's sums did not arrive within kDevicePollBound polls
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned long long clk0 = __builtin_readcyclecounter();

  // (1) moments = sum of the per-block partials.  One compute unit pulls 64 B per clock from L2: the 512 rows a find with the moment
  // epilogue leaves (336 KB) would take it 4 us, so that launch comes as several workgroups -- each folds its share of the rows, the
  // others hand their 82 sums to workgroup 0 (write-through stores, then a release of this call's sequence number) and leave
  const uint32_t nfold = gridDim.x;
  const uint32_t rows_per = (p.nblocks + nfold - 1u) / nfold;
  const uint32_t row0 = min(blockIdx.x * rows_per, p.nblocks), row1 = min(row0 + rows_per, p.nblocks);
  fold_moment_partials(p.partials + static_cast<size_t>(row0) * kMom, row1 - row0, s_part, tid);
  if (blockIdx.x != 0u) {
    __syncthreads();
    if (tid < kMomUsed) {
      double a = s_part[0][tid];
#pragma unroll
      for (uint32_t g = 1; g < kGroups; ++g) a += s_part[g][tid];
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.fold_rows) + blockIdx.x * kMom + tid,
                         static_cast<unsigned long long>(__double_as_longlong(a)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0u) __hip_atomic_store(p.fold_flags + blockIdx.x, RMCL_FCALL(p, seq), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // (2) the uncertain correspondences, in index order: count per thread over a contiguous range of mask words, block scan
  const uint32_t nwords = (p.mask_tiled != 0u) ? p.mask_nwords : ((p.n + 63u) >> 6);
  const uint32_t wpt = (nwords + kFastThreads - 1u) / kFastThreads;
  const uint32_t w0 = min(tid * wpt, nwords), w1 = min(w0 + wpt, nwords);
  uint32_t cnt = 0;
  for (uint32_t w = w0; w < w1; w += 8u) {   // eight words requested together (a word per iteration was one round trip per word)
    unsigned long long m[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) m[u] = (w + u < w1) ? p.unc_mask[w + u] : 0ull;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) cnt += static_cast<uint32_t>(__popcll(m[u]));
  }
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += v;
  }
  if (lane == 63u) s_wave_cnt[wave] = incl;
  if (tid == 0u) { s_flag = 0u; s_abort = 0u; }
  __syncthreads();
  if (tid < kMom) {
    double a = s_part[0][tid];
#pragma unroll
    for (uint32_t g = 1; g < kGroups; ++g) a += s_part[g][tid];
    if (nfold > 1u && tid < kMomUsed) {
      // the other workgroups' sums, in workgroup order; every reading thread acquires the flag itself
      // (all flags polled together, ONE acquire, all rows requested together: two memory round trips, not two per workgroup)
      const uint32_t seq = RMCL_FCALL(p, seq);
      bool ready;
      uint32_t polls = 0;
      do {
        uint32_t f[kMicpFoldBlocks];
#pragma unroll
        for (uint32_t b = 1; b < kMicpFoldBlocks; ++b)
          f[b] = (b < nfold) ? __hip_atomic_load(p.fold_flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
        ready = true;
#pragma unroll
        for (uint32_t b = 1; b < kMicpFoldBlocks; ++b) ready = ready && (f[b] == seq);
      } while (!ready && ++polls < kDevicePollBound);
      if (!ready) s_abort = 1u;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      unsigned long long v[kMicpFoldBlocks];
#pragma unroll
      for (uint32_t b = 1; b < kMicpFoldBlocks; ++b)
        v[b] = (b < nfold) ? __hip_atomic_load(reinterpret_cast<unsigned long long*>(p.fold_rows) + b * kMom + tid, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)
                           : 0ull;
#pragma unroll
      for (uint32_t b = 1; b < kMicpFoldBlocks; ++b) a += __longlong_as_double(static_cast<long long>(v[b]));
    }
    s_mom[tid] = a;
  }
  uint32_t wave_base = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kFastThreads / 64; ++w) {
    const uint32_t c = s_wave_cnt[w];
    if (w < wave) wave_base += c;
    total += c;
  }
  if (nfold > 1u) {   // (launch-uniform) did every sibling's row arrive?
    __syncthreads();
    if (s_abort != 0u) total = kFastMaxUncertain + 1u;
  }
  if (total > kFastMaxUncertain) {
    if (tid == 0u) {
      MicpFastStatus st;
      st.code = 2u; st.iter = 0u; st.n_uncertain = total; st.max_rho = 0.f; st.max_tau = 0.f; st.pad[0] = st.pad[1] = st.pad[2] = 0u;
      publish_status(p.status, st, p.done, RMCL_FCALL(p, seq), 0u);
    }
    return;
  }
  if (cnt != 0u) {
    uint32_t pos = wave_base + incl - cnt;
    for (uint32_t w = w0; w < w1; ++w) {
      unsigned long long bits = p.unc_mask[w];
      while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1ull;
        s_list[pos++] = micp_mask_index(p, w, static_cast<uint32_t>(b));
      }
    }
  }
  const float max_dist = RMCL_FCALL(p, max_dist), rho_cap = RMCL_FCALL(p, rho_cap), tau_cap = RMCL_FCALL(p, tau_cap);
  const uint32_t nrows = min(total, kFastThreads);
  // No undecided correspondence (the usual case of a tracking-size correction): everything the iterations need is the 82
  // moments, and wave 0 alone runs them -- no workgroup barrier inside the loop (the LDS operations of ONE wave complete in
  // program order); the other waves leave once the moments they summed are in LDS.
  const bool lone = (total == 0u);
  if (lone) {
    __syncthreads();
    if (wave != 0u) return;
  }
  // thread 0 owns the loop state (registers): the sensor-frame pre-transform and the statistics of the last iteration
  xform T_s = xidentity();
  cstats last = cs_identity();
  float max_rho = 0.f, max_tau = 0.f;
  const unsigned long long clk1 = __builtin_readcyclecounter();
  for (uint32_t it = 0; it < p.n_iter; ++it) {
    if (tid == 0u) {
      const float rho = 2.0f * sqrtf((T_s.R.x * T_s.R.x + T_s.R.y * T_s.R.y) + T_s.R.z * T_s.R.z);
      const float tau = sqrtf(dot_plain(T_s.t, T_s.t));
      max_rho = fmaxf(max_rho, rho);
      max_tau = fmaxf(max_tau, tau);
      if (!(rho <= rho_cap) || !(tau <= tau_cap)) s_flag = 1u;
      s_Tpre = T_s;
      // the linear map of qrot (q v q*), in double from the f32 components
      const double x = T_s.R.x, y = T_s.R.y, z = T_s.R.z, w = T_s.R.w;
      const double ww = w * w, uu = (x * x + y * y) + z * z;
      s_R[0] = (ww - uu) + 2.0 * x * x; s_R[1] = 2.0 * (x * y - w * z);   s_R[2] = 2.0 * (x * z + w * y);
      s_R[3] = 2.0 * (x * y + w * z);   s_R[4] = (ww - uu) + 2.0 * y * y; s_R[5] = 2.0 * (y * z - w * x);
      s_R[6] = 2.0 * (x * z - w * y);   s_R[7] = 2.0 * (y * z + w * x);   s_R[8] = (ww - uu) + 2.0 * z * z;
      s_t[0] = T_s.t.x; s_t[1] = T_s.t.y; s_t[2] = T_s.t.z;
    }
    if (lone) __builtin_amdgcn_wave_barrier();
    else __syncthreads();   // (A) pre-transform published; also orders the previous iteration's reads of s_rows / s_tot
    if (s_flag != 0u) {
      if (tid == 0u) {
        MicpFastStatus st;
        st.code = 1u; st.iter = it; st.n_uncertain = total; st.max_rho = max_rho; st.max_tau = max_tau; st.pad[0] = st.pad[1] = st.pad[2] = 0u;
        publish_status(p.status, st, p.done, RMCL_FCALL(p, seq), 0u);
      }
      return;
    }
    if (tid < nrows) {
      // the uncertain correspondences with the reduction's own arithmetic (k_micp_iter); re-read every iteration (L2 hits,
      // normally a few hundred elements) so that nothing of them is live across the one-lane solve.  Thread t sums elements
      // t, t + 256, ... into row t; the rows are added in index order below: no cross-lane butterfly, deterministic.
      const xform Tpre = s_Tpre;
      double acc[kAcc];
#pragma unroll
      for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
      for (uint32_t e = tid; e < total; e += kFastThreads) {
        const uint32_t i = s_list[e];
        const float* dp = p.dataset_points + 3 * static_cast<size_t>(i);
        const float* mp = p.model_points + 3 * static_cast<size_t>(i);
        const float* mn = p.model_normals + 3 * static_cast<size_t>(i);
        const f3 Di = xapply(Tpre, mk3(dp[0], dp[1], dp[2]));
        const f3 Ii = mk3(mp[0], mp[1], mp[2]);
        const f3 Ni = mk3(mn[0], mn[1], mn[2]);
        const float spd = dot_plain(sub3(Ii, Di), Ni);
        if (fabsf(spd) < max_dist) {
          const f3 Mi = add3(Di, scale3(Ni, spd));
          const double d[3] = {Di.x, Di.y, Di.z}, m[3] = {Mi.x, Mi.y, Mi.z};
#pragma unroll
          for (int k = 0; k < 3; ++k) { acc[k] += d[k]; acc[3 + k] += m[k]; }
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[6 + 3 * r + c] += m[r] * d[c];
          acc[15] += 1.0;
        }
      }
#pragma unroll
      for (int k = 0; k < kAcc; ++k) s_rows[tid][k] = acc[k];
    }
    if (nrows > 64u) __syncthreads();   // (B) rows of other waves (block-uniform condition); wave 0's own rows are in order
    if (wave == 0u) {
      micp_moment_sums_wave(lane, s_mom, s_R, s_t, &s_ws, s_tot);
      if (lane < 16u && nrows != 0u) {
        double v = s_tot[lane];
        for (uint32_t r = 0; r < nrows; ++r) v += s_rows[r][lane];
        s_tot[lane] = v;
      }
    }
    if (tid == 0u) {
      // lanes 0..15 of this wave wrote s_tot just above (LDS operations of one wave complete in order)
      double tot[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) tot[k] = s_tot[k];
      last = cstats_from_sums(tot);
      T_s = xmul(T_s, umeyama_fast(last));   // micp_advance_sensor with the cheaper reciprocals
    }
  }
  if (tid == 0u) {
    MicpState out;
    micp_close_sensor(last, T_s, RMCL_FCALL(p, Tsb), RMCL_FCALL(p, Tbo), &out);
    *p.state_out = out;
    MicpFastStatus st;
    st.code = 0u; st.iter = p.n_iter; st.n_uncertain = total; st.max_rho = max_rho; st.max_tau = max_tau;
    st.pad[0] = static_cast<uint32_t>(clk1 - clk0);                              // diagnostics: shader clocks of the set-up
    st.pad[1] = static_cast<uint32_t>(__builtin_readcyclecounter() - clk1);      // ... and of all iterations
    st.pad[2] = 0u;
    publish_status(p.status, st, p.done, RMCL_FCALL(p, seq), xor_words(out));
  }
}

// ---------------------------------------------------------------------------------------------
// Round 4: the iterations leave the device.  k_micp_fast_loop above spends ~4.4 k cycles per iteration in ONE lane's dependent f64
// chain (Horn's quartic, the frame products) -- 34 of a correction's 60 us -- although an iteration is a closed-form function of
// the 82 moments and the few undecided correspondences.  k_micp_publish folds the per-workgroup rows exactly as the loop kernel
// does (same order, same sums) and writes {moments, undecided count, D | I | N of every undecided correspondence} into pinned
// host memory behind one completion tag; the host (micp_host.h) then runs the iterations -- rmclhip_rcc_correct_once -- or
// answers every computeCrossStatistics of the reference's unchanged caller loop (micp_localization.cpp:915-964) with no launch
// at all.  More than kMicpHostMaxUnc undecided correspondences: code 2, the caller launches the device loop on the same rows.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kFastThreads) k_micp_publish(const MicpFastParams p) {
  constexpr uint32_t kGroups = kFoldGroups;
  __shared__ double s_part[kGroups][kMom];
  __shared__ uint32_t s_list[kMicpHostMaxUnc];
  __shared__ float s_stage[9u * kMicpHostMaxUnc];
  __shared__ uint32_t s_wave_cnt[kFastThreads / 64];
  __shared__ uint32_t s_xor[kFastThreads / 64];
  __shared__ uint32_t s_abort;   // a sibling workgroup's sums did not arrive within kDevicePollBound polls: code 2
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t nfold = gridDim.x;
  const uint32_t rows_per = (p.nblocks + nfold - 1u) / nfold;
  const uint32_t row0 = min(blockIdx.x * rows_per, p.nblocks), row1 = min(row0 + rows_per, p.nblocks);
  fold_moment_partials(p.partials + static_cast<size_t>(row0) * kMom, row1 - row0, s_part, tid);
  if (blockIdx.x != 0u) {
    __syncthreads();
    if (tid < kMomUsed) {
      double a = s_part[0][tid];
#pragma unroll
      for (uint32_t g = 1; g < kGroups; ++g) a += s_part[g][tid];
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.fold_rows) + blockIdx.x * kMom + tid,
                         static_cast<unsigned long long>(__double_as_longlong(a)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0u) __hip_atomic_store(p.fold_flags + blockIdx.x, p.cv.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // the undecided correspondences, in index order (as k_micp_fast_loop counts them)
  const uint32_t nwords = (p.mask_tiled != 0u) ? p.mask_nwords : ((p.n + 63u) >> 6);
  const uint32_t wpt = (nwords + kFastThreads - 1u) / kFastThreads;
  const uint32_t w0 = min(tid * wpt, nwords), w1 = min(w0 + wpt, nwords);
  uint32_t cnt = 0;
  for (uint32_t w = w0; w < w1; w += 8u) {
    unsigned long long m[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) m[u] = (w + u < w1) ? p.unc_mask[w + u] : 0ull;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) cnt += static_cast<uint32_t>(__popcll(m[u]));
  }
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += v;
  }
  if (lane == 63u) s_wave_cnt[wave] = incl;
  if (tid == 0u) s_abort = 0u;
  __syncthreads();
  uint32_t x = 0u;   // xor of the words this thread writes for the host
  if (tid < kMom) {
    double a = 0.0;
    if (tid < kMomUsed) {
      a = s_part[0][tid];
#pragma unroll
      for (uint32_t g = 1; g < kGroups; ++g) a += s_part[g][tid];
      if (nfold > 1u) {
        const uint32_t seq = p.cv.seq;
        bool ready;
        uint32_t polls = 0;
        do {
          uint32_t f[kMicpFoldBlocks];
#pragma unroll
          for (uint32_t b = 1; b < kMicpFoldBlocks; ++b)
            f[b] = (b < nfold) ? __hip_atomic_load(p.fold_flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
          ready = true;
#pragma unroll
          for (uint32_t b = 1; b < kMicpFoldBlocks; ++b) ready = ready && (f[b] == seq);
        } while (!ready && ++polls < kDevicePollBound);
        if (!ready) s_abort = 1u;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned long long v[kMicpFoldBlocks];
#pragma unroll
        for (uint32_t b = 1; b < kMicpFoldBlocks; ++b)
          v[b] = (b < nfold) ? __hip_atomic_load(reinterpret_cast<unsigned long long*>(p.fold_rows) + b * kMom + tid, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)
                             : 0ull;
#pragma unroll
        for (uint32_t b = 1; b < kMicpFoldBlocks; ++b) a += __longlong_as_double(static_cast<long long>(v[b]));
      }
    }
    p.host_block->mom[tid] = a;
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(a));
    x ^= static_cast<uint32_t>(bits) ^ static_cast<uint32_t>(bits >> 32);
  }
  uint32_t wave_base = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kFastThreads / 64; ++w) {
    const uint32_t c = s_wave_cnt[w];
    if (w < wave) wave_base += c;
    total += c;
  }
  const bool fits = total <= kMicpHostMaxUnc;   // block-uniform
  if (fits && total != 0u) {
    if (cnt != 0u) {
      uint32_t pos = wave_base + incl - cnt;
      for (uint32_t w = w0; w < w1; ++w) {
        unsigned long long bits = p.unc_mask[w];
        while (bits) {
          const int b = __builtin_ctzll(bits);
          bits &= bits - 1ull;
          s_list[pos++] = micp_mask_index(p, w, static_cast<uint32_t>(b));
        }
      }
    }
    __syncthreads();
    // gather D | I | N of the listed correspondences into LDS, then write the block to the host in address order (consecutive lanes,
    // consecutive dwords: a thread storing its own 36-B record put nine narrow writes per correspondence on the bus)
    for (uint32_t e = tid; e < total; e += kFastThreads) {
      const uint32_t i = s_list[e];
      const float* dp = p.dataset_points + 3 * static_cast<size_t>(i);
      const float* mp = p.model_points + 3 * static_cast<size_t>(i);
      const float* mn = p.model_normals + 3 * static_cast<size_t>(i);
      const float v[9] = {dp[0], dp[1], dp[2], mp[0], mp[1], mp[2], mn[0], mn[1], mn[2]};
#pragma unroll
      for (int k = 0; k < 9; ++k) s_stage[9u * e + static_cast<uint32_t>(k)] = v[k];
    }
    __syncthreads();
    float* dst = &p.host_block->unc[0][0];
    for (uint32_t w = tid; w < 9u * total; w += kFastThreads) {
      const float v = s_stage[w];
      dst[w] = v;
      x ^= __float_as_uint(v);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x ^= __shfl_xor(x, off, 64);
  if (lane == 0u) s_xor[wave] = x;
  __threadfence_system();   // this thread's stores to the host block, before the tag below
  __syncthreads();
  if (tid == 0u) {
    const uint32_t code = (fits && s_abort == 0u) ? 0u : 2u;
    p.host_block->code = code;
    p.host_block->n_uncertain = total;
    p.host_block->pad[0] = 0u; p.host_block->pad[1] = 0u;
    publish_tag(p.done, p.cv.seq, ((s_xor[0] ^ s_xor[1]) ^ (s_xor[2] ^ s_xor[3])) ^ (code ^ total));
  }
}


// N sensors, one iteration of MICPLocalizationNode::correctOnce (micp_localization.cpp:915-964), ONE wave:
//   per sensor (in order): stats_s <- partials; Cs_b = Tsb * stats_s (MICPSensor.hpp:182); Cs_o = Tbo * Cs_b (:931);
//   Cs_weighted_o = Cs_o with n_meas *= merge_weight_multiplier (truncating, :934); Cmerged_o += Cs_o; Cmerged_weighted_o += ...
//   T_inner = umeyama(Cmerged_weighted_o) (:952); T_onew_oold *= T_inner (:963);
//   next pre-transforms: T_bnew_bold = ~Tbo * T_onew_oold * Tbo (:926), T_snew_sold = ~Tsb * T_bnew_bold * Tsb (MICPSensor.hpp:178)
__global__ void __launch_bounds__(64) k_micp_multi_step(const MicpMultiCall* __restrict__ call, MicpMultiState* __restrict__ st) {
  const uint32_t ns = call->n_sensors;
  cstats merged = cs_identity(), merged_w = cs_identity();
  for (uint32_t s = 0; s < ns; ++s) {
    const cstats stats_s = finalize_pose(call->partials[s], call->nblocks[s]);   // whole wave
    if (threadIdx.x == 0) {
      const cstats Cs_o = cs_transform(call->Tbo[s], cs_transform(call->Tsb[s], stats_s));
      cstats Cs_w = Cs_o;
      Cs_w.n_meas = static_cast<uint32_t>(static_cast<double>(Cs_w.n_meas) * call->weight[s]);
      merged = cs_merge(merged, Cs_o);
      merged_w = cs_merge(merged_w, Cs_w);
    }
  }
  if (threadIdx.x == 0) {
    const xform T_inner = umeyama(merged_w);
    const xform T_onew_oold = xmul(st->T_onew_oold, T_inner);
    st->T_onew_oold = T_onew_oold;
    st->merged_o = merged;
    st->merged_weighted_o = merged_w;
    for (uint32_t s = 0; s < ns; ++s) {
      const xform T_bnew_bold = xmul(xmul(xinv(call->Tbo[s]), T_onew_oold), call->Tbo[s]);
      st->T_snew_sold[s] = xmul(xmul(xinv(call->Tsb[s]), T_bnew_bold), call->Tsb[s]);
    }
  }
}

__global__ void k_micp_multi_init(const MicpMultiCall* __restrict__ call, MicpMultiState* __restrict__ st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->T_onew_oold = xidentity();
  st->merged_o = cs_identity();
  st->merged_weighted_o = cs_identity();
  for (uint32_t s = 0; s < kMaxMicpSensors; ++s) st->T_snew_sold[s] = xidentity();
}

// stale v1 corrector (lidar_corrector_embree_benchmark.cpp:127-135): per pose, Tdelta_b = Tsb * T_s * ~Tsb
__global__ void __launch_bounds__(64) k_batch_solve(const double* __restrict__ partials, uint32_t nblocks, xform Tsb,
                                                    xform* __restrict__ Tdelta, cstats* __restrict__ stats) {
  const uint32_t pose = blockIdx.x;
  const cstats s = finalize_pose(partials + static_cast<size_t>(pose) * nblocks * kAcc, nblocks);
  if (threadIdx.x == 0) {
    const xform Ts = umeyama(s);
    Tdelta[pose] = xmul(xmul(Tsb, Ts), xinv(Tsb));
    if (stats) stats[pose] = s;
  }
}

// MICPSphericalSensorCPU::unpackMessage / MICPO1DnSensorCPU::unpackMessage (dataset construction)
__global__ void k_dataset_from_ranges(const float* __restrict__ ranges, const float* __restrict__ tab, uint32_t kind,
                                      uint32_t W, uint32_t H, f3 orig, float fx, float fy, float cx, float cy,
                                      float rmin, float rmax, float* __restrict__ points,
                                      uint8_t* __restrict__ mask, uint32_t* __restrict__ n_valid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const uint32_t vid = i / W, hid = i - vid * W;
  const float r = ranges[i];
  f3 dir, o = orig;
  if (kind == kModelSpherical) {
    const float cp = tab[vid], sp = tab[H + vid], ct = tab[2u * H + hid], st = tab[2u * H + W + hid];
    dir = mk3(cp * ct, cp * st, sp);
  } else if (kind == kModelPinhole) {
    dir = pinhole_direction(fx, fy, cx, cy, vid, hid);
  } else if (kind == kModelOnDn) {
    o = mk3(tab[3u * i], tab[3u * i + 1u], tab[3u * i + 2u]);
    const float* dr = tab + 3u * (static_cast<size_t>(W) * H + i);
    dir = mk3(dr[0], dr[1], dr[2]);
  } else {
    dir = mk3(tab[3u * i], tab[3u * i + 1u], tab[3u * i + 2u]);
  }
  // unpackMessage: spherical adds no origin (MICPSphericalSensorCPU.cpp:218), the others do (MICPO1DnSensorCPU.cpp:211-213)
  f3 pt = scale3(dir, r);
  if (kind != kModelSpherical) pt = add3(pt, o);
  points[3u * i] = pt.x; points[3u * i + 1u] = pt.y; points[3u * i + 2u] = pt.z;
  const bool out_of_range = (r < rmin) || (r > rmax);
  mask[i] = out_of_range ? 0 : 1;
  if (!out_of_range) atomicAdd(n_valid, 1u);
}

// sensor_msgs/PointCloud2 bytes -> O1Dn model (dirs) + dataset (points, mask) in one pass:
// estimateModelAndData (conversions.cpp:869-1002) + filter (scan_operations.cpp:41-116) + MICPO1DnSensorCPU::unpackMessage
struct Pc2Params {
  const uint8_t* data;
  uint32_t point_step, row_step, off_x, off_y, off_z, is_f64;
  uint32_t h_skip, h_inc, w_skip, w_inc;
  uint32_t out_w, out_h;
  float range_min, range_max;
  float* dirs;
  float* points;
  uint8_t* mask;
  uint32_t* n_valid;
};

__device__ __forceinline__ float pc2_load(const uint8_t* p, bool f64) {
  if (f64) { double d; __builtin_memcpy(&d, p, 8); return static_cast<float>(d); }
  float f; __builtin_memcpy(&f, p, 4); return f;
}

__global__ void __launch_bounds__(256) k_pointcloud2_unpack(const Pc2Params p) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= p.out_w * p.out_h) return;
  const uint32_t ti = id / p.out_w, tj = id - ti * p.out_w;
  const size_t si = static_cast<size_t>(ti) * p.h_inc + p.h_skip, sj = static_cast<size_t>(tj) * p.w_inc + p.w_skip;
  const uint8_t* ptr = p.data + si * p.row_step + sj * p.point_step;
  const float x = pc2_load(ptr + p.off_x, p.is_f64), y = pc2_load(ptr + p.off_y, p.is_f64), z = pc2_load(ptr + p.off_z, p.is_f64);
  const bool fin = (fabsf(x) <= 3.402823466e38f) && (fabsf(y) <= 3.402823466e38f) && (fabsf(z) <= 3.402823466e38f);
  float range = 0.0f;
  f3 d = mk3(0.f, 0.f, 0.f);
  if (fin) {
    range = sqrtf((x * x + y * y) + z * z);
    d = mk3(x / range, y / range, z / range);
  }
  p.dirs[3u * id] = d.x; p.dirs[3u * id + 1u] = d.y; p.dirs[3u * id + 2u] = d.z;
  p.points[3u * id] = d.x * range + 0.0f; p.points[3u * id + 1u] = d.y * range + 0.0f; p.points[3u * id + 2u] = d.z * range + 0.0f;
  const bool out_of_range = (range < p.range_min) || (range > p.range_max);
  p.mask[id] = out_of_range ? 0 : 1;
  if (!out_of_range) atomicAdd(p.n_valid, 1u);
}

// ---------------------------------------------------------------------------------------------
// Round 3 form of the persistent-lane kernel (the default).  tools/pfsim.py -- a wave-level model of this schedule on the
// product's own BVH arrays that reproduces the round-2 PMC figures (2.0 ms, ~60 % of the lanes active) -- says the kernel is
// bound by VALU instruction ISSUES, and that what moves the issue count is (a) shorter leaves, (b) what the refill block and
// the merge tail cost, not how many lanes sit in each issue:
//   * the filter's OWN tree (bvh_build.h): leaves of <= 2 records, same record array; a leaf visit is ONE round trip for
//     both records (leaf_pair) instead of a loop over up to four;
//   * the refill block only derives the point-to-plane ERROR of a finished beam; the double-precision exp / divide of the
//     likelihood runs afterwards as a dense pass over the block's beams (every lane busy) -- same operations, same result;
//   * the in-order Gaussian1D merge (one lane per particle, sequential by the reference's semantics) no longer divides:
//     the count sequence of a particle is known up front (n0 + k, clamped), so the two weights of every (particle, beam) are
//     computed by all lanes into the LDS the traversal stacks have just vacated, and the chain is ~14 dependent flops;
//   * ray set-up takes v_rcp_f32 for the slab reciprocals (the slab test is free-form and conservative: boxes are padded
//     1000 x wider than the reciprocal's error).
// Results are those of k_pf_update / k_pf_update_persist bit for bit (tests/test_gpu_pf.py runs all of them).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_inv(float d) {
  const float ad = fabsf(d);
  const float s = (ad < 1e-30f) ? copysignf(1e-30f, d) : d;
  return __builtin_amdgcn_rcpf(s);
}

// a leaf of the filter's tree (<= kPfLeafTris = 2 records) in one memory round trip; a lane whose leaf holds one record
// re-tests it (a repeated test cannot change (best_t, best_rec))
__device__ __forceinline__ void leaf_pair(const uint32_t* __restrict__ tris, uint32_t cur, f3 O, f3 D, float ray_tfar,
                                          float& best_t, uint32_t& best_rec) {
  const uint32_t first = cur & 0x0FFFFFFFu;
  const bool two = ((cur >> 28) & 7u) != 0u;
  const uint32_t second = first + (two ? 1u : 0u);
  const uint4* t0 = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(first) * 4u;
  const uint4* t1 = reinterpret_cast<const uint4*>(tris) + static_cast<size_t>(second) * 4u;
  const uint4 a0 = t0[0], b0 = t0[1], c0 = t0[2];
  const uint4 a1 = t1[0], b1 = t1[1], c1 = t1[2];
  tri_update(a0, b0, c0, first, tris, O, D, ray_tfar, best_t, best_rec);
  if (__any(two)) tri_update(a1, b1, c1, second, tris, O, D, ray_tfar, best_t, best_rec);
}

// ---------------------------------------------------------------------------------------------
// Round 5: ORDER-INDEPENDENT likelihood accumulation (kAccum).  The reference merges a particle's beams one by one,
//   likelihood += Gaussian1D{eval, 0, 1};  n_meas = min(n_meas, MAX_N_MEAS)             (PCDSensorUpdaterEmbree.cpp:232-238)
// which is a linear recurrence with weights that depend on the COUNTS alone: with S = sigma + mean^2,
//   mean' = w1 mean + w2 e,   S' = w1 S + w2 e^2,   w1 = a / (a + 1),  w2 = 1 / (a + 1),  a = the count before the beam.
// So   mean_K = P0 mean_0 + sum_k W_k e_k,   S_K = P0 S_0 + sum_k W_k e_k^2,   W_k = w2_k prod_{j > k} w1_j,  P0 = prod_j w1_j,
// and the counts are a_k = min(n0 + k, MAX) (a_0 = n0 whatever it is): W_k is the SAME for every beam merged before the clamp
// (g^(K - Kc) / (n0 + Kc), Kc = min(MAX - n0, K), g = MAX / (MAX + 1)) and g^(K - 1 - k) / (MAX + 1) for the beams after it.
// Rounds 3 / 4 kept every beam's error (LDS, then 100 MB of global scratch written and read back) for a dense pass and ONE lane per
// particle walking the chain.  Here a finished ray adds W_k e and W_k e^2 straight into its particle's accumulators and is forgotten.
// The sums must not depend on the order the rays finish in (results are compared bit for bit across schedules, shards and runs):
// the accumulators are FIXED-POINT -- per quantity a row of 64-bit integers, one per 16 binades ("bin"), a term t = m 2^e going to
// bin (e - e_min) / 16 as m shifted so that the bin's top is 2^48 units: integer adds commute, a term keeps >= 32 significant bits
// (the eval is a float: 24), 2^16 terms fit.  Final value = the bins added in ascending order in double.
// Differences to the sequential float chain: ~1e-7 relative (its own rounding).  sigma = max(S - mean^2, 0): every term keeps >= 32
// significant bits (rounded to nearest into its bin, so the error has no sign), hence S and mean^2 each carry ~2^-33 relative and a
// variance below ~1e-10 mean^2 is noise of either sign -- clamped at 0, which is what the reference's chain of non-negative terms
// (P1 + P2 >= 0 in Gaussian1D::operator+=) can reach at the least (the unclamped difference went negative for particles
// whose beams all evaluate alike, and a downstream sqrt(sigma) would have been NaN).  n_meas is the closed form as before, bit-exact.
// ---------------------------------------------------------------------------------------------
// exp(arg) * inv_den for arg <= 0 in float arithmetic: arg log2(e) split Cody-Waite style into an integer n and a residual r in
// [-0.5, 0.5] (one fused multiply-add against each half of log2 e: |error of r| ~ 1e-8), 2^r by v_exp_f32 (1 ulp), the scale before
// the exponent so that results below 2^-126 round like any other denormal.  Below 2^-149 the result is 0.  A NaN argument gives 0, not
// NaN (fmaxf is maxnum): it can only come from NaN penalty parameters, which rmclhip_pf_set_params rejects.
__device__ __forceinline__ float pf_eval_fast(float arg, float inv_den) {
  const float kL2eHi = 1.44269502162933349609375f, kL2eLo = 1.925963033500011e-08f;
  const float n = fmaxf(rintf(arg * kL2eHi), -400.0f);
  float r = fmaf(arg, kL2eHi, -n);
  r = fmaf(arg, kL2eLo, r);
  return ldexpf(__builtin_amdgcn_exp2f(fmaxf(r, -2.0f)) * inv_den, static_cast<int>(n));
}

constexpr int kAccBins1 = 18, kAccEmin1 = -256;   // sum W e:   terms in [2^-256, 2^32): evals up to 4e9, i.e. dist_sigma down to 1e-10
constexpr int kAccBins2 = 30, kAccEmin2 = -416;   // sum W e^2: terms in [2^-416, 2^64)
constexpr int kAccWords = kAccBins1 + kAccBins2;  // 64-bit words per particle

// returns false when t lies above the row's range (the caller poisons the particle: NaN, as an infinite term would make the chain)
__device__ __forceinline__ bool acc_add(unsigned long long* row, int nbins, int e_min, double t) {
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(t));
  const int e = static_cast<int>((b >> 52) & 0x7FFull) - 1023;   // floor(log2 t) of a positive normal double
  const int j = (e - e_min) >> 4;
  if (j < 0) return true;                                         // below 2^e_min: cannot reach a float result
  if (j >= nbins) return false;
  const int shift = 4 + (e_min + 16 * (j + 1) - e);               // 5..20: the bin's top 2^E is 2^48 units
  const unsigned long long mant = (b & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  atomicAdd(row + j, (mant + (1ull << (shift - 1))) >> shift);   // round to nearest: truncation biased every sum downwards
  return true;
}
__device__ __forceinline__ double acc_value(const unsigned long long* row, int nbins, int e_min) {
  double v = 0.0;
  for (int j = 0; j < nbins; ++j) v += ldexp(static_cast<double>(row[j]), e_min + 16 * (j + 1) - 48);
  return v;
}

template <int kRows, bool kLeaf2, bool kSlotOrder = false, bool kAccum = false>
// (kAccum: 72 instead of 78 registers = 7 instead of 6 waves per SIMD, which is also what the LDS admits: the C5 shard -5 %, room-100k -2 %,
//  sphere-100k unchanged -- profiles/r05_pf_forms_ab.txt; the stored forms keep the 6 they were measured with)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kAccum ? 7 : 6, kAccum ? 7 : 6))) k_pf_update_v3(const PfParams p) {
  // LDS: [ per-lane stacks kRows*256 (later: merge weights) | Tsm (PB xforms) | n0 (PB) | errors -> evals (PB*n_beams floats) ]
  // (round 4 A/B, removed: the top 85 / 341 nodes of the tree in LDS -- the TD / TA units read 94 % / 82 % busy, but those are
  // "non-idle" counters, not bandwidth: 14 % / 33 % SLOWER, profiles/r04_pf_lds_top.txt.  The kernel is VALU-issue bound at the
  // per-class issue costs of profiles/r04_valu_issue_rate.txt: DESIGN.md, particle filter.)
  extern __shared__ uint32_t lds_dyn[];
  __shared__ uint32_t s_next;
  uint32_t* lds_col = lds_dyn + threadIdx.x;   // stack rows of this lane: row r at lds_col[r * 256], row 0 = sentinel
  xform* s_Tsm = reinterpret_cast<xform*>(lds_dyn + kRows * 256);
  uint32_t* s_n0 = reinterpret_cast<uint32_t*>(s_Tsm + p.particles_per_block);

  const uint32_t PB = p.particles_per_block;
  const uint32_t p0 = blockIdx.x * PB;
  if (p0 >= p.n_particles) return;
  // the workgroup's beam errors: LDS behind n0 (rounds 3), or its slice of the global scratch (round 4: see PfParams::evals)
  const bool evals_global = p.evals != nullptr;
  float* s_eval = evals_global ? (p.evals + static_cast<size_t>(p0) * p.n_beams) : reinterpret_cast<float*>(s_n0 + p.particles_per_block);
  const uint32_t np = min(PB, p.n_particles - p0);
  pattrs* attrs = reinterpret_cast<pattrs*>(p.attrs);
  // kAccum: behind n0 (8-byte aligned) per particle { W of an unclamped beam (double) | first clamped beam | poison } and the bins
  // (the dynamic segment starts behind the 4 bytes of s_next: 8-byte alignment is taken from the ADDRESS, not from the offset -- 64-bit LDS
  // atomics fault on a misaligned word; the launcher reserves the dword this may skip)
  const uint32_t acc_dw = kRows * 256u + 8u * PB + PB;
  double* s_wu = reinterpret_cast<double*>(lds_dyn + acc_dw + (((static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_dyn)) >> 2) + acc_dw) & 1u));
  uint32_t* s_kc = reinterpret_cast<uint32_t*>(s_wu + PB);
  uint32_t* s_bad = s_kc + PB;
  unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(s_bad + PB);
  // slot j of the block = particle `mine` (identity, or the caller's spatial order)
  const uint32_t mine_p = (threadIdx.x < np) ? (p.order ? p.order[p0 + threadIdx.x] : p0 + threadIdx.x) : 0u;
  if (threadIdx.x < np) {
    s_Tsm[threadIdx.x] = xmul(p.poses[mine_p], p.Tsb);
    const uint32_t n0 = attrs[mine_p].likelihood.n_meas;
    s_n0[threadIdx.x] = n0;
    if (kAccum) {
      const uint32_t K = p.n_beams;
      // beams 0 .. kc-1 are merged before the count clamps (n0 >= MAX: beam 0 alone, with a = n0)
      const uint32_t kc = (n0 < p.max_n_meas) ? min(p.max_n_meas - n0, K) : 1u;
      s_wu[threadIdx.x] = p.gpow[K - kc] / (static_cast<double>(n0) + static_cast<double>(kc));
      s_kc[threadIdx.x] = kc;
      s_bad[threadIdx.x] = 0u;
    }
  }
  if (kAccum)
    for (uint32_t i = threadIdx.x; i < np * static_cast<uint32_t>(kAccWords); i += 256u) s_acc[i] = 0ull;
  if (threadIdx.x == 0) s_next = 0u;
  __syncthreads();

  const float sq = p.dist_sigma * p.dist_sigma;
  const float inv_den_f = static_cast<float>(1.0 / sqrt(static_cast<double>(2 * sq) * 3.14159265358979323846));
  const uint32_t nrays = np * p.n_beams;
  const uint32_t np_magic = (np == 1u) ? 0u : static_cast<uint32_t>(0x100000000ull / np) + 1u;   // ray / np (particle-minor order)
  const uint32_t lane = threadIdx.x & 63u;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  // per-lane ray state
  uint32_t rr = 0;
  bool has_ray = false, exhausted = false;
  f3 O = mk3(0.f, 0.f, 0.f), D = O;
  RaySlab rs = make_ray_slab(O, mk3(1.f, 1.f, 1.f));
  float range = 0.f, best_t = 0.f;
  uint32_t best_rec = kNone;
  uint32_t rev_mask = 0u;   // kSlotOrder: bit a = the ray runs towards negative `a`
  uint32_t priv[(kRows < 65) ? (65 - kRows) : 1];
  lds_col[0] = kDone;
  uint32_t sp = 1, cur = kDone;
#define RMCL_ROW_ST(r, v) { if ((r) < static_cast<uint32_t>(kRows)) lds_col[(r) * kBfStride] = (v); else priv[(r) - kRows] = (v); }
#define RMCL_ROW_LD(r) (((r) < static_cast<uint32_t>(kRows)) ? lds_col[(r) * kBfStride] : priv[(r) - kRows])
  for (;;) {
    const bool idle = (cur == kDone) && !exhausted;
    const uint64_t want = __ballot(idle);
    const uint64_t busy = __ballot(cur != kDone);
    if (want == 0 && busy == 0) break;
    if (want != 0 && (busy == 0 || static_cast<uint32_t>(__popcll(want)) >= p.refill_thr)) {
      if (idle && has_ray) {
        // evaluate_rcc (PCDSensorUpdaterEmbree.cpp:18-86) with unit face normals (BeamEvaluateProgram.cu:104-113): the error only
        const bool real_hit = (range >= p.range_min) && (range <= p.range_max);
        const bool sim_hit = (best_rec != kNone) && (!p.sim_min_range || best_t > p.range_min);
        float error;
        if (sim_hit) {
          if (real_hit) {
            const f3 n = pf_error_normal(p.tris, best_rec, p.raw_ng);
            const f3 preal = add3(O, scale3(D, range));
            const f3 pint = add3(O, scale3(D, best_t));
            error = fabsf(dot_plain(sub3(pint, preal), n));
          } else {
            error = p.rmsh;
          }
        } else {
          error = real_hit ? p.rhsm : p.rmsm;
        }
        if (kAccum) {
          const uint32_t pi = (p.n_beams == 1u) ? rr : __umulhi(rr, p.nb_magic), k = rr - pi * p.n_beams;
          if (p.errors) p.errors[static_cast<size_t>(p.order ? p.order[p0 + pi] : (p0 + pi)) * p.n_beams + k] = error;
          // PCDSensorUpdaterEmbree.cpp:224 evaluates exp(-e^2 / sigma^2 / 2) / sqrt(2 sigma^2 pi) with a float argument, double exp /
          // sqrt and a float result; here the same float argument through pf_eval_fast: <= 2e-7 relative from that value
          const float arg = -(error * error) / sq / 2;
          const float ev = pf_eval_fast(arg, inv_den_f);
          const double e1 = static_cast<double>(ev);
          const double w = (k < s_kc[pi]) ? s_wu[pi] : p.gpow[p.n_beams - 1u - k] * p.inv_max1;
          unsigned long long* row = s_acc + pi * static_cast<uint32_t>(kAccWords);
          bool ok = e1 < __builtin_inf();                // false for NaN and +inf
          if (ok && e1 > 0.0) {
            const double t1 = w * e1;
            ok = acc_add(row, kAccBins1, kAccEmin1, t1) && acc_add(row + kAccBins1, kAccBins2, kAccEmin2, t1 * e1);
          }
          if (!ok) atomicOr(&s_bad[pi], 1u);
        } else {
          s_eval[rr] = error;
        }
        has_ray = false;
      }
      // next rays for the idle lanes: one LDS atomic per wave and refill
      const uint32_t nwant = static_cast<uint32_t>(__popcll(want));
      const int leader = __builtin_ctzll(want);
      uint32_t base = 0;
      if (static_cast<int>(lane) == leader) base = atomicAdd(&s_next, nwant);
      base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(base), leader));
      if (idle) {
        const uint32_t mine = base + static_cast<uint32_t>(__popcll(want & ((1ull << lane) - 1ull)));
        if (mine < nrays) {
          // (n_beams == 1: floor(2^32 / 1) + 1 does not fit the 32-bit magic -- every ray is its own particle)
          uint32_t pi, b;
          if (p.particle_minor) { b = (np == 1u) ? mine : __umulhi(mine, np_magic); pi = mine - b * np; }
          else { pi = (p.n_beams == 1u) ? mine : __umulhi(mine, p.nb_magic); b = mine - pi * p.n_beams; }
          rr = pi * p.n_beams + b;     // the beam's slot in s_eval (particle-major whatever the dealing order)
          const xform Tsm = s_Tsm[pi];
          const float* bm = p.beams + 16u * b;
          // meas_m = Tsm * meas_s (RangeMeasurement.hpp:28-42)
          D = qrot(Tsm.R, mk3(bm[3], bm[4], bm[5]));
          O = p.beams_at_origin ? Tsm.t : xapply(Tsm, mk3(bm[0], bm[1], bm[2]));
          range = bm[6];
          rs.inv = mk3(fast_inv(D.x), fast_inv(D.y), fast_inv(D.z));
          rs.noi = mk3(-(O.x * rs.inv.x), -(O.y * rs.inv.y), -(O.z * rs.inv.z));
          if (kSlotOrder) rev_mask = (D.x < 0.0f ? 1u : 0u) | (D.y < 0.0f ? 2u : 0u) | (D.z < 0.0f ? 4u : 0u);
          best_t = p.ray_tfar;
          best_rec = kNone;
          sp = 1;
          has_ray = true;
          const bool finite = (D.x == D.x) && (D.y == D.y) && (D.z == D.z);
          cur = finite ? 0u : kDone;  // a non-finite beam is a miss: evaluated at the next refill
        } else {
          exhausted = true;
        }
      }
    }
    // phase 1: inner nodes -- left early once at most kTailLanes lanes are still descending while others hold a leaf
    for (;;) {
      const bool inner = (cur != kDone) && !(cur & kLeafBit);
      const uint64_t m_inner = __ballot(inner);
      if (m_inner == 0) break;
      if (static_cast<uint32_t>(__popcll(m_inner)) <= p.tail_lanes && __ballot((cur != kDone) && (cur & kLeafBit)) != 0) break;
      if (inner && kSlotOrder) {
        // children in the ray's slot order (node_hits_q4_so): the first one hit is entered, the later ones wait on the stack, nearest
        // on top; three unconditional stores as in the sorted form
        bool hit[4];
        uint32_t ref[4];
        if (!__any(sp + 3u > static_cast<uint32_t>(kRows))) {
          const uint32_t top = lds_col[(sp - 1u) * kBfStride];
          node_hits_q_so(p.qnodes, cur, rs, best_t, rev_mask, hit, ref);
          const bool h01 = hit[0] || hit[1], h012 = h01 || hit[2];
          lds_col[sp * kBfStride] = ref[3]; sp += (hit[3] && h012) ? 1u : 0u;
          lds_col[sp * kBfStride] = ref[2]; sp += (hit[2] && h01) ? 1u : 0u;
          lds_col[sp * kBfStride] = ref[1]; sp += (hit[1] && hit[0]) ? 1u : 0u;
          const bool any = h012 || hit[3];
          cur = hit[0] ? ref[0] : (hit[1] ? ref[1] : (hit[2] ? ref[2] : (hit[3] ? ref[3] : top)));
          sp = any ? sp : (sp - 1u);
        } else {
          node_hits_q_so(p.qnodes, cur, rs, best_t, rev_mask, hit, ref);
          const bool h01 = hit[0] || hit[1], h012 = h01 || hit[2];
          if (hit[3] && h012) { RMCL_ROW_ST(sp, ref[3]) ++sp; }
          if (hit[2] && h01) { RMCL_ROW_ST(sp, ref[2]) ++sp; }
          if (hit[1] && hit[0]) { RMCL_ROW_ST(sp, ref[1]) ++sp; }
          if (h012 || hit[3]) cur = hit[0] ? ref[0] : (hit[1] ? ref[1] : (hit[2] ? ref[2] : ref[3]));
          else { --sp; cur = RMCL_ROW_LD(sp); }
        }
      } else if (inner) {
        uint32_t key[4], ref[4];
        if (!__any(sp + 3u > static_cast<uint32_t>(kRows))) {
          const uint32_t top = lds_col[(sp - 1u) * kBfStride];
          node_keys_q(p.qnodes, cur, rs, best_t, key, ref);
          RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
          lds_col[sp * kBfStride] = ref[3]; sp += (key[3] != kNone) ? 1u : 0u;
          lds_col[sp * kBfStride] = ref[2]; sp += (key[2] != kNone) ? 1u : 0u;
          lds_col[sp * kBfStride] = ref[1]; sp += (key[1] != kNone) ? 1u : 0u;
          const bool any = key[0] != kNone;
          cur = any ? ref[0] : top;
          sp = any ? sp : (sp - 1u);
        } else {
          node_keys_q(p.qnodes, cur, rs, best_t, key, ref);
          RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
          if (key[3] != kNone) { RMCL_ROW_ST(sp, ref[3]) ++sp; }
          if (key[2] != kNone) { RMCL_ROW_ST(sp, ref[2]) ++sp; }
          if (key[1] != kNone) { RMCL_ROW_ST(sp, ref[1]) ++sp; }
          if (key[0] != kNone) cur = ref[0];
          else { --sp; cur = RMCL_ROW_LD(sp); }
        }
      }
    }
    // phase 2: this lane's leaf (if any); tfar = infinity
    if ((cur != kDone) && (cur & kLeafBit)) {
      if (kLeaf2) leaf_pair(p.tris, cur, O, D, p.ray_tfar, best_t, best_rec);
      else leaf_loop(p.tris, cur, O, D, p.ray_tfar, best_t, best_rec);
      --sp;
      cur = RMCL_ROW_LD(sp);
    }
  }
#undef RMCL_ROW_ST
#undef RMCL_ROW_LD
  __syncthreads();  // every beam of the block has its error in s_eval; the stack rows are free from here on

  const double den = sqrt(static_cast<double>(2 * sq) * 3.14159265358979323846);
  float* s_w = reinterpret_cast<float*>(lds_dyn);
  g1d L = {0.f, 0.f, 0u};
  if (threadIdx.x < np) L = attrs[mine_p].likelihood;
  if (kAccum) {
    if (threadIdx.x < np) {
      const unsigned long long* row = s_acc + threadIdx.x * static_cast<uint32_t>(kAccWords);
      const double p0w = static_cast<double>(s_n0[threadIdx.x]) * s_wu[threadIdx.x];   // prod of every w1: the weight of the history
      const double m0 = static_cast<double>(L.mean);
      const double mean = p0w * m0 + acc_value(row, kAccBins1, kAccEmin1);
      const double S = p0w * (static_cast<double>(L.sigma) + m0 * m0) + acc_value(row + kAccBins1, kAccBins2, kAccEmin2);
      const bool bad = s_bad[threadIdx.x] != 0u;
      L.mean = bad ? __uint_as_float(0x7FC00000u) : static_cast<float>(mean);
      L.sigma = bad ? __uint_as_float(0x7FC00000u) : static_cast<float>(fmax(S - mean * mean, 0.0));   // never negative: see above
    }
  } else if (!evals_global) {
  // dense pass: error -> likelihood (PCDSensorUpdaterEmbree.cpp:224: float argument, double exp / sqrt, float result)
  for (uint32_t i = threadIdx.x; i < nrays; i += 256u) {
    const float error = s_eval[i];
    if (p.errors) {
      if (p.order) { const uint32_t pi = i / p.n_beams; p.errors[static_cast<size_t>(p.order[p0 + pi]) * p.n_beams + (i - pi * p.n_beams)] = error; }
      else p.errors[static_cast<size_t>(p0) * p.n_beams + i] = error;
    }
    const float arg = -(error * error) / sq / 2;
    s_eval[i] = static_cast<float>(exp(static_cast<double>(arg)) / den);
  }

  // in-order merge (sequential semantics of sensorUpdate, :232-238): likelihood += Gaussian1D{eval, 0, 1}; n_meas = min(n_meas, max)
  // after every beam.  The count a particle carries INTO beam k is a_k = n0 (k = 0), n0 + min(k, max - n0) (n0 < max), max
  // (otherwise): both weights of rm::Gaussian1D::operator+= for every (particle, beam) come from all lanes, chunk by chunk,
  // into the stack rows; the chain of one lane per particle is then the multiply-adds of g1d_add in g1d_add's order.
  const uint32_t chunk = min(p.n_beams, (static_cast<uint32_t>(kRows * 256) / np - 2u) / 2u);  // beams per chunk; np <= 64 => >= 39
  const uint32_t wstride = 2u * chunk + 2u;
  for (uint32_t b0 = 0; b0 < p.n_beams; b0 += chunk) {
    const uint32_t nb = min(chunk, p.n_beams - b0);
    __syncthreads();
    for (uint32_t pi = 0; pi < np; ++pi) {
      const uint32_t n0 = s_n0[pi];
      for (uint32_t j = threadIdx.x; j < nb; j += 256u) {
        const uint32_t k = b0 + j;
        const uint32_t a = (k == 0u) ? n0 : ((n0 < p.max_n_meas) ? n0 + min(k, p.max_n_meas - n0) : p.max_n_meas);
        const uint32_t n = a + 1u;
        s_w[pi * wstride + 2u * j] = static_cast<float>(a) / static_cast<float>(n);
        s_w[pi * wstride + 2u * j + 1u] = static_cast<float>(1u) / static_cast<float>(n);
      }
    }
    __syncthreads();
    if (threadIdx.x < np) {
      const float* ev = s_eval + threadIdx.x * p.n_beams + b0;
      const float* w = s_w + threadIdx.x * wstride;
      for (uint32_t j = 0; j < nb; ++j) {
        const float w1 = w[2u * j], w2 = w[2u * j + 1u], e = ev[j];
        const float mean = L.mean * w1 + e * w2;
        const float P1 = L.sigma * w1 + 0.0f * w2;
        const float P2 = ((L.mean - mean) * (L.mean - mean)) * w1 + ((e - mean) * (e - mean)) * w2;
        L.mean = mean;
        L.sigma = P1 + P2;
      }
    }
  }
  } else {
  // The same two steps with the errors in global scratch: chunk by chunk, ALL lanes turn the chunk's errors into likelihoods
  // (same operations, same result) and the two merge weights, three floats per (particle, beam) in the vacated stack rows
  // [ w1 | w2 | likelihood ]; then one lane per particle runs the chain over the chunk from LDS as before.
  const uint32_t chunk = min(p.n_beams, (static_cast<uint32_t>(kRows * 256) / np) / 3u);   // np <= 64, kRows >= 16 => >= 21
  const uint32_t cstride = 3u * chunk;
  for (uint32_t b0 = 0; b0 < p.n_beams; b0 += chunk) {
    const uint32_t nb = min(chunk, p.n_beams - b0);
    const uint32_t cells = np * nb;
    const uint32_t nb_m = (nb == 1u) ? 0u : static_cast<uint32_t>(0x100000000ull / nb) + 1u;   // cell / nb (cells < 2^16 * 2^8)
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < cells; c += 256u) {
      const uint32_t pi = (nb == 1u) ? c : __umulhi(c, nb_m), j = c - pi * nb, k = b0 + j;
      const float error = s_eval[pi * p.n_beams + k];
      if (p.errors) {
        const uint32_t dst = p.order ? p.order[p0 + pi] : (p0 + pi);
        p.errors[static_cast<size_t>(dst) * p.n_beams + k] = error;
      }
      const float arg = -(error * error) / sq / 2;
      const uint32_t n0 = s_n0[pi];
      const uint32_t a = (k == 0u) ? n0 : ((n0 < p.max_n_meas) ? n0 + min(k, p.max_n_meas - n0) : p.max_n_meas);
      const uint32_t n = a + 1u;
      float* cell = s_w + pi * cstride + 3u * j;
      cell[0] = static_cast<float>(a) / static_cast<float>(n);
      cell[1] = static_cast<float>(1u) / static_cast<float>(n);
      cell[2] = static_cast<float>(exp(static_cast<double>(arg)) / den);
    }
    __syncthreads();
    if (threadIdx.x < np) {
      const float* w = s_w + threadIdx.x * cstride;
      for (uint32_t j = 0; j < nb; ++j) {
        const float w1 = w[3u * j], w2 = w[3u * j + 1u], e = w[3u * j + 2u];
        const float mean = L.mean * w1 + e * w2;
        const float P1 = L.sigma * w1 + 0.0f * w2;
        const float P2 = ((L.mean - mean) * (L.mean - mean)) * w1 + ((e - mean) * (e - mean)) * w2;
        L.mean = mean;
        L.sigma = P1 + P2;
      }
    }
  }
  }
  if (threadIdx.x < np) {
    const uint32_t n0 = s_n0[threadIdx.x], k = p.n_beams;
    L.n_meas = (n0 < p.max_n_meas) ? n0 + min(k, p.max_n_meas - n0) : p.max_n_meas;
    attrs[mine_p].likelihood = L;
  }
}

// particle_move_and_forget_kernel (rmcl_ros/src/rmcl/particle_motion.cu:11-34) + the wall-collision test of the
// CPU updater (collision_in_between, TFMotionUpdaterCPU.cpp:17-50,207-221): one lane per particle; the occlusion
// ray runs from the old to the new particle position with tfar = segment length.
template <bool kCollision>
__global__ void __launch_bounds__(256) k_pf_motion(const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ tris,
                                                   xform* __restrict__ poses, pattrs* __restrict__ attrs, uint32_t n,
                                                   xform T_bnew_bold, double forget_rate, uint32_t max_n_meas) {
  extern __shared__ uint32_t lds_dyn[];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const uint32_t ii = live ? i : 0u;
  const xform pose_old = poses[ii];
  g1d L = attrs[ii].likelihood;
  const xform pose_new = xmul(pose_old, T_bnew_bold);
  // `n_meas -= forget_rate * n_meas` on a uint32: double arithmetic, truncating store
  L.n_meas = static_cast<uint32_t>(static_cast<double>(L.n_meas) - forget_rate * static_cast<double>(L.n_meas));
  if (kCollision) {
    f3 vec = sub3(pose_new.t, pose_old.t);
    const float length = sqrtf((vec.x * vec.x + vec.y * vec.y) + vec.z * vec.z);
    const bool moving = !(static_cast<double>(length) < 0.00001);
    vec = mk3(vec.x / length, vec.y / length, vec.z / length);
    RayHit h;
    trace_lane_bf<16, true>(nodes, tris, pose_old.t, vec, (live && moving) ? length : -1.0f, lds_dyn + threadIdx.x, h);
    if (moving && h.rec != kNone) { L.mean = 0.0f; L.sigma = 0.0f; L.n_meas = max_n_meas; }
  }
  if (live) {
    poses[i] = pose_new;
    attrs[i].likelihood = L;
  }
}

__global__ void k_pf_extract_weights(const pattrs* __restrict__ attrs, uint32_t n, float* __restrict__ w) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = attrs[i].likelihood.mean;
}

// ---------------------------------------------------------------------------------------------
// gladiator resampling (resampling.cu:41-219).  Random stream = Philox4x32-10 keyed by the seed with counter
// (champion index, step, draw, 0): reproducible, independent of the launch shape and of how the particle range
// is sharded across GPUs.  Transcendentals are evaluated in double and rounded to float (see oracle).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const double u1 = (static_cast<double>(a) + 0.5) * (1.0 / 4294967296.0);
  const double u2 = (static_cast<double>(b) + 0.5) * (1.0 / 4294967296.0);
  const double r = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925 * u2;
  z0 = static_cast<float>(r * cos(ang));
  z1 = static_cast<float>(r * sin(ang));
}

struct GladiatorConfig {
  float min_noise_tx, min_noise_ty, min_noise_tz, min_noise_roll, min_noise_pitch, min_noise_yaw;
  float likelihood_forget_per_meter, likelihood_forget_per_radian;
  uint32_t trans_dist_metric;
};

__global__ void __launch_bounds__(256) k_gladiator_resample(const xform* __restrict__ poses, const pattrs* __restrict__ attrs,
                                                            uint32_t n, xform* __restrict__ poses_new,
                                                            pattrs* __restrict__ attrs_new, uint32_t first, uint32_t count,
                                                            GladiatorConfig cfg, uint32_t key0, uint32_t key1,
                                                            uint32_t step) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const uint32_t champion = first + k;
  uint32_t ra[4], rb[4];
  philox4x32_10(champion, step, 0u, 0u, key0, key1, ra);
  philox4x32_10(champion, step, 1u, 0u, key0, key1, rb);
  const uint32_t enemy = ra[0] % n;
  const float Lc = attrs[champion].likelihood.mean, Le = attrs[enemy].likelihood.mean;
  if (Le > Lc) {
    float Nd_tx, Nd_ty, Nd_tz, Nd_rx, Nd_ry, Nd_rz;
    box_muller(ra[1], ra[2], Nd_tx, Nd_ty);
    box_muller(ra[3], rb[0], Nd_tz, Nd_rx);
    box_muller(rb[1], rb[2], Nd_ry, Nd_rz);
    const xform pose = poses[enemy];
    xform pn = pose;
    pattrs an = attrs[enemy];
    pn.t.x = pn.t.x + Nd_tx * cfg.min_noise_tx;
    pn.t.y = pn.t.y + Nd_ty * cfg.min_noise_ty;
    pn.t.z = pn.t.z + Nd_tz * cfg.min_noise_tz;
    // EulerAngles e = pose_new.R (textbook ZYX extraction)
    const quat q = pn.R;
    const float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
    const float cosr_cosp = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
    const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
    const float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
    const float cosy_cosp = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
    float roll = static_cast<float>(atan2(static_cast<double>(sinr_cosp), static_cast<double>(cosr_cosp)));
    float pitch = (fabsf(sinp) >= 1.0f) ? copysignf(static_cast<float>(3.14159265358979323846 / 2.0), sinp)
                                        : static_cast<float>(asin(static_cast<double>(sinp)));
    float yaw = static_cast<float>(atan2(static_cast<double>(siny_cosp), static_cast<double>(cosy_cosp)));
    roll = roll + Nd_rx * cfg.min_noise_roll;
    pitch = pitch + Nd_ry * cfg.min_noise_pitch;
    yaw = yaw + Nd_rz * cfg.min_noise_yaw;
    // pose_new.R = e
    const float cr = static_cast<float>(cos(static_cast<double>(roll / 2.0f))), sr = static_cast<float>(sin(static_cast<double>(roll / 2.0f)));
    const float cp = static_cast<float>(cos(static_cast<double>(pitch / 2.0f))), sp = static_cast<float>(sin(static_cast<double>(pitch / 2.0f)));
    const float cy = static_cast<float>(cos(static_cast<double>(yaw / 2.0f))), sy = static_cast<float>(sin(static_cast<double>(yaw / 2.0f)));
    pn.R.w = cr * cp * cy + sr * sp * sy;
    pn.R.x = sr * cp * cy - cr * sp * sy;
    pn.R.y = cr * sp * cy + sr * cp * sy;
    pn.R.z = cr * cp * sy - sr * sp * cy;
    const xform diff = xmul(xinv(pose), pn);
    const float t2 = (diff.t.x * diff.t.x + diff.t.y * diff.t.y) + diff.t.z * diff.t.z;
    const float trans_dist = (cfg.trans_dist_metric == 1u) ? t2 : sqrtf(t2);
    const float rot_dist = sqrtf(((diff.R.w * diff.R.w + diff.R.x * diff.R.x) + diff.R.y * diff.R.y) + diff.R.z * diff.R.z);
    const float frs = static_cast<float>(1.0 - pow(1.0 - static_cast<double>(cfg.likelihood_forget_per_meter), static_cast<double>(trans_dist)));
    const float frr = static_cast<float>(1.0 - pow(1.0 - static_cast<double>(cfg.likelihood_forget_per_radian), static_cast<double>(rot_dist)));
    const float forget_rate = (frs > frr) ? frs : frr;
    const float remember_rate = static_cast<float>(1.0 - static_cast<double>(forget_rate));
    an.likelihood.n_meas = static_cast<uint32_t>(static_cast<float>(an.likelihood.n_meas) * remember_rate);
    poses_new[k] = pn;
    attrs_new[k] = an;
  } else {
    poses_new[k] = poses[champion];
    attrs_new[k] = attrs[champion];
  }
}

// ---------------------------------------------------------------------------------------------
// residual resampling (ResidualResamplerCPU.cpp:55-203) -- the reference's SEQUENTIAL loop "draw a particle, insert
// size_t(L / sum * N_new) perturbed copies, until the new cloud is full" as four data-parallel passes over a block of draws:
//   counts   c_k = copies draw k inserts (the draw's particle and its share; independent of every other draw),
//   scan     inclusive prefix sums of c_k (64-bit): draw k fills slots [incl_k - c_k, incl_k),
//   fill     slot j finds its draw by binary search, perturbs the copy with ITS Gaussians (Philox counter = slot index).
// Same stream, same arithmetic as oracle/rmcl_oracle.c: orc_residual_resample (which restates the loop statement by statement):
// particles, likelihoods and n_meas bit-exact, poses to float rounding of the double-evaluated transcendentals.
// ---------------------------------------------------------------------------------------------
struct ResidualStats {
  double sum, max;
  unsigned long long expect;   // sum over the particles of their share's integer part = n * E[c_k]
  unsigned long long n_draws;  // written by k_residual_fill: draws the sequential loop would have used
};

__global__ void __launch_bounds__(64) k_residual_stats_final(const double* __restrict__ psum, const float* __restrict__ pmax,
                                                            uint32_t nblocks, ResidualStats* __restrict__ out) {
  // fixed order: lane l sums blocks l, l + 64, ...; then a fixed butterfly
  double s = 0.0;
  float m = 0.0f;
  for (uint32_t b = threadIdx.x; b < nblocks; b += 64u) { s += psum[b]; m = fmaxf(m, pmax[b]); }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off);
    m = fmaxf(m, __shfl_down(m, off));
  }
  if (threadIdx.x == 0) { out->sum = s; out->max = static_cast<double>(m); out->expect = 0ull; out->n_draws = 0ull; }
}

// copies a draw of particle likelihood L inserts when `left` slots are free: the reference's size_t(L / sum * N_new), clamped
__device__ __forceinline__ uint32_t residual_share(float Lf, double weight_sum, uint32_t n_new) {
  const double share = (static_cast<double>(Lf) / weight_sum) * static_cast<double>(n_new);
  if (!(share > 0.0)) return 0u;
  return (share >= static_cast<double>(n_new)) ? n_new : static_cast<uint32_t>(share);
}

__global__ void __launch_bounds__(256) k_residual_expect(const pattrs* __restrict__ attrs, uint32_t n, uint32_t n_new,
                                                         ResidualStats* __restrict__ st) {
  __shared__ unsigned long long s_part[4];
  const double sum = st->sum;
  unsigned long long acc = 0ull;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc += residual_share(attrs[i].likelihood.mean, sum, n_new);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&st->expect, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));   // integers: order independent
}

__global__ void __launch_bounds__(256) k_residual_counts(const pattrs* __restrict__ attrs, uint32_t n, uint32_t n_new,
                                                         const ResidualStats* __restrict__ st, uint32_t n_draws, uint32_t key0,
                                                         uint32_t key1, uint32_t step, uint32_t* __restrict__ idx_out,
                                                         uint32_t* __restrict__ cnt_out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_draws) return;
  uint32_t r[4];
  philox4x32_10(k, step, 2u, 0u, key0, key1, r);
  const uint32_t random_index = r[0] % n;
  idx_out[k] = random_index;
  cnt_out[k] = residual_share(attrs[random_index].likelihood.mean, st->sum, n_new);
}

// inclusive 64-bit prefix sums of 32-bit counts, three passes: 1024 elements per block -> block totals -> totals scanned by ONE
// block -> added back.  (Counts are <= N_new each, so 32 bits would overflow for peaked weights.)
__device__ __forceinline__ unsigned long long block_scan_256(unsigned long long v, unsigned long long* s_wave, unsigned long long& total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(incl, off, 64);
    if (lane >= static_cast<uint32_t>(off)) incl += o;
  }
  if (lane == 63u) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long base = 0ull;
  for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
  total = ((s_wave[0] + s_wave[1]) + s_wave[2]) + s_wave[3];
  __syncthreads();
  return base + incl;
}

__global__ void __launch_bounds__(256) k_scan_blocks(const uint32_t* __restrict__ cnt, uint32_t n, unsigned long long* __restrict__ incl,
                                                     unsigned long long* __restrict__ block_total) {
  __shared__ unsigned long long s_wave[4];
  const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
  unsigned long long c[4];
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) c[u] = (base + u < n) ? cnt[base + u] : 0u;
  const unsigned long long mine = (c[0] + c[1]) + (c[2] + c[3]);
  unsigned long long total;
  const unsigned long long end = block_scan_256(mine, s_wave, total);   // inclusive over the threads
  unsigned long long run = end - mine;
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) {
    run += c[u];
    if (base + u < n) incl[base + u] = run;
  }
  if (threadIdx.x == 0) block_total[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_scan_totals(unsigned long long* __restrict__ block_total, uint32_t nblocks) {
  __shared__ unsigned long long s_wave[4];
  unsigned long long carry = 0ull;
  for (uint32_t b0 = 0; b0 < nblocks; b0 += 256u) {
    const uint32_t b = b0 + threadIdx.x;
    const unsigned long long v = (b < nblocks) ? block_total[b] : 0ull;
    unsigned long long total;
    const unsigned long long inc = block_scan_256(v, s_wave, total);
    if (b < nblocks) block_total[b] = carry + inc - v;   // exclusive: what precedes block b
    carry += total;
  }
}

__global__ void __launch_bounds__(256) k_scan_add(unsigned long long* __restrict__ incl, uint32_t n, const unsigned long long* __restrict__ block_excl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) incl[i] += block_excl[i >> 10];
}

__global__ void __launch_bounds__(256) k_residual_fill(const xform* __restrict__ poses, const pattrs* __restrict__ attrs,
                                                       const uint32_t* __restrict__ draw_idx, const unsigned long long* __restrict__ incl,
                                                       uint32_t n_draws, xform* __restrict__ poses_new, pattrs* __restrict__ attrs_new,
                                                       uint32_t n_new, uint32_t first, uint32_t count, GladiatorConfig cfg,
                                                       ResidualStats* __restrict__ st, uint32_t key0, uint32_t key1, uint32_t step) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const uint32_t j = first + t;                 // global output slot
  // the draw that fills slot j: the first k with incl[k] > j (the host launches this only when incl[n_draws - 1] >= n_new)
  uint32_t lo = 0u, hi = n_draws - 1u;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (incl[mid] > static_cast<unsigned long long>(j)) hi = mid; else lo = mid + 1u;
  }
  const uint32_t k = lo;
  if (j + 1u == n_new) st->n_draws = static_cast<unsigned long long>(k) + 1ull;   // the sequential loop stops after this draw
  const uint32_t src = draw_idx[k];
  const xform pose = poses[src];
  pattrs an = attrs[src];
  const double L_max_normed = static_cast<double>(an.likelihood.mean) / st->max;
  uint32_t ra[4], rb[4];
  philox4x32_10(j, step, 3u, 0u, key0, key1, ra);
  philox4x32_10(j, step, 4u, 0u, key0, key1, rb);
  float Nd_tx, Nd_ty, Nd_tz, Nd_rx, Nd_ry, Nd_rz;
  box_muller(ra[0], ra[1], Nd_tx, Nd_ty);
  box_muller(ra[2], ra[3], Nd_tz, Nd_rx);
  box_muller(rb[0], rb[1], Nd_ry, Nd_rz);
  const float noise_tx = static_cast<float>(static_cast<double>(cfg.min_noise_tx) / L_max_normed);
  const float noise_ty = static_cast<float>(static_cast<double>(cfg.min_noise_ty) / L_max_normed);
  const float noise_tz = static_cast<float>(static_cast<double>(cfg.min_noise_tz) / L_max_normed);
  const float noise_roll = static_cast<float>(static_cast<double>(cfg.min_noise_roll) / L_max_normed);
  const float noise_pitch = static_cast<float>(static_cast<double>(cfg.min_noise_pitch) / L_max_normed);
  const float noise_yaw = static_cast<float>(static_cast<double>(cfg.min_noise_yaw) / L_max_normed);
  xform pn = pose;
  pn.t.x = pn.t.x + Nd_tx * noise_tx;
  pn.t.y = pn.t.y + Nd_ty * noise_ty;
  pn.t.z = pn.t.z + Nd_tz * noise_tz;
  // EulerAngles <- Quaternion (textbook ZYX extraction, as in k_gladiator_resample)
  const quat q = pn.R;
  const float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
  const float cosr_cosp = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
  const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
  const float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
  const float cosy_cosp = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
  float roll = static_cast<float>(atan2(static_cast<double>(sinr_cosp), static_cast<double>(cosr_cosp)));
  float pitch = (fabsf(sinp) >= 1.0f) ? copysignf(static_cast<float>(3.14159265358979323846 / 2.0), sinp)
                                      : static_cast<float>(asin(static_cast<double>(sinp)));
  float yaw = static_cast<float>(atan2(static_cast<double>(siny_cosp), static_cast<double>(cosy_cosp)));
  roll = roll + Nd_rx * noise_roll;
  pitch = pitch + Nd_ry * noise_pitch;
  yaw = yaw + Nd_rz * noise_yaw;
  const float cr = static_cast<float>(cos(static_cast<double>(roll / 2.0f))), sr = static_cast<float>(sin(static_cast<double>(roll / 2.0f)));
  const float cp = static_cast<float>(cos(static_cast<double>(pitch / 2.0f))), sp = static_cast<float>(sin(static_cast<double>(pitch / 2.0f)));
  const float cy = static_cast<float>(cos(static_cast<double>(yaw / 2.0f))), sy = static_cast<float>(sin(static_cast<double>(yaw / 2.0f)));
  pn.R.w = cr * cp * cy + sr * sp * sy;
  pn.R.x = sr * cp * cy - cr * sp * sy;
  pn.R.y = cr * sp * cy + sr * cp * sy;
  pn.R.z = cr * cp * sy - sr * sp * cy;
  const xform diff = xmul(xinv(pose), pn);
  const float trans_dist = (diff.t.x * diff.t.x + diff.t.y * diff.t.y) + diff.t.z * diff.t.z;   // l2normSquared (:164)
  const float rot_dist = sqrtf(((diff.R.w * diff.R.w + diff.R.x * diff.R.x) + diff.R.y * diff.R.y) + diff.R.z * diff.R.z);
  const float reduction_factor = static_cast<float>(pow(static_cast<double>(cfg.likelihood_forget_per_meter), static_cast<double>(trans_dist))) *
                                 static_cast<float>(pow(static_cast<double>(cfg.likelihood_forget_per_radian), static_cast<double>(rot_dist)));
  an.likelihood.n_meas = static_cast<uint32_t>(static_cast<float>(an.likelihood.n_meas) * reduction_factor);
  poses_new[t] = pn;
  attrs_new[t] = an;
}

// simple_stats_kernel (resampling.cu:41-81): {sum, max} of likelihood.mean; max seeded with 0 like the reference's
// shared-memory init, sum accumulated in double.  Stage 1: <=256 blocks of grid-stride partials; stage 2: one wave.
// likelihoods: `first` + i * stride floats -- the likelihood.mean members of an attribute array (stride 9) or a dense weight vector
// (stride 1: what the sharded filter's all-gather leaves on every rank); the summation order depends on n alone, so both forms of the
// same n values give the same bits
__global__ void __launch_bounds__(256) k_likelihood_stats_partial(const float* __restrict__ first, uint32_t stride, uint32_t n,
                                                                  double* __restrict__ psum, float* __restrict__ pmax) {
  __shared__ double s_sum[4];
  __shared__ float s_max[4];
  double sum = 0.0;
  float mx = 0.0f;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float L = first[static_cast<size_t>(i) * stride];
    sum += static_cast<double>(L);
    mx = (L > mx) ? L : mx;
  }
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off);
    const float o = __shfl_down(mx, off);
    mx = (o > mx) ? o : mx;
  }
  if ((threadIdx.x & 63u) == 0u) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    psum[blockIdx.x] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
    pmax[blockIdx.x] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  }
}

__global__ void __launch_bounds__(64) k_likelihood_stats_final(const double* __restrict__ psum, const float* __restrict__ pmax,
                                                               uint32_t nblocks, float* __restrict__ out) {
  double sum = 0.0;
  float mx = 0.0f;
  for (uint32_t i = threadIdx.x; i < nblocks; i += 64u) {
    sum += psum[i];
    mx = fmaxf(mx, pmax[i]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off);
    mx = fmaxf(mx, __shfl_down(mx, off));
  }
  if (threadIdx.x == 0) { out[0] = static_cast<float>(sum); out[1] = mx; }
}

// ---------------------------------------------------------------------------------------------
// pose estimate of the particle cloud (RmclNode::estimateStats, rmcl_ros/src/nodes/rmcl_localization.cpp:642-731) as
// three linear moment passes, so that it shards: every GPU reduces its block of the particles to <= 24 doubles and the
// ranks all-reduce them ("Stats for pose estimate: all-reduce of ~32 floats").
//   pass 0: sum L, sum L^2, count | max L, -min L, bb_max xyz, -bb_min xyz (max-reduced)           (:664-689)
//   pass 1: sum w q q^T (10 unique entries, x y z w order), sum w t, with w = L / L_sum                  (:703-705)
//           -- the Markley mean is the eigenvector of the largest eigenvalue of that 4x4 matrix
//   pass 2: sum w d d^T (21 unique entries), d = (dt, roll, pitch, yaw) of ~Tbm * T_i                    (:716-718)
// Partials per block: 24 sums (double) + 8 maxima (float as double).
// ---------------------------------------------------------------------------------------------
constexpr int kMomSums = 24, kMomMax = 8;

__global__ void __launch_bounds__(256) k_pose_moments(const xform* __restrict__ poses, const pattrs* __restrict__ attrs, uint32_t n,
                                                      int pass, double L_sum, xform Tbm, double* __restrict__ partials) {
  __shared__ double red[4][kMomSums + kMomMax];
  double acc[kMomSums];
  float mx[kMomMax];
#pragma unroll
  for (int k = 0; k < kMomSums; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < kMomMax; ++k) mx[k] = -3.402823466e38f;
  const xform Tmb = xinv(Tbm);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const xform T = poses[i];
    const float Lf = attrs[i].likelihood.mean;
    const double L = static_cast<double>(Lf);
    if (pass == 0) {
      acc[0] += L; acc[1] += L * L; acc[2] += 1.0;
      mx[0] = fmaxf(mx[0], Lf); mx[1] = fmaxf(mx[1], -Lf);
      mx[2] = fmaxf(mx[2], T.t.x); mx[3] = fmaxf(mx[3], T.t.y); mx[4] = fmaxf(mx[4], T.t.z);
      mx[5] = fmaxf(mx[5], -T.t.x); mx[6] = fmaxf(mx[6], -T.t.y); mx[7] = fmaxf(mx[7], -T.t.z);
    } else if (pass == 1) {
      const double w = L / L_sum;
      const double q[4] = {T.R.x, T.R.y, T.R.z, T.R.w};
      int k = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) acc[k++] += w * q[a] * q[b];
      acc[10] += w * T.t.x; acc[11] += w * T.t.y; acc[12] += w * T.t.z;
    } else {
      const double w = L / L_sum;
      const xform Td = xmul(Tmb, T);
      // EulerAngles <- Quaternion (textbook ZYX extraction, as in k_gladiator_resample)
      const quat qd = Td.R;
      const float sinr_cosp = 2.0f * (qd.w * qd.x + qd.y * qd.z), cosr_cosp = 1.0f - 2.0f * (qd.x * qd.x + qd.y * qd.y);
      const float sinp = 2.0f * (qd.w * qd.y - qd.z * qd.x);
      const float siny_cosp = 2.0f * (qd.w * qd.z + qd.x * qd.y), cosy_cosp = 1.0f - 2.0f * (qd.y * qd.y + qd.z * qd.z);
      const double d[6] = {Td.t.x, Td.t.y, Td.t.z,
                           atan2(static_cast<double>(sinr_cosp), static_cast<double>(cosr_cosp)),
                           (fabsf(sinp) >= 1.0f) ? copysign(3.14159265358979323846 / 2.0, static_cast<double>(sinp)) : asin(static_cast<double>(sinp)),
                           atan2(static_cast<double>(siny_cosp), static_cast<double>(cosy_cosp))};
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) acc[k++] += w * d[a] * d[b];
    }
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kMomSums; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0u) red[wave][k] = v;
  }
#pragma unroll
  for (int k = 0; k < kMomMax; ++k) {
    float v = mx[k];
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if (lane == 0u) red[wave][kMomSums + k] = static_cast<double>(v);
  }
  __syncthreads();
  if (threadIdx.x < kMomSums + kMomMax) {
    const int k = static_cast<int>(threadIdx.x);
    double v;
    if (k < kMomSums) v = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    else v = fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]));
    partials[static_cast<size_t>(blockIdx.x) * (kMomSums + kMomMax) + k] = v;
  }
}

// One workgroup folds the per-block partials: thread = (moment k, group g of 8); a group takes every 8th block with four loads in
// flight, the eight group sums are folded in a fixed order through LDS (deterministic).  Round 2's version -- one 64-lane wave,
// one dependent load per block and lane -- took 23.7 us for 98 blocks, twice the streaming pass it finalizes.
__global__ void __launch_bounds__(256) k_pose_moments_final(const double* __restrict__ partials, uint32_t nblocks, double* __restrict__ out) {
  constexpr int kW = kMomSums + kMomMax;   // 32
  __shared__ double s_g[8][kW];
  const int k = static_cast<int>(threadIdx.x) & (kW - 1);
  const uint32_t g = threadIdx.x >> 5;
  const bool is_sum = k < kMomSums;
  double v = is_sum ? 0.0 : -1.7976931348623157e308;
  uint32_t b = g;
  for (; b + 24u < nblocks; b += 32u) {
    const double x0 = partials[static_cast<size_t>(b) * kW + k], x1 = partials[static_cast<size_t>(b + 8u) * kW + k];
    const double x2 = partials[static_cast<size_t>(b + 16u) * kW + k], x3 = partials[static_cast<size_t>(b + 24u) * kW + k];
    v = is_sum ? (v + ((x0 + x1) + (x2 + x3))) : fmax(fmax(v, fmax(x0, x1)), fmax(x2, x3));
  }
  for (; b < nblocks; b += 8u) {
    const double x = partials[static_cast<size_t>(b) * kW + k];
    v = is_sum ? (v + x) : fmax(v, x);
  }
  s_g[g][k] = v;
  __syncthreads();
  if (threadIdx.x < static_cast<uint32_t>(kW)) {
    double r = s_g[0][k];
#pragma unroll
    for (int q = 1; q < 8; ++q) r = is_sum ? (r + s_g[q][k]) : fmax(r, s_g[q][k]);
    out[k] = r;
  }
}

// dense weight vector from the padded all-gather layout: rank r's shard sits at r * cap
__global__ void k_compact_shards(const float* __restrict__ padded, float* __restrict__ dense, uint32_t n_total, uint32_t world, uint32_t cap) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  const uint32_t base = n_total / world, rem = n_total % world;
  // owner of global index i under the block partition (the first `rem` ranks own base + 1)
  const uint32_t split = rem * (base + 1u);
  const uint32_t r = (i < split) ? (i / (base + 1u)) : (rem + (i - split) / max(base, 1u));
  const uint32_t lo = r * base + min(r, rem);
  dense[i] = padded[static_cast<size_t>(r) * cap + (i - lo)];
}

// the same for records of `wpr` dwords (poses: 8, attributes: 9): one thread per dword
__global__ void k_compact_records(const uint32_t* __restrict__ padded, uint32_t* __restrict__ dense, uint32_t n_total, uint32_t world, uint32_t cap,
                                  uint32_t wpr) {
  const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= static_cast<size_t>(n_total) * wpr) return;
  const uint32_t i = static_cast<uint32_t>(g / wpr), w = static_cast<uint32_t>(g - static_cast<size_t>(i) * wpr);
  const uint32_t base = n_total / world, rem = n_total % world;
  const uint32_t split = rem * (base + 1u);
  const uint32_t r = (i < split) ? (i / (base + 1u)) : (rem + (i - split) / max(base, 1u));
  const uint32_t lo = r * base + min(r, rem);
  dense[g] = padded[(static_cast<size_t>(r) * cap + (i - lo)) * wpr + w];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
namespace {
const LabHooks* g_lab = nullptr;
}
const LabHooks* lab_hooks() { return g_lab; }

// ---------------------------------------------------------------------------------------------
// World order of a pose batch (VERDICT r5 #7b): key of wave-tile (pose, tile) = Morton code (4 bits per axis) of the point where the
// tile's central ray leaves the map's bounding box -- for a sensor inside its map that is where the tile's rays end up, whatever pose
// they start from.  4096 cells and a counting sort (count while the keys are made, one block scans the cells, scatter): three small
// launches, ~15 us for the 228 k tiles of the reference's v1 batch; rocPRIM's radix sort of 30-bit keys took 16 launches and 100 us.
// The order inside a cell is whatever the atomics make it -- it decides which workgroup computes a tile, never what is computed.
// k_find then walks the tiles in that order (FindParams::tile_order).
// ---------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kOrderCells = 4096u;
__device__ __forceinline__ uint32_t morton_spread4(uint32_t v) {   // bits 0..3 -> bits 0, 3, 6, 9
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
template <uint32_t kModel>
__global__ void __launch_bounds__(256) k_batch_tile_keys(const FindParams p, uint32_t group, uint32_t ngroups, f3 bb_min, f3 bb_max,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ cell_count) {
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= ngroups * p.nposes) return;
  const uint32_t pose = i / ngroups, tile = min((i - pose * ngroups) * group + (group >> 1), ntiles - 1u);   // the middle tile of the workgroup's tiles
  const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const uint32_t twl = p.tile_w_log2;
  const uint32_t cv = min((ty << (6u - twl)) + (32u >> twl), p.H - 1u), ch = min((tx << twl) + ((1u << twl) >> 1), p.W - 1u);
  f3 dir_s, orig_s = p.orig_s;
  find_ray_s<kModel>(p, cv, ch, cv * p.W + ch, dir_s, orig_s);
  const xform Tsm = p.Tsm_arr[pose];
  const f3 o = xapply(Tsm, orig_s), d = qrot(Tsm.R, dir_s);
  float t = 3.0e38f;
  const float od[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, lo[3] = {bb_min.x, bb_min.y, bb_min.z}, hi[3] = {bb_max.x, bb_max.y, bb_max.z};
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (dd[a] != 0.0f) t = fminf(t, fmaxf(((dd[a] > 0.0f ? hi[a] : lo[a]) - od[a]) / dd[a], 0.0f));
  if (!(t < 3.0e38f)) t = 0.0f;   // (NaN directions, a ray without direction: the origin's cell)
  uint32_t q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float u = (od[a] + t * dd[a] - lo[a]) / fmaxf(hi[a] - lo[a], 1e-20f);
    q[a] = static_cast<uint32_t>(fminf(fmaxf(u, 0.0f), 1.0f) * 15.0f);   // (NaN -> 0)
  }
  const uint32_t key = morton_spread4(q[0]) | (morton_spread4(q[1]) << 1) | (morton_spread4(q[2]) << 2);
  keys[i] = key;
  atomicAdd(cell_count + key, 1u);
}
// exclusive scan of the 4096 cell counts, in place (one block; the counts become the cells' first slots)
__global__ void __launch_bounds__(256) k_batch_cell_scan(uint32_t* __restrict__ cell) {
  __shared__ uint32_t s_sum[256];
  const uint32_t t = threadIdx.x;
  uint32_t v[16], run = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { v[k] = cell[t * 16u + k]; run += v[k]; }
  s_sum[t] = run;
  __syncthreads();
  for (uint32_t off = 1; off < 256u; off <<= 1) {
    const uint32_t add = (t >= off) ? s_sum[t - off] : 0u;
    __syncthreads();
    s_sum[t] += add;
    __syncthreads();
  }
  uint32_t base = s_sum[t] - run;
#pragma unroll
  for (int k = 0; k < 16; ++k) { cell[t * 16u + k] = base; base += v[k]; }
}
__global__ void __launch_bounds__(256) k_batch_tile_scatter(const uint32_t* __restrict__ keys, uint32_t* __restrict__ cell_next, uint32_t ngroups,
                                                            uint32_t n, uint32_t* __restrict__ order) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t pose = i / ngroups, g = i - pose * ngroups;
  order[atomicAdd(cell_next + keys[i], 1u)] = (pose << 16) | g;
}
}  // namespace

uint32_t batch_order_scratch_dwords(uint32_t n) { return n + kOrderCells; }

// One entry per WORKGROUP of the find (its `group` = 4 consecutive tiles of one pose -- 1 for the quad kind --: neighbours in the scan image
// stay together, their waves share the CU's L1 as in the pose-major launch; sorting single tiles cost the 100 k-face map 6 %).
// scratch: n keys | 4096 cells; order: n = nposes x ceil(ntiles / group) entries pose << 16 | group index
hipError_t launch_batch_tile_order(const FindParams& p, ModelKind kind, uint32_t group, f3 bb_min, f3 bb_max, uint32_t* scratch, uint32_t* order,
                                   hipStream_t s) {
  const uint32_t ntiles = p.tiles_x * p.tiles_y, ngroups = (ntiles + group - 1u) / group, n = ngroups * p.nposes;
  uint32_t* keys = scratch;
  uint32_t* cells = scratch + n;
  hipError_t e = hipMemsetAsync(cells, 0, kOrderCells * sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  const dim3 grid((n + 255u) / 256u), block(256);
  switch (kind) {
    case kModelSpherical: hipLaunchKernelGGL((k_batch_tile_keys<kModelSpherical>), grid, block, 0, s, p, group, ngroups, bb_min, bb_max, keys, cells); break;
    case kModelO1Dn: hipLaunchKernelGGL((k_batch_tile_keys<kModelO1Dn>), grid, block, 0, s, p, group, ngroups, bb_min, bb_max, keys, cells); break;
    case kModelPinhole: hipLaunchKernelGGL((k_batch_tile_keys<kModelPinhole>), grid, block, 0, s, p, group, ngroups, bb_min, bb_max, keys, cells); break;
    case kModelOnDn: hipLaunchKernelGGL((k_batch_tile_keys<kModelOnDn>), grid, block, 0, s, p, group, ngroups, bb_min, bb_max, keys, cells); break;
    default: return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_batch_cell_scan, dim3(1), block, 0, s, cells);
  hipLaunchKernelGGL(k_batch_tile_scatter, grid, block, 0, s, keys, cells, ngroups, n, order);
  return hipGetLastError();
}

hipError_t launch_find(const FindParams& p, ModelKind kind, int variant, hipStream_t s) {
  if (!find_kind_in_product(variant) || p.wave_clock != nullptr) {
    // an experiment's kind, or a clocked launch of any kind (tools/wave_timeline.py): librmclhip_lab.so
    if (g_lab && g_lab->find) return g_lab->find(p, kind, variant, p.wave_clock != nullptr, s);
    return kLabMissing;
  }
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  uint32_t nblocks = (variant == 2) ? ntiles : (ntiles + 3u) / 4u;
  nblocks = (nblocks + 7u) & ~7u;  // the XCD remap in k_find needs gridDim.x % 8 == 0
  dim3 grid(nblocks, p.nposes, 1), block(256, 1, 1);
  if (p.tile_order != nullptr) {   // world order: one row of blocks over the sorted (pose, tile) list
    const uint32_t nb = p.n_tile_order, turn = 8u * max(p.tile_order_granule, 1u);   // (one entry per workgroup)
    grid = dim3((nb + turn - 1u) / turn * turn, 1, 1);   // whole turns of the eight XCDs (k_find's slot mapping)
  }
#define RMCL_LAUNCH_FIND(TRAV, LDS)                                                                                   \
  switch (kind) {                                                                                                     \
    case kModelSpherical: hipLaunchKernelGGL((k_find<kModelSpherical, TRAV>), grid, block, LDS, s, p); break;         \
    case kModelO1Dn: hipLaunchKernelGGL((k_find<kModelO1Dn, TRAV>), grid, block, LDS, s, p); break;                   \
    case kModelPinhole: hipLaunchKernelGGL((k_find<kModelPinhole, TRAV>), grid, block, LDS, s, p); break;             \
    case kModelOnDn: hipLaunchKernelGGL((k_find<kModelOnDn, TRAV>), grid, block, LDS, s, p); break;                   \
    default: return hipErrorInvalidValue;                                                                             \
  }
  if (variant == 0) {  // wave-packet traversal (needs map stack_need <= 64, checked at map creation)
    RMCL_LAUNCH_FIND(0, 0)
  } else if (variant == 2) {  // quad-cooperative: 64 rays per block, 64 stack entries per ray in LDS
    const size_t lds = kQuadStackEntries * 64u * sizeof(uint32_t);
    RMCL_LAUNCH_FIND(2, lds)
  } else if (variant == 23) {  // one lane per ray: frontier start, branch-free step, one-round-trip leaves, quad-finished tails, leaf trigger
    const size_t lds = (static_cast<size_t>(kFindBfRows) * 256u + kQuadStackEntries * 64u + 4u * kTailRays * kTailXferDwords) * sizeof(uint32_t);
    RMCL_LAUNCH_FIND(23, lds)
  } else if (variant == 32) {  // the cooperative descent below the frontier, one bit per final leaf and ray: + the four waves' lists
    const size_t lds = static_cast<size_t>(kFind31LdsDwords) * sizeof(uint32_t);
    RMCL_LAUNCH_FIND(32, lds)
  } else {                     // 24: one lane per ray on the 64-B quantised nodes: frontier start, leaf trigger, 16 LDS rows
    const size_t lds4 = 16u * 256u * sizeof(uint32_t);
    RMCL_LAUNCH_FIND(24, lds4)
  }
#undef RMCL_LAUNCH_FIND
  return hipGetLastError();
}

uint32_t find_moments_blocks(const FindParams& p, int variant) {
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  return (((variant == 2) ? ntiles : (ntiles + 3u) / 4u) + 7u) & ~7u;
}

hipError_t launch_find_moments(const FindParams& p, ModelKind kind, int variant, hipStream_t s) {
  static_assert(kMomRow == kMicpFastMoments && kMomRow == static_cast<uint32_t>(kMom), "one partial-row layout");
  if (p.nposes != 1u || p.wave_clock != nullptr || p.mom_partials == nullptr || p.mom_unc_mask == nullptr || (variant != 23 && variant != 32 && variant != 2))
    return hipErrorInvalidValue;
  dim3 grid(find_moments_blocks(p, variant), 1, 1), block(256, 1, 1);
  if (variant == 2) {
    // quad kind: the ray stacks, then the four waves' staging rows (21 rows of 256 dwords, find_moments_wave)
    const size_t lds = (static_cast<size_t>(kQuadStackEntries) * 64u + 21u * 256u) * sizeof(uint32_t);
    switch (kind) {
      case kModelSpherical: hipLaunchKernelGGL((k_find<kModelSpherical, 2, false, true>), grid, block, lds, s, p); break;
      case kModelO1Dn: hipLaunchKernelGGL((k_find<kModelO1Dn, 2, false, true>), grid, block, lds, s, p); break;
      case kModelPinhole: hipLaunchKernelGGL((k_find<kModelPinhole, 2, false, true>), grid, block, lds, s, p); break;
      case kModelOnDn: hipLaunchKernelGGL((k_find<kModelOnDn, 2, false, true>), grid, block, lds, s, p); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  if (variant == 32) {   // kind 23's epilogue behind the cooperative descent (+ the four waves' lists)
    const size_t lds31 = static_cast<size_t>(kFind31LdsDwords) * sizeof(uint32_t);
    switch (kind) {
      case kModelSpherical: hipLaunchKernelGGL((k_find<kModelSpherical, 32, false, true>), grid, block, lds31, s, p); break;
      case kModelO1Dn: hipLaunchKernelGGL((k_find<kModelO1Dn, 32, false, true>), grid, block, lds31, s, p); break;
      case kModelPinhole: hipLaunchKernelGGL((k_find<kModelPinhole, 32, false, true>), grid, block, lds31, s, p); break;
      case kModelOnDn: hipLaunchKernelGGL((k_find<kModelOnDn, 32, false, true>), grid, block, lds31, s, p); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  const size_t lds = (static_cast<size_t>(kFindBfRows) * 256u + kQuadStackEntries * 64u + 4u * kTailRays * kTailXferDwords) * sizeof(uint32_t);
  switch (kind) {
    case kModelSpherical: hipLaunchKernelGGL((k_find<kModelSpherical, 23, false, true>), grid, block, lds, s, p); break;
    case kModelO1Dn: hipLaunchKernelGGL((k_find<kModelO1Dn, 23, false, true>), grid, block, lds, s, p); break;
    case kModelPinhole: hipLaunchKernelGGL((k_find<kModelPinhole, 23, false, true>), grid, block, lds, s, p); break;
    case kModelOnDn: hipLaunchKernelGGL((k_find<kModelOnDn, 23, false, true>), grid, block, lds, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_tile_planes(const FindParams& p, ModelKind kind, float* planes, hipStream_t s) {
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  if (ntiles == 0u) return hipSuccess;
  const dim3 grid((ntiles + 3u) / 4u), block(256);
  switch (kind) {
    case kModelSpherical: hipLaunchKernelGGL((k_tile_planes<kModelSpherical>), grid, block, 0, s, p, planes); break;
    case kModelO1Dn: hipLaunchKernelGGL((k_tile_planes<kModelO1Dn>), grid, block, 0, s, p, planes); break;
    case kModelPinhole: hipLaunchKernelGGL((k_tile_planes<kModelPinhole>), grid, block, 0, s, p, planes); break;
    default: return hipErrorInvalidValue;   // OnDn: one origin per ray, no pyramid
  }
  return hipGetLastError();
}

hipError_t launch_find_probe(const FindParams& p, int mode, uint32_t* probe_log, hipStream_t s) {
  if (g_lab && g_lab->find_probe) return g_lab->find_probe(p, mode, probe_log, s);
  return kLabMissing;
}

hipError_t launch_cpc_find(const uint32_t* nodes, const uint32_t* tris, const float* dataset_points, uint32_t n,
                           float max_dist, xform Tsm, xform Tms, uint8_t* hits, float* dists, float* points,
                           float* normals, uint32_t* face_ids, bool quad, hipStream_t s, const uint32_t* seed_rec, uint32_t* rec_out,
                           uint32_t n_tris, float bound_d2, const NearGrid* grid, const NearGrid* cells, float skip_d2) {
  if (n == 0) return hipSuccess;
  CpcParams p;
  p.nodes = nodes; p.tris = tris; p.dataset_points = dataset_points; p.n = n; p.max_dist = max_dist;
  p.Tsm = Tsm; p.Tms = Tms; p.hits = hits; p.dists = dists; p.points = points; p.normals = normals; p.face_ids = face_ids;
  p.seed_rec = seed_rec; p.rec_out = rec_out; p.n_tris = n_tris; p.bound_d2 = bound_d2;
  p.near_grid = nullptr; p.from_cells = (cells != nullptr) ? 1u : 0u; p.skip_d2 = skip_d2;
  for (int k = 0; k < 3; ++k) { p.gn[k] = 1u; p.gorg[k] = 0.f; p.ginv[k] = 1.f; p.cn[k] = 1u; p.corg[k] = 0.f; p.cinv[k] = 1.f; }
  if (grid != nullptr) {
    p.near_grid = grid->cells;
    for (int k = 0; k < 3; ++k) { p.gn[k] = grid->n[k]; p.gorg[k] = grid->org[k]; p.ginv[k] = grid->inv[k]; }
  }
  if (cells != nullptr)
    for (int k = 0; k < 3; ++k) { p.cn[k] = cells->n[k]; p.corg[k] = cells->org[k]; p.cinv[k] = cells->inv[k]; }
  if (quad) hipLaunchKernelGGL((k_cpc_find<true>), dim3((n + 63u) / 64u), dim3(256), kQuadStackEntries * 64u * sizeof(uint32_t), s, p);
  else hipLaunchKernelGGL((k_cpc_find<false>), dim3((n + 255u) / 256u), dim3(256), 16u * 256u * sizeof(uint32_t), s, p);
  return hipGetLastError();
}

hipError_t launch_compose_poses(const xform* Tbm_dev, xform Tsb, xform* Tsm_out, xform* Tms_out, uint32_t n,
                                hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_compose_poses, dim3((n + 255u) / 256u), dim3(256), 0, s, Tbm_dev, Tsb, Tsm_out, Tms_out, n);
  return hipGetLastError();
}

uint32_t reduce_num_blocks(uint32_t n, uint32_t nposes) {
  // single reduction: two elements per thread keeps >= 256 workgroups in flight for a 128x1024 scan (the streaming
  // side needs the whole chip; measured 9.5 vs 10.0 us with half as many).  Pose batches already fill the chip with
  // nposes x blocks, and every pose's partials are consumed by ONE wave that fetches them 32 at a time per lane, so
  // fewer, larger blocks win there (64-pose correction 0.82 -> 0.69 ms at 4096 elements per block).
  const uint32_t per_block = (nposes >= 4u) ? 4096u : 512u;
  uint32_t nb = (n + per_block - 1u) / per_block;
  if (nb < 1u) nb = 1u;
  if (nb > 1024u) nb = 1024u;
  return nb;
}

hipError_t launch_reduce_partials(const ReduceParams& p, hipStream_t s) {
  const dim3 grid(p.nblocks, p.nposes), block(256);
  if (p.tail_mode == kTailStats) hipLaunchKernelGGL((k_reduce_partials<kTailStats>), grid, block, 0, s, p);
  else if (p.tail_mode == kTailMicp) hipLaunchKernelGGL((k_reduce_partials<kTailMicp>), grid, block, 0, s, p);
  else if (p.tail_mode == kTailBatchSolve) hipLaunchKernelGGL((k_reduce_partials<kTailBatchSolve>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((k_reduce_partials<kTailNone>), grid, block, 0, s, p);
  return hipGetLastError();
}

// Moment form of the N-sensor loop (k_micp_multi_step's iteration, micp_localization.cpp:915-964): every sensor's statistics
// come from its moments + its undecided correspondences at ITS pre-transform (see k_micp_fast_loop); the merge over the sensors
// and the solve keep the frame-by-frame order of k_micp_multi_step.
__global__ void __launch_bounds__(kFastThreads) k_micp_multi_fast_loop(const MicpMultiFastParams p) {
  constexpr uint32_t kGroups = kFoldGroups;
  __shared__ double s_mom[kMaxMicpSensors][kMom];
  __shared__ double s_part[kGroups][kMom];
  __shared__ double s_rows[kFastThreads][17];
  __shared__ double s_tot[kMaxMicpSensors][16];
  __shared__ double s_R[kMaxMicpSensors][9], s_t[kMaxMicpSensors][3];
  __shared__ MomentScratch s_ws[kFastThreads / 64];   // one per wave: without undecided correspondences wave w sums sensor w
  __shared__ uint32_t s_list[kFastMaxUncertain];
  __shared__ uint32_t s_seg[kMaxMicpSensors + 1];
  __shared__ uint32_t s_wave_cnt[kFastThreads / 64];
  __shared__ xform s_Ts[kMaxMicpSensors];
  __shared__ cstats s_Cs[kMaxMicpSensors];       // sensor statistics in the odom frame
  __shared__ uint32_t s_wn[kMaxMicpSensors];     // ... and their weighted n_meas
  __shared__ xform s_Tone;                       // T_onew_oold of this iteration, for the sensors' lanes
  __shared__ float s_max_rho[kMaxMicpSensors], s_max_tau[kMaxMicpSensors];
  __shared__ uint32_t s_bad;                     // smallest index of a sensor whose pre-transform left its caps
  __shared__ uint32_t s_join_lost;               // a joined stream's signal did not arrive within kDevicePollBound polls
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t ns = p.n_sensors;

  // set-up, sensor by sensor: moments and the index-ordered list segment of its undecided correspondences
  uint32_t total = 0;
  for (uint32_t s = 0; s < ns; ++s) {
    if ((p.join_mask >> s) & 1u) {
      // this sensor's rows and mask words come from another stream: its signal kernel (behind its moment pass) stores the call's
      // sequence number; one lane acquires it, the barrier hands the visibility to the workgroup
      if (tid == 0u) {
        uint32_t polls = 0;
        while (__hip_atomic_load(p.join_flags + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != p.seq && ++polls < kDevicePollBound)
          __builtin_amdgcn_s_sleep(2);
        s_join_lost = (polls >= kDevicePollBound) ? 1u : 0u;
      }
      __syncthreads();
      if (s_join_lost != 0u) {   // the other stream's signal never came: code 2, the host takes the per-iteration form
        if (tid == 0u) {
          MicpMultiFastStatus st;
          st.code = 2u; st.iter = 0u; st.n_uncertain = 0xffffffffu; st.sensor = s;
          for (uint32_t q = 0; q < kMaxMicpSensors; ++q) { st.max_rho[q] = 0.f; st.max_tau[q] = 0.f; }
          publish_status(p.status, st, p.done, p.seq, 0u);
        }
        return;
      }
    }
    fold_moment_partials(p.partials[s], p.nblocks[s], s_part, tid);
    const unsigned long long* mask = p.unc_mask[s];
    const uint32_t nwords = (p.n[s] + 63u) >> 6;
    const uint32_t wpt = (nwords + kFastThreads - 1u) / kFastThreads;
    const uint32_t w0 = min(tid * wpt, nwords), w1 = min(w0 + wpt, nwords);
    uint32_t cnt = 0;
    for (uint32_t w = w0; w < w1; w += 8u) {
      unsigned long long m[8];
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) m[u] = (w + u < w1) ? mask[w + u] : 0ull;
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) cnt += static_cast<uint32_t>(__popcll(m[u]));
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off, 64);
      if (lane >= static_cast<uint32_t>(off)) incl += v;
    }
    if (lane == 63u) s_wave_cnt[wave] = incl;
    __syncthreads();
    if (tid < kMom) {
      double a = s_part[0][tid];
#pragma unroll
      for (uint32_t g = 1; g < kGroups; ++g) a += s_part[g][tid];
      s_mom[s][tid] = a;
    }
    uint32_t wave_base = 0, cnt_s = 0;
#pragma unroll
    for (uint32_t w = 0; w < kFastThreads / 64; ++w) {
      const uint32_t c = s_wave_cnt[w];
      if (w < wave) wave_base += c;
      cnt_s += c;
    }
    if (tid == 0u) s_seg[s] = total;
    if (total + cnt_s > kFastMaxUncertain) {
      if (tid == 0u) {
        MicpMultiFastStatus st;
        st.code = 2u; st.iter = 0u; st.n_uncertain = total + cnt_s; st.sensor = s;
        for (uint32_t q = 0; q < kMaxMicpSensors; ++q) { st.max_rho[q] = 0.f; st.max_tau[q] = 0.f; }
        publish_status(p.status, st, p.done, p.seq, 0u);
      }
      return;
    }
    if (cnt != 0u) {
      uint32_t pos = total + wave_base + incl - cnt;
      for (uint32_t w = w0; w < w1; ++w) {
        unsigned long long bits = mask[w];
        while (bits) {
          const int b = __builtin_ctzll(bits);
          bits &= bits - 1ull;
          s_list[pos++] = (w << 6) + static_cast<uint32_t>(b);
        }
      }
    }
    total += cnt_s;
    __syncthreads();   // s_part / s_wave_cnt are reused by the next sensor
  }
  if (tid == 0u) {
    s_seg[ns] = total;
    s_bad = 0xFFFFFFFFu;
  }
  // Lane s of wave 0 OWNS sensor s (round 3): its frames, their inverses, its pre-transform and its caps live in that lane's
  // registers, and everything that is per sensor -- the pre-transform's rotation matrix, the frame changes of its statistics, its
  // next pre-transform -- runs in the ns lanes at once.  Lane 0 alone only merges and solves.  (Round 2 ran all of it on lane 0:
  // ~2000 dependent instructions per iteration for two sensors, 9.4 us.)
  const bool is_sensor = tid < ns;
  const uint32_t sx = is_sensor ? tid : 0u;
  const xform my_Tsb = p.Tsb[sx], my_Tbo = p.Tbo[sx];
  const xform my_Tsb_inv = xinv(my_Tsb), my_Tbo_inv = xinv(my_Tbo);
  const double my_weight = p.weight[sx];
  const float my_rho_cap = p.rho_cap[sx], my_tau_cap = p.tau_cap[sx];
  xform my_Ts = xidentity();
  float my_max_rho = 0.f, my_max_tau = 0.f;
  if (is_sensor) { s_Ts[tid] = my_Ts; s_max_rho[tid] = 0.f; s_max_tau[tid] = 0.f; }
  // loop state of thread 0
  xform T_onew_oold = xidentity();
  cstats merged = cs_identity(), merged_w = cs_identity();
  __syncthreads();
  for (uint32_t it = 0; it < p.n_iter; ++it) {
    if (is_sensor) {
      const xform T = my_Ts;
      const float rho = 2.0f * sqrtf((T.R.x * T.R.x + T.R.y * T.R.y) + T.R.z * T.R.z);
      const float tau = sqrtf(dot_plain(T.t, T.t));
      my_max_rho = fmaxf(my_max_rho, rho);
      my_max_tau = fmaxf(my_max_tau, tau);
      s_max_rho[tid] = my_max_rho; s_max_tau[tid] = my_max_tau;
      if (!(rho <= my_rho_cap) || !(tau <= my_tau_cap)) atomicMin(&s_bad, tid);   // the FIRST sensor outside its caps is reported
      const double x = T.R.x, y = T.R.y, z = T.R.z, w = T.R.w;
      const double ww = w * w, uu = (x * x + y * y) + z * z;
      double* R = s_R[tid];
      R[0] = (ww - uu) + 2.0 * x * x; R[1] = 2.0 * (x * y - w * z);   R[2] = 2.0 * (x * z + w * y);
      R[3] = 2.0 * (x * y + w * z);   R[4] = (ww - uu) + 2.0 * y * y; R[5] = 2.0 * (y * z - w * x);
      R[6] = 2.0 * (x * z - w * y);   R[7] = 2.0 * (y * z + w * x);   R[8] = (ww - uu) + 2.0 * z * z;
      s_t[tid][0] = T.t.x; s_t[tid][1] = T.t.y; s_t[tid][2] = T.t.z;
    }
    __syncthreads();
    if (s_bad != 0xFFFFFFFFu) {
      if (tid == 0u) {
        MicpMultiFastStatus st;
        st.code = 1u; st.iter = it; st.n_uncertain = total; st.sensor = s_bad;
        for (uint32_t q = 0; q < kMaxMicpSensors; ++q) { st.max_rho[q] = (q < ns) ? s_max_rho[q] : 0.f; st.max_tau[q] = (q < ns) ? s_max_tau[q] : 0.f; }
        publish_status(p.status, st, p.done, p.seq, 0u);
      }
      return;
    }
    if (total == 0u) {
      // nothing to re-evaluate (the usual tracking case): the sensors' 16 sums come from their moments alone, one sensor per wave
      for (uint32_t s = wave; s < ns; s += kFastThreads / 64u) micp_moment_sums_wave(lane, s_mom[s], s_R[s], s_t[s], &s_ws[wave], s_tot[s]);
      __syncthreads();
    } else
    for (uint32_t s = 0; s < ns; ++s) {
      const uint32_t seg0 = s_seg[s], seg1 = s_seg[s + 1];
      const uint32_t nrows = min(seg1 - seg0, kFastThreads);
      if (tid < nrows) {
        const xform Tpre = s_Ts[s];
        const float max_dist = p.max_dist[s];
        const float* dpts = p.dataset_points[s];
        const float* mpts = p.model_points[s];
        const float* mnrm = p.model_normals[s];
        double acc[kAcc];
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
        for (uint32_t e = seg0 + tid; e < seg1; e += kFastThreads) {
          const uint32_t i = s_list[e];
          const float* dp = dpts + 3 * static_cast<size_t>(i);
          const float* mp = mpts + 3 * static_cast<size_t>(i);
          const float* mn = mnrm + 3 * static_cast<size_t>(i);
          const f3 Di = xapply(Tpre, mk3(dp[0], dp[1], dp[2]));
          const f3 Ii = mk3(mp[0], mp[1], mp[2]);
          const f3 Ni = mk3(mn[0], mn[1], mn[2]);
          const float spd = dot_plain(sub3(Ii, Di), Ni);
          if (fabsf(spd) < max_dist) {
            const f3 Mi = add3(Di, scale3(Ni, spd));
            const double d[3] = {Di.x, Di.y, Di.z}, m[3] = {Mi.x, Mi.y, Mi.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) { acc[k] += d[k]; acc[3 + k] += m[k]; }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int c = 0; c < 3; ++c) acc[6 + 3 * r + c] += m[r] * d[c];
            acc[15] += 1.0;
          }
        }
#pragma unroll
        for (int k = 0; k < kAcc; ++k) s_rows[tid][k] = acc[k];
      }
      __syncthreads();
      if (wave == 0u) {
        micp_moment_sums_wave(lane, s_mom[s], s_R[s], s_t[s], &s_ws[0], s_tot[s]);
        if (lane < 16u && nrows != 0u) {
          double v = s_tot[s][lane];
          for (uint32_t r = 0; r < nrows; ++r) v += s_rows[r][lane];
          s_tot[s][lane] = v;
        }
      }
      __syncthreads();   // s_rows / s_ws are reused by the next sensor
    }
    // k_micp_multi_step's merge and solve, frame by frame: the frame changes sensor-parallel, merge + solve on lane 0, the next
    // pre-transforms sensor-parallel again.  All in wave 0, whose LDS operations complete in program order.
    if (is_sensor) {
      double tot[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) tot[k] = s_tot[tid][k];
      const cstats stats_s = cstats_from_sums(tot);
      s_Cs[tid] = cs_transform(my_Tbo, cs_transform(my_Tsb, stats_s));
      s_wn[tid] = static_cast<uint32_t>(static_cast<double>(stats_s.n_meas) * my_weight);   // n_meas *= merge_weight_multiplier (:934)
    }
    if (wave == 0u) __builtin_amdgcn_wave_barrier();
    if (tid == 0u) {
      merged = cs_identity();
      merged_w = cs_identity();
      for (uint32_t s = 0; s < ns; ++s) {
        const cstats Cs_o = s_Cs[s];
        cstats Cs_w = Cs_o;
        Cs_w.n_meas = s_wn[s];
        merged = cs_merge(merged, Cs_o);
        merged_w = cs_merge(merged_w, Cs_w);
      }
      T_onew_oold = xmul(T_onew_oold, umeyama_fast(merged_w));
      s_Tone = T_onew_oold;
    }
    if (wave == 0u) __builtin_amdgcn_wave_barrier();
    if (is_sensor) {
      const xform T1 = s_Tone;
      const xform T_bnew_bold = xmul(xmul(my_Tbo_inv, T1), my_Tbo);
      my_Ts = xmul(xmul(my_Tsb_inv, T_bnew_bold), my_Tsb);
      s_Ts[tid] = my_Ts;
    }
  }
  __syncthreads();
  if (tid == 0u) {
    MicpMultiState* out = p.state_out;
    out->T_onew_oold = T_onew_oold;
    out->merged_o = merged;
    out->merged_weighted_o = merged_w;
    for (uint32_t s = 0; s < ns; ++s) out->T_snew_sold[s] = s_Ts[s];
    MicpMultiFastStatus st;
    st.code = 0u; st.iter = p.n_iter; st.n_uncertain = total; st.sensor = 0u;
    for (uint32_t q = 0; q < kMaxMicpSensors; ++q) { st.max_rho[q] = (q < ns) ? s_max_rho[q] : 0.f; st.max_tau[q] = (q < ns) ? s_max_tau[q] : 0.f; }
    // the host reads T_onew_oold and the merged statistics of this block (rmclhip_micp_correct_once)
    publish_status(p.status, st, p.done, p.seq, xor_words(T_onew_oold) ^ xor_words(merged) ^ xor_words(merged_w));
  }
}

hipError_t launch_reduce_finalize(const double* partials, uint32_t nblocks, uint32_t nposes, cstats* out,
                                  unsigned long long* done, uint32_t seq, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_finalize, dim3(nposes), dim3(64), 0, s, partials, nblocks, out, (nposes == 1u) ? done : nullptr, seq);
  return hipGetLastError();
}

hipError_t launch_micp_loop(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                            const float* model_normals, const uint8_t* model_mask, uint32_t n, uint32_t n_iter,
                            const MicpCall* call, double* partials, uint32_t* barrier, MicpState* state,
                            uint32_t nblocks, bool one_xcd, hipStream_t s) {
  MicpLoopParams p{dataset_points, dataset_mask, model_points, model_normals, model_mask, n, n_iter, call, partials,
                   barrier, state};
  if (one_xcd) hipLaunchKernelGGL((k_micp_loop<true>), dim3(nblocks * 8u), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((k_micp_loop<false>), dim3(nblocks), dim3(512), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_micp_iter(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                            const float* model_normals, const uint8_t* model_mask, uint32_t n, uint32_t nblocks,
                            const MicpCall* call, const double* partials_prev, double* partials_out,
                            const MicpState* state_in, MicpState* state_out, bool first, hipStream_t s) {
  MicpIterParams p{dataset_points, dataset_mask, model_points, model_normals, model_mask, n, nblocks, call,
                   partials_prev, partials_out, state_in, state_out, first ? 1u : 0u};
  hipLaunchKernelGGL(k_micp_iter, dim3(nblocks), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_micp_moments(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                               const float* model_normals, const uint8_t* model_mask, uint32_t n, const MicpCall* call,
                               double* partials, unsigned long long* unc_mask, hipStream_t s, const MicpCallLite* call_by_value) {
  MicpFastParams p{dataset_points, dataset_mask, model_points, model_normals, model_mask, n, micp_fast_blocks(n), call,
                   partials, unc_mask, 0u, nullptr, nullptr, nullptr, {}};
  if (call_by_value) { p.call = nullptr; p.cv = *call_by_value; }
  hipLaunchKernelGGL(k_micp_moments, dim3(p.nblocks), dim3(256), 0, s, p);
  return hipGetLastError();
}

namespace {
__global__ void k_signal_flag(uint32_t* flag, uint32_t seq) {
  if (threadIdx.x == 0u && blockIdx.x == 0u) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace
namespace {
// completion tag of a launch chain that returns nothing to the host but its own end: {seq, sum 0} to pinned memory, behind the chain
// on its stream (the kernel boundary before this launch is what makes the chain's results visible)
__global__ void k_host_tag(unsigned long long* done, uint32_t seq) {
  if (threadIdx.x == 0u && blockIdx.x == 0u) publish_tag(done, seq, 0u);
}
}  // namespace
hipError_t launch_host_tag(unsigned long long* done, uint32_t seq, hipStream_t s) {
  hipLaunchKernelGGL(k_host_tag, dim3(1), dim3(64), 0, s, done, seq);
  return hipGetLastError();
}
hipError_t launch_signal_flag(uint32_t* flag, uint32_t seq, hipStream_t s) {
  hipLaunchKernelGGL(k_signal_flag, dim3(1), dim3(64), 0, s, flag, seq);
  return hipGetLastError();
}

hipError_t launch_micp_multi_fast_loop(const MicpMultiFastParams& p, hipStream_t s) {
  hipLaunchKernelGGL(k_micp_multi_fast_loop, dim3(1), dim3(kFastThreads), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_micp_fast(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                            const float* model_normals, const uint8_t* model_mask, uint32_t n, const MicpCall* call,
                            double* partials, unsigned long long* unc_mask, uint32_t n_iter, MicpState* state_out,
                            MicpFastStatus* status, unsigned long long* done, hipStream_t s, const MicpCallLite* call_by_value) {
  MicpFastParams p{dataset_points, dataset_mask, model_points, model_normals, model_mask, n, micp_fast_blocks(n), call,
                   partials, unc_mask, n_iter, state_out, status, done, {}};
  if (call_by_value) { p.call = nullptr; p.cv = *call_by_value; }
  hipLaunchKernelGGL(k_micp_moments, dim3(p.nblocks), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_micp_fast_loop, dim3(1), dim3(kFastThreads), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_micp_fast_loop_tiled(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                                       const float* model_normals, const uint8_t* model_mask, uint32_t n, uint32_t nblocks,
                                       const double* partials, const unsigned long long* unc_mask, uint32_t W, uint32_t tiles_x,
                                       uint32_t tile_w_log2, uint32_t words_per_block, uint32_t n_iter, MicpState* state_out,
                                       MicpFastStatus* status, unsigned long long* done, hipStream_t s, const MicpCallLite& call_by_value,
                                       double* fold_rows, uint32_t* fold_flags) {
  MicpFastParams p{dataset_points, dataset_mask, model_points, model_normals, model_mask, n, nblocks, nullptr,
                   const_cast<double*>(partials), const_cast<unsigned long long*>(unc_mask), n_iter, state_out, status, done, call_by_value,
                   1u, W, tiles_x, tile_w_log2, words_per_block * nblocks, fold_rows, fold_flags};
  const uint32_t nfold = (fold_rows != nullptr && fold_flags != nullptr && nblocks >= 256u) ? kMicpFoldBlocks : 1u;
  hipLaunchKernelGGL(k_micp_fast_loop, dim3(nfold), dim3(kFastThreads), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_micp_publish_tiled(const float* dataset_points, const float* model_points, const float* model_normals, uint32_t n,
                                     uint32_t nblocks, const double* partials, const unsigned long long* unc_mask, uint32_t W,
                                     uint32_t tiles_x, uint32_t tile_w_log2, uint32_t words_per_block, MicpHostBlock* host_block,
                                     unsigned long long* done, uint32_t seq, double* fold_rows, uint32_t* fold_flags, hipStream_t s) {
  MicpCallLite cv{};
  cv.seq = seq;
  MicpFastParams p{dataset_points, nullptr, model_points, model_normals, nullptr, n, nblocks, nullptr,
                   const_cast<double*>(partials), const_cast<unsigned long long*>(unc_mask), 0u, nullptr, nullptr, done, cv,
                   1u, W, tiles_x, tile_w_log2, words_per_block * nblocks, fold_rows, fold_flags, host_block};
  const uint32_t nfold = (fold_rows != nullptr && fold_flags != nullptr && nblocks >= 256u) ? kMicpFoldBlocks : 1u;
  hipLaunchKernelGGL(k_micp_publish, dim3(nfold), dim3(kFastThreads), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_micp_moments_publish(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                                       const float* model_normals, const uint8_t* model_mask, uint32_t n, double* partials,
                                       unsigned long long* unc_mask, const MicpCallLite& cv, MicpHostBlock* host_block,
                                       unsigned long long* done, hipStream_t s) {
  MicpFastParams p{dataset_points, dataset_mask, model_points, model_normals, model_mask, n, micp_fast_blocks(n), nullptr,
                   partials, unc_mask, 0u, nullptr, nullptr, done, cv};
  p.host_block = host_block;
  hipLaunchKernelGGL(k_micp_moments, dim3(p.nblocks), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_micp_publish, dim3(1), dim3(kFastThreads), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_micp_multi_init(const MicpMultiCall* call, MicpMultiState* state, hipStream_t s) {
  hipLaunchKernelGGL(k_micp_multi_init, dim3(1), dim3(64), 0, s, call, state);
  return hipGetLastError();
}

hipError_t launch_micp_multi_step(const MicpMultiCall* call, MicpMultiState* state, hipStream_t s) {
  hipLaunchKernelGGL(k_micp_multi_step, dim3(1), dim3(64), 0, s, call, state);
  return hipGetLastError();
}

hipError_t launch_micp_init(MicpState* state, uint32_t* barrier, hipStream_t s) {
  hipLaunchKernelGGL(k_micp_init, dim3(1), dim3(64), 0, s, state, barrier);
  return hipGetLastError();
}

hipError_t launch_micp_close(const double* partials, uint32_t nblocks, const MicpCall* call, const MicpState* state,
                             MicpState* state_out, unsigned long long* done, hipStream_t s) {
  hipLaunchKernelGGL(k_micp_close, dim3(1), dim3(64), 0, s, partials, nblocks, call, state, state_out, done);
  return hipGetLastError();
}

hipError_t launch_micp_step(const double* partials, uint32_t nblocks, xform Tsb, xform Tbo, const MicpCall* call,
                            const MicpState* state, MicpState* state_out, hipStream_t s) {
  hipLaunchKernelGGL(k_micp_step, dim3(1), dim3(64), 0, s, partials, nblocks, Tsb, Tbo, call, state, state_out);
  return hipGetLastError();
}

hipError_t launch_batch_solve(const double* partials, uint32_t nblocks, uint32_t nposes, xform Tsb,
                              xform* Tdelta_out, cstats* stats_out, hipStream_t s) {
  hipLaunchKernelGGL(k_batch_solve, dim3(nposes), dim3(64), 0, s, partials, nblocks, Tsb, Tdelta_out, stats_out);
  return hipGetLastError();
}

hipError_t launch_dataset_from_ranges(const float* ranges, const float* model_tab, uint32_t kind, uint32_t W,
                                      uint32_t H, f3 orig, const float* pin_fc, float rmin, float rmax, float* points,
                                      uint8_t* mask, uint32_t* n_valid, hipStream_t s) {
  const uint32_t n = W * H;
  hipLaunchKernelGGL(k_dataset_from_ranges, dim3((n + 255u) / 256u), dim3(256), 0, s, ranges, model_tab, kind, W, H,
                     orig, pin_fc[0], pin_fc[1], pin_fc[2], pin_fc[3], rmin, rmax, points, mask, n_valid);
  return hipGetLastError();
}

hipError_t launch_pf_update(const PfParams& p, int variant, hipStream_t s) {
  const uint32_t nblocks = (p.n_particles + p.particles_per_block - 1u) / p.particles_per_block;
  const size_t tail_fixed = sizeof(xform) * p.particles_per_block;
  const size_t tail = tail_fixed + sizeof(float) * static_cast<size_t>(p.particles_per_block) * p.n_beams;
  const int trav = variant & 3;          // traversal of the round kernels (lab)
  const bool cpc = (variant & 8) != 0;   // correspondence_type 1
  const int refill = (variant >> 4) & 7;  // 0 = rounds of one ray per lane (lab); 1..4 = persistent lanes
  const bool legacy = ((variant >> 8) & 1) != 0;   // bit 8: the round-2 kernel (k_pf_update_persist, lab), A/B
  const bool leaf2 = ((variant >> 10) & 1) == 0;   // bit 10 set: p.qnodes is the MAP's tree (leaves <= 4) -> loop over the leaf
  if (cpc) {
    // closest-point correspondences (evaluate_cpc): rounds of one beam per lane on the full nodes
    const size_t lds = 16u * 256u * sizeof(uint32_t) + tail;
    if (lds > 160u * 1024u - 64u) return hipErrorInvalidValue;
    if (lds > 65536u) {
      const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pf_update<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                160 * 1024 - 64);
      if (ae != hipSuccess) return ae;
    }
    hipLaunchKernelGGL((k_pf_update<64, 3>), dim3(nblocks), dim3(256), lds, s, p);
    return hipGetLastError();
  }
  if (trav == 0 && refill != 0 && !legacy && ((variant >> 7) & 1) == 0 && p.qnodes != nullptr) {
    // bit 11: the children of a node in the ray's SLOT order (round 5) instead of sorted by entry distance; bit 12: order-independent
    // likelihood accumulation (round 5: no per-beam storage; 19 stack rows so that the accumulators fit beside them at 7 workgroups per CU)
    const bool slot_order = ((variant >> 11) & 1) != 0, accum = ((variant >> 12) & 1) != 0 && p.gpow != nullptr;
    constexpr int kAccRows = kPfRows - 1;
    const uint32_t PB = p.particles_per_block;
    const size_t acc_off = (static_cast<size_t>(kAccRows) * 256u + 8u * PB + PB + 1u) * sizeof(uint32_t);   // + 1: the kernel's alignment skip
    const size_t lds = accum ? acc_off + static_cast<size_t>(PB) * (sizeof(double) + 2u * sizeof(uint32_t) + static_cast<size_t>(kAccWords) * 8u)
                             : static_cast<size_t>(kPfRows) * 256u * sizeof(uint32_t) + (p.evals != nullptr ? tail_fixed : tail) + sizeof(uint32_t) * PB;
    if (lds > 160u * 1024u - 64u) return hipErrorInvalidValue;   // more beams per particle than one workgroup's LDS holds
#define RMCL_PF_V3(ROWS, L2, SO, AC)                                                                                           \
    {                                                                                                                          \
      if (lds > 65536u) {   /* per launch: the attribute belongs to the function ON THE CURRENT DEVICE (sharded filters) */   \
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pf_update_v3<ROWS, L2, SO, AC>),            \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);                \
        if (ae != hipSuccess) return ae;                                                                                       \
      }                                                                                                                        \
      hipLaunchKernelGGL((k_pf_update_v3<ROWS, L2, SO, AC>), dim3(nblocks), dim3(256), lds, s, p);                              \
      return hipGetLastError();                                                                                                \
    }
    if (accum) {
      if (leaf2 && slot_order) RMCL_PF_V3(kAccRows, true, true, true)
      if (leaf2) RMCL_PF_V3(kAccRows, true, false, true)
      if (slot_order) RMCL_PF_V3(kAccRows, false, true, true)
      RMCL_PF_V3(kAccRows, false, false, true)
    }
    if (leaf2 && slot_order) RMCL_PF_V3(kPfRows, true, true, false)
    if (leaf2) RMCL_PF_V3(kPfRows, true, false, false)
    if (slot_order) RMCL_PF_V3(kPfRows, false, true, false)
    RMCL_PF_V3(kPfRows, false, false, false)
#undef RMCL_PF_V3
  }
  if (g_lab && g_lab->pf_update) return g_lab->pf_update(p, variant, s);
  return kLabMissing;
}

hipError_t launch_pf_motion(const uint32_t* nodes, const uint32_t* tris, xform* poses, void* attrs, uint32_t n,
                            xform T_bnew_bold, double forget_rate, uint32_t max_n_meas, bool collision, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const dim3 grid((n + 255u) / 256u), block(256);
  if (collision)
    hipLaunchKernelGGL((k_pf_motion<true>), grid, block, 16u * 256u * sizeof(uint32_t), s, nodes, tris, poses,
                       reinterpret_cast<pattrs*>(attrs), n, T_bnew_bold, forget_rate, max_n_meas);
  else
    hipLaunchKernelGGL((k_pf_motion<false>), grid, block, 0, s, nodes, tris, poses, reinterpret_cast<pattrs*>(attrs), n,
                       T_bnew_bold, forget_rate, max_n_meas);
  return hipGetLastError();
}

hipError_t launch_pf_extract_weights(const void* attrs, uint32_t n, float* weights, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pf_extract_weights, dim3((n + 255u) / 256u), dim3(256), 0, s,
                     reinterpret_cast<const pattrs*>(attrs), n, weights);
  return hipGetLastError();
}

hipError_t launch_gladiator_resample(const xform* poses, const void* attrs, uint32_t n, xform* poses_new, void* attrs_new,
                                     uint32_t first, uint32_t count, const float* cfg8, uint32_t trans_dist_metric,
                                     uint64_t seed, uint32_t step, hipStream_t s) {
  if (count == 0) return hipSuccess;
  GladiatorConfig c{cfg8[0], cfg8[1], cfg8[2], cfg8[3], cfg8[4], cfg8[5], cfg8[6], cfg8[7], trans_dist_metric};
  hipLaunchKernelGGL(k_gladiator_resample, dim3((count + 255u) / 256u), dim3(256), 0, s, poses,
                     reinterpret_cast<const pattrs*>(attrs), n, poses_new, reinterpret_cast<pattrs*>(attrs_new), first,
                     count, c, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), step);
  return hipGetLastError();
}

// residual resampling, step 1: {sum, max} in double + the expected number of copies per draw (x n) -> *stats (device)
hipError_t launch_residual_prepare(const void* attrs, uint32_t n, uint32_t n_new, double* psum, float* pmax, void* stats, hipStream_t s) {
  uint32_t nblocks = (n + 1023u) / 1024u;
  if (nblocks < 1u) nblocks = 1u;
  if (nblocks > 256u) nblocks = 256u;
  hipLaunchKernelGGL(k_likelihood_stats_partial, dim3(nblocks), dim3(256), 0, s, reinterpret_cast<const float*>(attrs), 9u, n, psum, pmax);
  hipLaunchKernelGGL(k_residual_stats_final, dim3(1), dim3(64), 0, s, psum, pmax, nblocks, reinterpret_cast<ResidualStats*>(stats));
  hipLaunchKernelGGL(k_residual_expect, dim3(nblocks), dim3(256), 0, s, reinterpret_cast<const pattrs*>(attrs), n, n_new,
                     reinterpret_cast<ResidualStats*>(stats));
  return hipGetLastError();
}

// step 2: the particle and the copy count of draws 0 .. n_draws-1 and the inclusive prefix sums of the counts
hipError_t launch_residual_draws(const void* attrs, uint32_t n, uint32_t n_new, const void* stats, uint32_t n_draws, uint64_t seed,
                                 uint32_t step, uint32_t* draw_idx, uint32_t* draw_cnt, unsigned long long* incl,
                                 unsigned long long* block_tot, hipStream_t s) {
  if (n_draws == 0) return hipSuccess;
  hipLaunchKernelGGL(k_residual_counts, dim3((n_draws + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<const pattrs*>(attrs), n, n_new,
                     reinterpret_cast<const ResidualStats*>(stats), n_draws, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32),
                     step, draw_idx, draw_cnt);
  const uint32_t nb = (n_draws + 1023u) / 1024u;
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(256), 0, s, draw_cnt, n_draws, incl, block_tot);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(256), 0, s, block_tot, nb);
  hipLaunchKernelGGL(k_scan_add, dim3((n_draws + 255u) / 256u), dim3(256), 0, s, incl, n_draws, block_tot);
  return hipGetLastError();
}

// step 3: slots first .. first+count-1 of the new cloud -> poses_new / attrs_new [0 .. count)
hipError_t launch_residual_fill(const xform* poses, const void* attrs, const uint32_t* draw_idx, const unsigned long long* incl,
                                uint32_t n_draws, xform* poses_new, void* attrs_new, uint32_t n_new, uint32_t first, uint32_t count,
                                const float* cfg8, void* stats, uint64_t seed, uint32_t step, hipStream_t s) {
  if (count == 0) return hipSuccess;
  GladiatorConfig c{cfg8[0], cfg8[1], cfg8[2], cfg8[3], cfg8[4], cfg8[5], cfg8[6], cfg8[7], 1u};
  hipLaunchKernelGGL(k_residual_fill, dim3((count + 255u) / 256u), dim3(256), 0, s, poses, reinterpret_cast<const pattrs*>(attrs), draw_idx,
                     incl, n_draws, poses_new, reinterpret_cast<pattrs*>(attrs_new), n_new, first, count, c,
                     reinterpret_cast<ResidualStats*>(stats), static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), step);
  return hipGetLastError();
}

hipError_t launch_build_cnodes16(const uint32_t* cnodes, uint32_t n_nodes, uint32_t* cnodes16, hipStream_t s) {
  if (n_nodes == 0u) return hipSuccess;
  const uint32_t nb = (n_nodes * 16u + 255u) / 256u;
  hipLaunchKernelGGL(k_build_cnodes16, dim3(nb), dim3(256), 0, s, reinterpret_cast<const uint4*>(cnodes), n_nodes, reinterpret_cast<uint4*>(cnodes16));
  return hipGetLastError();
}

hipError_t launch_likelihood_stats(const void* attrs, uint32_t n, double* psum, float* pmax, float* out2, hipStream_t s) {
  uint32_t nblocks = (n + 1023u) / 1024u;
  if (nblocks < 1u) nblocks = 1u;
  if (nblocks > 256u) nblocks = 256u;
  static_assert(sizeof(pattrs) == 36 && offsetof(pattrs, likelihood) == 0, "likelihood.mean is the first float of a 9-float record");
  hipLaunchKernelGGL(k_likelihood_stats_partial, dim3(nblocks), dim3(256), 0, s, reinterpret_cast<const float*>(attrs), 9u, n, psum, pmax);
  hipLaunchKernelGGL(k_likelihood_stats_final, dim3(1), dim3(64), 0, s, psum, pmax, nblocks, out2);
  return hipGetLastError();
}

hipError_t launch_likelihood_stats_dense(const float* weights, uint32_t n, double* psum, float* pmax, float* out2, hipStream_t s) {
  uint32_t nblocks = (n + 1023u) / 1024u;   // the rule of launch_likelihood_stats: same blocks, same order, same bits
  if (nblocks < 1u) nblocks = 1u;
  if (nblocks > 256u) nblocks = 256u;
  hipLaunchKernelGGL(k_likelihood_stats_partial, dim3(nblocks), dim3(256), 0, s, weights, 1u, n, psum, pmax);
  hipLaunchKernelGGL(k_likelihood_stats_final, dim3(1), dim3(64), 0, s, psum, pmax, nblocks, out2);
  return hipGetLastError();
}

hipError_t launch_pointcloud2_unpack(const uint8_t* data, uint32_t point_step, uint32_t row_step, uint32_t off_x,
                                     uint32_t off_y, uint32_t off_z, bool is_f64, uint32_t h_skip, uint32_t h_inc,
                                     uint32_t w_skip, uint32_t w_inc, uint32_t out_w, uint32_t out_h, float range_min,
                                     float range_max, float* dirs, float* points, uint8_t* mask, uint32_t* n_valid,
                                     hipStream_t s) {
  const uint32_t n = out_w * out_h;
  if (n == 0) return hipSuccess;
  Pc2Params p{data, point_step, row_step, off_x, off_y, off_z, is_f64 ? 1u : 0u, h_skip, h_inc, w_skip, w_inc,
              out_w, out_h, range_min, range_max, dirs, points, mask, n_valid};
  hipLaunchKernelGGL(k_pointcloud2_unpack, dim3((n + 255u) / 256u), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_pose_moments(const xform* poses, const void* attrs, uint32_t n, int pass, double L_sum, xform Tbm,
                               double* partials, double* out32, hipStream_t s) {
  uint32_t nblocks = (n + 1023u) / 1024u;
  if (nblocks < 1u) nblocks = 1u;
  if (nblocks > 256u) nblocks = 256u;
  hipLaunchKernelGGL(k_pose_moments, dim3(nblocks), dim3(256), 0, s, poses, reinterpret_cast<const pattrs*>(attrs), n, pass, L_sum, Tbm,
                     partials);
  hipLaunchKernelGGL(k_pose_moments_final, dim3(1), dim3(256), 0, s, partials, nblocks, out32);
  return hipGetLastError();
}

// in-process stand-in for ncclAllReduce on doubles (the loopback communicator of the tests, capi_multi.cpp): every "rank" runs this on its own
// stream over the send buffers of all of them, in rank order (deterministic)
struct LoopbackPtrs { const double* p[64]; };
__global__ void k_loopback_allreduce(LoopbackPtrs send, uint32_t world, double* __restrict__ recv, uint32_t count, uint32_t is_max) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double a = send.p[0][i];
  for (uint32_t r = 1; r < world; ++r) { const double b = send.p[r][i]; a = is_max ? fmax(a, b) : a + b; }
  recv[i] = a;
}
hipError_t launch_loopback_allreduce(const double* const* send, uint32_t world, double* recv, uint32_t count, bool is_max, hipStream_t s) {
  if (count == 0 || world == 0 || world > 64) return world > 64 ? hipErrorInvalidValue : hipSuccess;
  LoopbackPtrs lp;
  for (uint32_t r = 0; r < 64; ++r) lp.p[r] = send[r < world ? r : 0];
  hipLaunchKernelGGL(k_loopback_allreduce, dim3((count + 63u) / 64u), dim3(64), 0, s, lp, world, recv, count, is_max ? 1u : 0u);
  return hipGetLastError();
}

hipError_t launch_compact_shards(const float* padded, float* dense, uint32_t n_total, uint32_t world, uint32_t cap, hipStream_t s) {
  if (n_total == 0) return hipSuccess;
  hipLaunchKernelGGL(k_compact_shards, dim3((n_total + 255u) / 256u), dim3(256), 0, s, padded, dense, n_total, world, cap);
  return hipGetLastError();
}

hipError_t launch_compact_records(const void* padded, void* dense, uint32_t n_total, uint32_t world, uint32_t cap, uint32_t record_bytes,
                                  hipStream_t s) {
  if (n_total == 0) return hipSuccess;
  if (record_bytes % 4u != 0u) return hipErrorInvalidValue;
  const uint32_t wpr = record_bytes / 4u;
  const size_t threads = static_cast<size_t>(n_total) * wpr;
  hipLaunchKernelGGL(k_compact_records, dim3(static_cast<uint32_t>((threads + 255u) / 256u)), dim3(256), 0, s, static_cast<const uint32_t*>(padded),
                     static_cast<uint32_t*>(dense), n_total, world, cap, wpr);
  return hipGetLastError();
}

}  // namespace rmclhip

extern "C" void rmclhip_internal_register_lab(const rmclhip::LabHooks* hooks) { rmclhip::g_lab = hooks; }
