// kernels_lab.hip -- EXPERIMENTS of the find / particle-filter kernels (librmclhip_lab.so; never on the product path):
//   * every traversal kind of k_find the automatic rule cannot select (1, 5..14, 16, 17, 20: measured and rejected, DESIGN.md 4 /
//     profiles/r02_find_variants_ab.txt) and the clocked instantiations of all kinds for tools/wave_timeline.py,
//   * k_find_probe (tools/probe_find.py),
//   * the round kernels k_pf_update<*, 0/1/2> and the round-2 persistent kernel k_pf_update_persist (A/B of k_pf_update_v3).
// Loading the library registers its launchers with librmclhip.so (lab_hooks.h); tests marked `lab` and tools/ use it.
#include "find_kernel.hip.h"
#include "lab_hooks.h"
#include "pf_common.hip.h"

namespace rmclhip {

namespace {

// ---------------------------------------------------------------------------------------------
// DIAGNOSTIC kernel (tools/probe_find.py; never on the product path): the per-lane while-while traversal of k_find<spherical>
// with s_memtime stamps around every node step and every leaf step of every wave, so that the cost of a step can be
// split into "loads issued -> data arrived" and "arithmetic + stack traffic" per tree depth.
// Log entry (2 dwords): {cycles since wave start, kind | active lanes << 8 | uniform << 16 | step << 20}; kinds: 1 node step
// begins, 2 its node data arrived, 3 it ends, 4 leaf step begins, 5 its records arrived, 6 it ends, 7 traversal done, 8 stores
// issued.  probe_log[wave][0] = {number of entries, XCC id}.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kProbeEntries = 255;

__device__ __forceinline__ uint32_t probe_clock(bool drain) {
  uint64_t t;
  if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  else asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return static_cast<uint32_t>(t);
}

template <bool kLeafBatch, int kTop>
__global__ void __launch_bounds__(256) k_find_probe(const FindParams p, uint32_t* __restrict__ probe_log) {
  extern __shared__ uint32_t lds_dyn[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if constexpr (kTop > 0) {
    uint4* dst = reinterpret_cast<uint4*>(lds_dyn + 16u * 256u);
    const uint4* src = reinterpret_cast<const uint4*>(p.nodes);
    const uint32_t n16 = min(static_cast<uint32_t>(kTop), p.n_nodes) * 8u;
    for (uint32_t i = threadIdx.x; i < n16; i += 256u) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t chunk = gridDim.x >> 3;
  const uint32_t vb = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
  const uint32_t tile = vb * 4u + wave;
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  if (tile >= ntiles) return;
  const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const uint32_t twl = p.tile_w_log2;
  const uint32_t lx = lane & ((1u << twl) - 1u), ly = lane >> twl;
  const uint32_t vid = (ty << (6u - twl)) + ly, hid = (tx << twl) + lx;
  const bool valid = (vid < p.H) && (hid < p.W);
  const uint32_t cv = valid ? vid : 0u, ch = valid ? hid : 0u;
  const uint32_t loc = cv * p.W + ch;
  const xform Tsm = p.Tsm, Tms = p.Tms;
  uint32_t* log = probe_log + static_cast<size_t>(tile) * (2u * (kProbeEntries + 1u));
  uint32_t nlog = 0;
  const uint32_t t_begin = probe_clock(false);
#define RMCL_PROBE(KIND, DRAIN, ACTIVE_MASK, UNIFORM, STEP)                                                              \
  {                                                                                                                      \
    const uint32_t tc_ = probe_clock(DRAIN) - t_begin;                                                                    \
    if (nlog < kProbeEntries && lane == 0u) {                                                                            \
      log[2u * (nlog + 1u)] = tc_;                                                                                       \
      log[2u * (nlog + 1u) + 1u] = (KIND) | (static_cast<uint32_t>(__popcll(ACTIVE_MASK)) << 8) | ((UNIFORM) ? 0x10000u : 0u) | ((STEP) << 20); \
    }                                                                                                                    \
    ++nlog;                                                                                                              \
  }
  const float cp = p.model_tab[cv], sp_ = p.model_tab[p.H + cv];
  const float ct = p.model_tab[2u * p.H + ch], st = p.model_tab[2u * p.H + p.W + ch];
  const f3 dir_s = mk3(cp * ct, cp * st, sp_);
  const f3 O = Tsm.t;
  const f3 D = qrot(Tsm.R, dir_s);
  const bool finite = (D.x == D.x) && (D.y == D.y) && (D.z == D.z);
  const float ray_tfar = (valid && finite) ? p.tfar : -1.0f;

  const RaySlab rs = make_ray_slab(O, D);
  float best_t = ray_tfar;
  uint32_t best_rec = kNone;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  uint32_t* lds_stack = lds_dyn + threadIdx.x;
  constexpr uint32_t lds_stride = 256u;
  uint32_t priv[48];
  uint32_t sp = 0;
  uint32_t cur = (ray_tfar >= 0.0f) ? 0u : kDone;
  uint32_t step = 0;
#define RMCL_PUSH(v) { if (sp < 16u) lds_stack[sp * lds_stride] = (v); else priv[sp - 16u] = (v); ++sp; }
#define RMCL_POP() { if (sp == 0) cur = kDone; else { --sp; if (sp < 16u) cur = lds_stack[sp * lds_stride]; else cur = priv[sp - 16u]; } }
  // calibration: two stamps with nothing between them = the cost every interval below includes once
  RMCL_PROBE(9u, true, __ballot(true), false, 0u)
  RMCL_PROBE(10u, true, __ballot(true), false, 0u)
  while (__any(cur != kDone)) {
    for (;;) {
      const bool inner = (cur != kDone) && !(cur & kLeafBit);
      const uint64_t m = __ballot(inner);
      if (m == 0) break;
      const uint32_t c0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cur), __builtin_ctzll(m)));
      const bool uni = __ballot(inner && cur != c0) == 0;
      RMCL_PROBE(1u, true, m, uni, step)
      uint4 qnx = {0, 0, 0, 0}, qfx = qnx, qny = qnx, qfy = qnx, qnz = qnx, qfz = qnx, qch = qnx;
      if (inner) {
        const char* nb = node_address<kTop>(p.nodes, lds_dyn + 16u * 256u, cur);
        qnx = *reinterpret_cast<const uint4*>(nb + rs.onx); qfx = *reinterpret_cast<const uint4*>(nb + rs.ofx);
        qny = *reinterpret_cast<const uint4*>(nb + rs.ony); qfy = *reinterpret_cast<const uint4*>(nb + rs.ofy);
        qnz = *reinterpret_cast<const uint4*>(nb + rs.onz); qfz = *reinterpret_cast<const uint4*>(nb + rs.ofz);
        qch = *reinterpret_cast<const uint4*>(nb + 96);
      }
      RMCL_PROBE(2u, true, m, uni, step)   // the stamp drains vmcnt: node data arrived
      if (inner) {
        const f2 ix = {rs.inv.x, rs.inv.x}, iy = {rs.inv.y, rs.inv.y}, iz = {rs.inv.z, rs.inv.z};
        const f2 nx = {rs.noi.x, rs.noi.x}, ny = {rs.noi.y, rs.noi.y}, nz = {rs.noi.z, rs.noi.z};
        const f2 nx01 = __builtin_elementwise_fma(f2{asf(qnx.x), asf(qnx.y)}, ix, nx), nx23 = __builtin_elementwise_fma(f2{asf(qnx.z), asf(qnx.w)}, ix, nx);
        const f2 fx01 = __builtin_elementwise_fma(f2{asf(qfx.x), asf(qfx.y)}, ix, nx), fx23 = __builtin_elementwise_fma(f2{asf(qfx.z), asf(qfx.w)}, ix, nx);
        const f2 ny01 = __builtin_elementwise_fma(f2{asf(qny.x), asf(qny.y)}, iy, ny), ny23 = __builtin_elementwise_fma(f2{asf(qny.z), asf(qny.w)}, iy, ny);
        const f2 fy01 = __builtin_elementwise_fma(f2{asf(qfy.x), asf(qfy.y)}, iy, ny), fy23 = __builtin_elementwise_fma(f2{asf(qfy.z), asf(qfy.w)}, iy, ny);
        const f2 nz01 = __builtin_elementwise_fma(f2{asf(qnz.x), asf(qnz.y)}, iz, nz), nz23 = __builtin_elementwise_fma(f2{asf(qnz.z), asf(qnz.w)}, iz, nz);
        const f2 fz01 = __builtin_elementwise_fma(f2{asf(qfz.x), asf(qfz.y)}, iz, nz), fz23 = __builtin_elementwise_fma(f2{asf(qfz.z), asf(qfz.w)}, iz, nz);
        const float tnx[4] = {nx01.x, nx01.y, nx23.x, nx23.y}, tfx[4] = {fx01.x, fx01.y, fx23.x, fx23.y};
        const float tny[4] = {ny01.x, ny01.y, ny23.x, ny23.y}, tfy[4] = {fy01.x, fy01.y, fy23.x, fy23.y};
        const float tnz[4] = {nz01.x, nz01.y, nz23.x, nz23.y}, tfz[4] = {fz01.x, fz01.y, fz23.x, fz23.y};
        uint32_t key[4], ref[4] = {qch.x, qch.y, qch.z, qch.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float tn = fmaxf(fmaxf(fmaxf(tnx[c], tny[c]), tnz[c]), 0.0f);
          const float tf = fminf(fminf(fminf(tfx[c], tfy[c]), tfz[c]), best_t);
          key[c] = (tn <= tf) ? __float_as_uint(tn) : kNone;
        }
        RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
        if (key[3] != kNone) RMCL_PUSH(ref[3])
        if (key[2] != kNone) RMCL_PUSH(ref[2])
        if (key[1] != kNone) RMCL_PUSH(ref[1])
        if (key[0] != kNone) cur = ref[0];
        else RMCL_POP()
      }
      RMCL_PROBE(3u, true, m, uni, step)
      ++step;
    }
    {
      const bool leaf = (cur != kDone);
      const uint64_t m = __ballot(leaf);
      if (m != 0) {
        if (kLeafBatch) {
          RMCL_PROBE(4u, true, m, false, step)
          if (leaf) {
            leaf_batch(p.tris, cur, O, D, ray_tfar, best_t, best_rec);
            RMCL_POP()
          }
          RMCL_PROBE(6u, true, m, false, step)
        } else {
          const uint32_t first = cur & 0x0FFFFFFFu;
          const uint32_t cnt = leaf ? (((cur >> 28) & 7u) + 1u) : 0u;
          for (uint32_t i = 0; __any(i < cnt); ++i) {
            const uint64_t mi = __ballot(i < cnt);
            RMCL_PROBE(4u, true, mi, false, step)
            uint4 a = {0, 0, 0, 0}, b = a, c = a;
            if (i < cnt) {
              const uint4* tp = reinterpret_cast<const uint4*>(p.tris) + static_cast<size_t>(first + i) * 4u;
              a = tp[0]; b = tp[1]; c = tp[2];
            }
            RMCL_PROBE(5u, true, mi, false, step)
            if (i < cnt) tri_update(a, b, c, first + i, p.tris, O, D, ray_tfar, best_t, best_rec);
            RMCL_PROBE(6u, true, mi, false, step)
          }
          if (leaf) RMCL_POP()
        }
        ++step;
      }
    }
  }
#undef RMCL_PUSH
#undef RMCL_POP
  RMCL_PROBE(7u, true, __ballot(true), false, step)
  if (valid) {
    const size_t g = loc;
    const bool found = (best_rec != kNone);
    if (found) {
      p.hits[g] = 1;
      p.ranges[g] = best_t;
      const f3 pt = scale3(dir_s, best_t);
      p.points[3 * g] = pt.x; p.points[3 * g + 1] = pt.y; p.points[3 * g + 2] = pt.z;
      const uint4 nrec = reinterpret_cast<const uint4*>(p.tris)[static_cast<size_t>(best_rec) * 4u + 3u];
      f3 n = qrot(Tms.R, mk3(asf(nrec.x), asf(nrec.y), asf(nrec.z)));
      if (dot_plain(dir_s, n) > 0.0f) n = neg3(n);
      p.normals[3 * g] = n.x; p.normals[3 * g + 1] = n.y; p.normals[3 * g + 2] = n.z;
      p.face_ids[g] = nrec.w;
    } else {
      const float qn = __uint_as_float(0x7FC00000u);
      p.hits[g] = 0;
      p.ranges[g] = p.tfar + 1.0f;
      p.points[3 * g] = qn; p.points[3 * g + 1] = qn; p.points[3 * g + 2] = qn;
      p.normals[3 * g] = qn; p.normals[3 * g + 1] = qn; p.normals[3 * g + 2] = qn;
      p.face_ids[g] = kInvalidFace;
    }
  }
  RMCL_PROBE(8u, true, __ballot(true), false, step)
#undef RMCL_PROBE
  if (lane == 0u) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    log[0] = min(nlog, kProbeEntries) | (xcc << 16);
    log[1] = t_begin;   // absolute shader clock (low 32 bits) at wave start
  }
}

// Persistent-lane variant of k_pf_update ("dynamic ray fetch", Aila & Laine 2009).  The beams of a particle point in
// all directions, so the 64 rays of a wave diverge almost immediately and with one ray per lane per round the
// wave waits for its slowest ray: PMC showed 52 % of the lanes active in VALU instructions.  Here a lane that has
// finished its ray takes the next one from the block's queue as soon as kRefill lanes of its wave are idle; every
// result is stored under its ray index, so the outcome does not depend on the schedule.  Traversal, acceptance
// rules and beam evaluation are those of trace_lane_ww / k_pf_update (bit-identical results).
template <int kLdsEntries, int kRefill, bool kQuant>
__global__ void __launch_bounds__(256) k_pf_update_persist(const PfParams p) {
  // LDS: [ per-lane stacks kLdsEntries*256 | Tsm (PB xforms) | evals (PB*n_beams floats) ]
  extern __shared__ uint32_t lds_dyn[];
  __shared__ uint32_t s_next;
  uint32_t* lds_col = lds_dyn + threadIdx.x;   // stack rows of this lane: row r at lds_col[r * 256], row 0 = sentinel
  xform* s_Tsm = reinterpret_cast<xform*>(lds_dyn + kLdsEntries * 256);
  float* s_eval = reinterpret_cast<float*>(s_Tsm + p.particles_per_block);

  const uint32_t PB = p.particles_per_block;
  const uint32_t p0 = blockIdx.x * PB;
  if (p0 >= p.n_particles) return;
  const uint32_t np = min(PB, p.n_particles - p0);
  if (threadIdx.x < np) s_Tsm[threadIdx.x] = xmul(p.poses[p0 + threadIdx.x], p.Tsb);
  if (threadIdx.x == 0) s_next = 0u;
  __syncthreads();

  const float sq = p.dist_sigma * p.dist_sigma;
  const uint32_t nrays = np * p.n_beams;
  const uint32_t lane = threadIdx.x & 63u;
  constexpr uint32_t kDone = 0x7FFFFFFFu;
  // per-lane ray state
  uint32_t rr = 0;
  bool has_ray = false, exhausted = false;
  f3 O = mk3(0.f, 0.f, 0.f), D = O;
  RaySlab rs = make_ray_slab(O, mk3(1.f, 1.f, 1.f));
  float range = 0.f, best_t = 0.f;
  uint32_t best_rec = kNone;
  // branch-free node step of trace_lane_bf: kLdsEntries rows in LDS (row 0 = sentinel kDone), deeper rows in scratch
  constexpr int kRows = kLdsEntries;
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
    if (want != 0 && (busy == 0 || __popcll(want) >= kRefill)) {
      if (idle) {
        if (has_ray) {
          // evaluate_rcc (PCDSensorUpdaterEmbree.cpp:18-86) with unit face normals (BeamEvaluateProgram.cu:104-113)
          const uint32_t pi = rr / p.n_beams, b = rr - pi * p.n_beams;
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
          if (p.errors) p.errors[static_cast<size_t>(p0 + pi) * p.n_beams + b] = error;
          // PCDSensorUpdaterEmbree.cpp:224 : float argument, double exp / sqrt, float result
          const float arg = -(error * error) / sq / 2;
          s_eval[rr] = static_cast<float>(exp(static_cast<double>(arg)) /
                                          sqrt(static_cast<double>(2 * sq) * 3.14159265358979323846));
          has_ray = false;
        }
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
          rr = mine;
          const uint32_t pi = rr / p.n_beams, b = rr - pi * p.n_beams;
          const xform Tsm = s_Tsm[pi];
          const float* bm = p.beams + 16u * b;
          // meas_m = Tsm * meas_s (RangeMeasurement.hpp:28-42)
          D = qrot(Tsm.R, mk3(bm[3], bm[4], bm[5]));
          O = xapply(Tsm, mk3(bm[0], bm[1], bm[2]));
          range = bm[6];
          rs = make_ray_slab(O, D);
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
    // phase 1: inner nodes (see trace_lane_ww) -- left EARLY once at most kTailLanes lanes are still descending while
    // others already hold a leaf: the stragglers resume in the next round and the leaf holders do not idle through
    // the tail (measured 7 % / 5 % faster on sphere / room; for coherent scans the plain loop of trace_lane_ww wins)
    constexpr int kTailLanes = 8;
    for (;;) {
      const bool inner = (cur != kDone) && !(cur & kLeafBit);
      const uint64_t m_inner = __ballot(inner);
      if (m_inner == 0) break;
      if (__popcll(m_inner) <= kTailLanes && __ballot((cur != kDone) && (cur & kLeafBit)) != 0) break;
      if (inner) {
        uint32_t key[4], ref[4];
        if (!__any(sp + 3u > static_cast<uint32_t>(kRows))) {
          const uint32_t top = lds_col[(sp - 1u) * kBfStride];
          if (kQuant) node_keys_q(p.qnodes, cur, rs, best_t, key, ref);
          else node_keys_off(p.nodes, cur << 7, rs, best_t, key, ref);
          RMCL_CSWAP(0, 1) RMCL_CSWAP(2, 3) RMCL_CSWAP(0, 2) RMCL_CSWAP(1, 3) RMCL_CSWAP(1, 2)
          lds_col[sp * kBfStride] = ref[3]; sp += (key[3] != kNone) ? 1u : 0u;
          lds_col[sp * kBfStride] = ref[2]; sp += (key[2] != kNone) ? 1u : 0u;
          lds_col[sp * kBfStride] = ref[1]; sp += (key[1] != kNone) ? 1u : 0u;
          const bool any = key[0] != kNone;
          cur = any ? ref[0] : top;
          sp = any ? sp : (sp - 1u);
        } else {
          if (kQuant) node_keys_q(p.qnodes, cur, rs, best_t, key, ref);
          else node_keys_off(p.nodes, cur << 7, rs, best_t, key, ref);
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
      leaf_loop(p.tris, cur, O, D, p.ray_tfar, best_t, best_rec);
      --sp;
      cur = RMCL_ROW_LD(sp);
    }
  }
#undef RMCL_ROW_ST
#undef RMCL_ROW_LD
  __syncthreads();
  // in-order merge, one lane per particle (sequential semantics of sensorUpdate, :232-238)
  if (threadIdx.x < np) {
    pattrs* A = reinterpret_cast<pattrs*>(p.attrs) + (p0 + threadIdx.x);
    g1d L = A->likelihood;
    const float* ev = s_eval + threadIdx.x * p.n_beams;
    for (uint32_t b = 0; b < p.n_beams; ++b) {
      g1d m; m.mean = ev[b]; m.sigma = 0.0f; m.n_meas = 1;
      L = g1d_add(L, m);
      L.n_meas = min(L.n_meas, p.max_n_meas);
    }
    A->likelihood = L;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers of the experiments (registered with librmclhip.so below)
// ---------------------------------------------------------------------------------------------
namespace {

// spherical model only: the experiments and the clock timelines are run on the benchmark scans
template <int kTrav, bool kClock>
hipError_t lab_find_one(const FindParams& p, dim3 grid, size_t lds, hipStream_t s) {
  if (lds > 65536u) {
    // more than 64 KB of dynamic LDS per block must be granted per kernel and per device; rare A/B kinds, so simply repeated
    const hipError_t ge = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_find<kModelSpherical, kTrav, kClock>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (ge != hipSuccess) return ge;
  }
  hipLaunchKernelGGL((k_find<kModelSpherical, kTrav, kClock>), grid, dim3(256), lds, s, p);
  return hipGetLastError();
}

template <bool kClock>
hipError_t lab_find_kind(const FindParams& p, int variant, hipStream_t s) {
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  uint32_t nblocks = (variant == 2 || variant == 25) ? ntiles : (ntiles + 3u) / 4u;
  nblocks = (nblocks + 7u) & ~7u;  // the XCD remap in k_find needs gridDim.x % 8 == 0
  const dim3 grid(nblocks, p.nposes, 1);
  const size_t lds_bf = static_cast<size_t>(kFindBfRows) * 256u * sizeof(uint32_t);
  const size_t lds_bf_tail = (static_cast<size_t>(kFindBfRows) * 256u + kQuadStackEntries * 64u + 4u * kTailRays * kTailXferDwords) * sizeof(uint32_t);
  const size_t lds_ww = 16u * 256u * sizeof(uint32_t);
  const size_t lds_ww_tail = (kFindTailLdsDwords + static_cast<uint32_t>(find_top_nodes(variant)) * kNodeDwords) * sizeof(uint32_t);
  switch (variant) {
    case 0: return lab_find_one<0, kClock>(p, grid, 0, s);                                     // (clocked only: the product owns kind 0)
    case 1: return lab_find_one<1, kClock>(p, grid, lds_bf, s);                                // branch-free step, no tail
    case 2: return lab_find_one<2, kClock>(p, grid, kQuadStackEntries * 64u * sizeof(uint32_t), s);
    case 4: return lab_find_one<4, kClock>(p, grid, lds_ww, s);
    case 5: return lab_find_one<5, kClock>(p, grid, lds_ww_tail, s);                           // while-while + quad-finished tails
    case 6: return lab_find_one<6, kClock>(p, grid, lds_ww_tail, s);                           // + 85-node LDS top
    case 7: return lab_find_one<7, kClock>(p, grid, lds_ww_tail, s);                           // + 341-node LDS top
    case 8: return lab_find_one<8, kClock>(p, grid, lds_ww_tail, s);                           // + one-round-trip leaves
    case 9: return lab_find_one<9, kClock>(p, grid, lds_ww_tail, s);
    case 10: return lab_find_one<10, kClock>(p, grid, lds_ww_tail, s);
    case 11: return lab_find_one<11, kClock>(p, grid, lds_ww, s);                              // the branchy while-while step
    case 12: return lab_find_one<12, kClock>(p, grid, lds_bf, s);                              // branch-free + one-round-trip leaves
    case 13: return lab_find_one<13, kClock>(p, grid, lds_bf, s);                              // wave-uniform nodes through the scalar cache
    case 14: return lab_find_one<14, kClock>(p, grid, lds_bf, s);
    case 16: return lab_find_one<16, kClock>(p, grid, lds_bf_tail, s);                         // branch-free + tails
    case 17: return lab_find_one<17, kClock>(p, grid, lds_bf_tail, s);                         // + one-round-trip leaves (round 2's default before the trigger)
    case 19: return lab_find_one<19, kClock>(p, grid, lds_bf_tail, s);
    case 20: return lab_find_one<20, kClock>(p, grid, lds_bf_tail, s);                         // 16 + leaf trigger
    case 21: return lab_find_one<21, kClock>(p, grid, kFindTailLdsDwords * sizeof(uint32_t), s);
    case 22: return lab_find_one<22, kClock>(p, grid, lds_ww, s);
    case 23: return lab_find_one<23, kClock>(p, grid, lds_bf_tail, s);                         // 19 + frontier start
    case 24: return lab_find_one<24, kClock>(p, grid, lds_ww, s);                              // 22 + frontier start
    case 30: return lab_find_one<30, kClock>(p, grid, lds_bf_tail, s);                           // 23, only the nearest child found
    case 29: return lab_find_one<29, kClock>(p, grid, lds_bf_tail, s);                           // 23, the middle two children unordered
    case 28: return lab_find_one<28, kClock>(p, grid, lds_bf_tail, s);                           // 23 with the software-pipelined node step
    case 27: return lab_find_one<27, kClock>(p, grid, lds_bf_tail, s);                           // 23 + prefetch of the hit record's normal (experiment iv)
    case 26: return lab_find_one<26, kClock>(p, grid, lds_bf_tail, s);                           // 23 on the quantised nodes
    case 25: return lab_find_one<25, kClock>(p, grid, kQuadStackEntries * 64u * sizeof(uint32_t), s);   // 2 + frontier start
    case 31: return lab_find_one<31, kClock>(p, grid, kFind31LdsDwords * sizeof(uint32_t), s);   // 23 + cooperative descent, sorted hand-over of the final entries (round 6's first form)
    case 32: return lab_find_one<32, kClock>(p, grid, kFind31LdsDwords * sizeof(uint32_t), s);   // (clocked only) 31 with one bit per final leaf and ray instead of the sorted hand-over
    default: return kLabMissing;
  }
}

hipError_t lab_find(const FindParams& p, ModelKind kind, int variant, bool with_clock, hipStream_t s) {
  if (kind != kModelSpherical) return kLabMissing;
  if (with_clock) return lab_find_kind<true>(p, variant, s);
  if (find_kind_in_product(variant)) return hipErrorInvalidValue;   // the product launches its own kinds
  return lab_find_kind<false>(p, variant, s);
}

hipError_t lab_find_probe(const FindParams& p, int mode, uint32_t* probe_log, hipStream_t s) {
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  uint32_t nblocks = ((ntiles + 3u) / 4u + 7u) & ~7u;
  const dim3 grid(nblocks, 1, 1), block(256, 1, 1);
  const size_t lds0 = 16u * 256u * sizeof(uint32_t);
  if (mode == 0) hipLaunchKernelGGL((k_find_probe<false, 0>), grid, block, lds0, s, p, probe_log);
  else if (mode == 1) hipLaunchKernelGGL((k_find_probe<true, 0>), grid, block, lds0, s, p, probe_log);
  else if (mode == 2) hipLaunchKernelGGL((k_find_probe<false, 85>), grid, block, lds0 + 85u * 128u, s, p, probe_log);
  else hipLaunchKernelGGL((k_find_probe<true, 85>), grid, block, lds0 + 85u * 128u, s, p, probe_log);
  return hipGetLastError();
}

// the round kernels (refill 0) and the round-2 persistent kernel (bit 8, or bit 7 = full 128-B nodes)
hipError_t lab_pf_update(const PfParams& p, int variant, hipStream_t s) {
  const uint32_t nblocks = (p.n_particles + p.particles_per_block - 1u) / p.particles_per_block;
  const size_t tail = sizeof(xform) * p.particles_per_block +
                      sizeof(float) * static_cast<size_t>(p.particles_per_block) * p.n_beams;
  const int trav = variant & 3;        // see k_pf_update
  const bool deep = (variant & 4) != 0;  // 64-deep LDS stack instead of 32 (maps with stack_need > 32)
  const size_t stack_lds = ((trav == 0) ? 16u : (deep ? 64u : 32u)) * 256u * sizeof(uint32_t);
  size_t lds = stack_lds + tail;
  const int refill = (variant >> 4) & 7;  // 0 = rounds of one ray per lane; 1..4 = persistent lanes, refill at 8/16/32/48 idle
  if ((variant & 8) != 0) return hipErrorInvalidValue;   // closest-point form: the product's own
  if (trav == 0 && refill != 0) {
    lds = static_cast<size_t>(kPfRows) * 256u * sizeof(uint32_t) + tail;
    const bool quant = ((variant >> 7) & 1) == 0 && p.qnodes != nullptr;  // bit 7: full-precision nodes (A/B)
    if (quant) {
      if (refill == 1) hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 8, true>), dim3(nblocks), dim3(256), lds, s, p);
      else if (refill == 2) hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 16, true>), dim3(nblocks), dim3(256), lds, s, p);
      else if (refill == 3) hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 32, true>), dim3(nblocks), dim3(256), lds, s, p);
      else hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 48, true>), dim3(nblocks), dim3(256), lds, s, p);
    } else {
      if (refill == 1) hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 8, false>), dim3(nblocks), dim3(256), lds, s, p);
      else if (refill == 2) hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 16, false>), dim3(nblocks), dim3(256), lds, s, p);
      else if (refill == 3) hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 32, false>), dim3(nblocks), dim3(256), lds, s, p);
      else hipLaunchKernelGGL((k_pf_update_persist<kPfRows, 48, false>), dim3(nblocks), dim3(256), lds, s, p);
    }
    return hipGetLastError();
  }
  if (trav == 0) hipLaunchKernelGGL((k_pf_update<64, 0>), dim3(nblocks), dim3(256), lds, s, p);
  else if (trav == 1 && deep) hipLaunchKernelGGL((k_pf_update<64, 1>), dim3(nblocks), dim3(256), lds, s, p);
  else if (trav == 1) hipLaunchKernelGGL((k_pf_update<32, 1>), dim3(nblocks), dim3(256), lds, s, p);
  else if (deep) hipLaunchKernelGGL((k_pf_update<64, 2>), dim3(nblocks), dim3(256), lds, s, p);
  else hipLaunchKernelGGL((k_pf_update<32, 2>), dim3(nblocks), dim3(256), lds, s, p);
  return hipGetLastError();
}

const LabHooks kHooks = {lab_find, lab_find_probe, lab_pf_update};

// registration when the library is loaded (dlopen / ctypes.CDLL), removal when it goes
struct Registrar {
  Registrar() { rmclhip_internal_register_lab(&kHooks); }
  ~Registrar() { rmclhip_internal_register_lab(nullptr); }
} g_registrar;

}  // namespace

}  // namespace rmclhip

extern "C" const char* rmclhip_lab_version(void) { return "rmclhip-lab 0.1 (gfx950): experiments, not the product"; }
