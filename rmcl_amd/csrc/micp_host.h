// micp_host.h -- the gate-stable moment form of the MICP iterations, evaluated on the HOST.
//
// Between the iterations of MICPLocalizationNode::correctOnce (micp_localization.cpp:915-964) the correspondences are fixed and
// only the pre-transform (R, t) of statistics_p2l changes (CorrespondencesCPU.cpp:26, MICPSensorCPU.cpp:70-84):
//   D' = R D + t;  dist = N . I - N . D';  M = D' + N dist;  keep iff |dist| < max_dist.
// For the correspondences whose gate decision cannot change (devmath.h micp_gate_class == 1) the 16 raw sums of the reduction
// (sum D', sum M, sum M D'^T, n) are polynomials in (R, t) whose coefficients are 82 moments of (D, N, s = N . I); the device
// forms them once per find (find_kernel.hip.h find_moments_wave / kernels.hip k_micp_moments), folds them and hands them to the
// host together with the few undecided correspondences (k_micp_publish).  An iteration is then O(1): ~1.5 k flops here instead of
// a 131 072-element streaming reduce plus a launch and a completion wait (or, in the device loop of round 3, ~4.4 k cycles of
// ONE lane's dependent f64 chain).
//
// Moment row layout (kernels.hip kMom):  n | D[3] | DD[6] | sN[3] | sND[9] | NN[6] | NND[18] | NNDD[36],  symmetric pairs in
// the order 00 01 02 11 12 22.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "devmath.h"

namespace rmclhip {

constexpr uint32_t kMicpMomentRow = 96;     // == kMicpFastMoments (82 used)
constexpr uint32_t kMicpMomentsUsed = 82;
constexpr uint32_t kMicpHostMaxUnc = 1024;  // undecided correspondences the device hands over (9 floats each); more -> device loop
constexpr uint32_t kMicpUncLanes = 8;       // the host sums them in eight interleaved partial sums (one AVX2 register of floats)
constexpr uint32_t kMicpUncPadded = kMicpHostMaxUnc + kMicpUncLanes;

// what k_micp_publish writes into pinned host memory (one block per operator).  The completion tag's sum covers mom[0..95], the
// four header words and 9 * min(n_uncertain, kMicpHostMaxUnc) words of `unc` when code == 0.
struct MicpHostBlock {
  double mom[kMicpMomentRow];
  uint32_t code;          // 0 = complete; 2 = more than kMicpHostMaxUnc undecided correspondences (unc not written)
  uint32_t n_uncertain;
  uint32_t pad[2];
  float unc[kMicpHostMaxUnc][9];   // D xyz | I xyz | N xyz of every undecided correspondence, in index order
};

// the host's copy of one published block + what it is valid for
struct MicpMomentSet {
  bool valid = false;
  double mom[kMicpMomentsUsed];
  float gate_lo = 0.f, gate_hi = 0.f, rho_cap = 0.f, tau_cap = 0.f;
  uint32_t n_unc = 0;
  // the undecided correspondences, component-major (D xyz | I xyz | N xyz), padded to a multiple of kMicpUncLanes with entries whose
  // distance is infinite (they fail every gate)
  alignas(32) float unc[9][kMicpUncPadded];
  void set_undecided(const float (*aos)[9], uint32_t n) {
    n_unc = n;
    for (uint32_t e = 0; e < n; ++e)
      for (int k = 0; k < 9; ++k) unc[k][e] = aos[e][k];
    const uint32_t np = (n + kMicpUncLanes - 1u) / kMicpUncLanes * kMicpUncLanes;
    for (uint32_t e = n; e < np; ++e) {
      for (int k = 0; k < 9; ++k) unc[k][e] = 0.0f;
      unc[3][e] = __builtin_inff();   // I.x = inf, N = (1, 0, 0): |dist| = inf
      unc[6][e] = 1.0f;
    }
  }
};

inline int mh_sym3(int a, int b) {
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  return lo == 0 ? hi : (lo == 1 ? hi + 2 : 5);
}
// X ranges over the ten per-correspondence factors  N_aN_b (6) | 1 | s N_a (3):  W(X) = sum X, P(X)_j = sum X D_j, Q(X)_jk = sum X D_j D_k
inline const double* mh_Q(const double* mom, int x) { return x < 6 ? mom + 46 + 6 * x : mom + 4; }
inline const double* mh_P(const double* mom, int x) { return x < 6 ? mom + 28 + 3 * x : (x == 6 ? mom + 1 : mom + 13 + 3 * (x - 7)); }
inline double mh_W(const double* mom, int x) { return x < 6 ? mom[22 + x] : (x == 6 ? mom[0] : mom[10 + (x - 7)]); }

// linear map of qrot (q v q*) in double from the f32 components (as k_micp_fast_loop forms it)
inline void micp_rotation_f64(const quat& q, double* R) {
  const double x = q.x, y = q.y, z = q.z, w = q.w;
  const double ww = w * w, uu = (x * x + y * y) + z * z;
  R[0] = (ww - uu) + 2.0 * x * x; R[1] = 2.0 * (x * y - w * z);   R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z);   R[4] = (ww - uu) + 2.0 * y * y; R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y);   R[7] = 2.0 * (y * z + w * x);   R[8] = (ww - uu) + 2.0 * z * z;
}

// the 16 raw sums (sd[3] sm[3] smd[9] n) over the certainly-gated-in correspondences at pre-transform (R row-major, t)
inline void micp_moment_sums(const double* mom, const double* R, const double* t, double* out) {
  double H[7][9];    // H(X)_bc = sum_jk R_bj R_ck Q(X)_jk, X = 0..6
  double rP[10][3];  // rP(X)_b = sum_j R_bj P(X)_j
  for (int x = 0; x < 7; ++x) {
    const double* Q = mh_Q(mom, x);
    double G[9];     // G_cj = sum_k Q_jk R_ck
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < 3; ++j)
        G[3 * c + j] = (Q[mh_sym3(j, 0)] * R[3 * c] + Q[mh_sym3(j, 1)] * R[3 * c + 1]) + Q[mh_sym3(j, 2)] * R[3 * c + 2];
    for (int b = 0; b < 3; ++b)
      for (int c = 0; c < 3; ++c)
        H[x][3 * b + c] = (R[3 * b] * G[3 * c] + R[3 * b + 1] * G[3 * c + 1]) + R[3 * b + 2] * G[3 * c + 2];
  }
  for (int x = 0; x < 10; ++x) {
    const double* P = mh_P(mom, x);
    for (int b = 0; b < 3; ++b) rP[x][b] = (R[3 * b] * P[0] + R[3 * b + 1] * P[1]) + R[3 * b + 2] * P[2];
  }
  const double n = mom[0];
  double sd[3];
  for (int c = 0; c < 3; ++c) sd[c] = rP[6][c] + n * t[c];   // sum D'_c
  for (int c = 0; c < 3; ++c) out[c] = sd[c];
  for (int a = 0; a < 3; ++a) {
    double ndp = 0.0;   // sum N_a (N . D')
    for (int b = 0; b < 3; ++b) {
      const int x = mh_sym3(a, b);
      ndp += rP[x][b] + t[b] * mh_W(mom, x);
    }
    out[3 + a] = sd[a] + mom[10 + a] - ndp;
  }
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) {
      const double ddp = H[6][3 * a + c] + rP[6][a] * t[c] + t[a] * rP[6][c] + n * t[a] * t[c];   // sum D'_a D'_c
      const double sndp = rP[7 + a][c] + mom[10 + a] * t[c];                                      // sum s N_a D'_c
      double nndd = 0.0;                                                                          // sum N_a (N . D') D'_c
      for (int b = 0; b < 3; ++b) {
        const int x = mh_sym3(a, b);
        nndd += H[x][3 * b + c] + t[c] * rP[x][b] + t[b] * rP[x][c] + t[b] * t[c] * mh_W(mom, x);
      }
      out[6 + 3 * a + c] = ddp + sndp - nndd;
    }
  out[15] = n;
}

// The undecided correspondences with the reduction's own f32 arithmetic (kernels.hip k_reduce_partials / k_micp_iter: xapply = the two
// quaternion products of devmath.h qrot, rmagine's dot, the gate |dist| < max_dist), f64 sums.  Order of the sums: eight interleaved
// partial sums (element e goes to lane e % 8), folded ((0+1)+(2+3))+((4+5)+(6+7)) at the end -- the shape of one AVX2 register, so the
// same routine compiles to vector code where the CPU has it and to the identical arithmetic, lane by lane, where it has not: results
// do not depend on the machine.  ~3 ns per correspondence with AVX2, ~16 ns in the one-at-a-time form it replaces (round 4).
namespace micp_simd {
typedef float v8f __attribute__((vector_size(32)));
typedef double v4d __attribute__((vector_size(32)));
typedef int v8i __attribute__((vector_size(32)));
typedef float v4f __attribute__((vector_size(16)));
struct q8 { v8f x, y, z, w; };
static inline __attribute__((always_inline)) v8f splat(float a) { return v8f{a, a, a, a, a, a, a, a}; }
static inline __attribute__((always_inline)) q8 qmul8(const q8& a, const q8& b) {   // devmath.h qmul, operation for operation
  q8 r;
  r.w = ((a.w * b.w - a.x * b.x) - a.y * b.y) - a.z * b.z;
  r.x = ((a.w * b.x + a.x * b.w) + a.y * b.z) - a.z * b.y;
  r.y = ((a.w * b.y - a.x * b.z) + a.y * b.w) + a.z * b.x;
  r.z = ((a.w * b.z + a.x * b.y) - a.y * b.x) + a.z * b.w;
  return r;
}
static inline __attribute__((always_inline)) void body(const float (*unc)[kMicpUncPadded], uint32_t n_unc, const xform& Tpre, float max_dist,
                                                      double* acc) {
  const q8 q = {splat(Tpre.R.x), splat(Tpre.R.y), splat(Tpre.R.z), splat(Tpre.R.w)};
  const q8 qi = {splat(-Tpre.R.x), splat(-Tpre.R.y), splat(-Tpre.R.z), splat(Tpre.R.w)};
  const v8f tx = splat(Tpre.t.x), ty = splat(Tpre.t.y), tz = splat(Tpre.t.z), gate = splat(max_dist), zero = splat(0.0f);
  v4d a[16][2];
  for (int k = 0; k < 16; ++k) { a[k][0] = v4d{0.0, 0.0, 0.0, 0.0}; a[k][1] = v4d{0.0, 0.0, 0.0, 0.0}; }
  const uint32_t np = (n_unc + kMicpUncLanes - 1u) / kMicpUncLanes * kMicpUncLanes;
  for (uint32_t e = 0; e < np; e += kMicpUncLanes) {
    v8f d0, d1, d2, i0, i1, i2, n0, n1, n2;
    __builtin_memcpy(&d0, unc[0] + e, 32); __builtin_memcpy(&d1, unc[1] + e, 32); __builtin_memcpy(&d2, unc[2] + e, 32);
    __builtin_memcpy(&i0, unc[3] + e, 32); __builtin_memcpy(&i1, unc[4] + e, 32); __builtin_memcpy(&i2, unc[5] + e, 32);
    __builtin_memcpy(&n0, unc[6] + e, 32); __builtin_memcpy(&n1, unc[7] + e, 32); __builtin_memcpy(&n2, unc[8] + e, 32);
    const q8 P = {d0, d1, d2, zero};
    const q8 PT = qmul8(qmul8(q, P), qi);                       // qrot
    const v8f D0 = PT.x + tx, D1 = PT.y + ty, D2 = PT.z + tz;   // xapply
    const v8f s0 = i0 - D0, s1 = i1 - D1, s2 = i2 - D2;
    const v8f spd = (s0 * n0 + s1 * n1) + s2 * n2;              // dot_plain(sub3(I, D'), N)
    const v8f aspd = spd < zero ? -spd : spd;                   // fabsf (NaN stays NaN and fails the gate)
    const v8i in = aspd < gate;
    const v8f M0 = D0 + n0 * spd, M1 = D1 + n1 * spd, M2 = D2 + n2 * spd;
    const v8f dm[6] = {D0, D1, D2, M0, M1, M2};
    v4d dv[6][2];
    for (int k = 0; k < 6; ++k) {
      // a lane outside the gate contributes +0.0 to every sum (x + 0.0 == x for the sums' x: they start at +0.0)
      const v8f v = in ? dm[k] : zero;
      dv[k][0] = __builtin_convertvector(__builtin_shufflevector(v, v, 0, 1, 2, 3), v4d);
      dv[k][1] = __builtin_convertvector(__builtin_shufflevector(v, v, 4, 5, 6, 7), v4d);
    }
    const v8f one = in ? splat(1.0f) : zero;
    const v4d cnt[2] = {__builtin_convertvector(__builtin_shufflevector(one, one, 0, 1, 2, 3), v4d),
                        __builtin_convertvector(__builtin_shufflevector(one, one, 4, 5, 6, 7), v4d)};
    for (int h = 0; h < 2; ++h) {
      for (int k = 0; k < 3; ++k) { a[k][h] += dv[k][h]; a[3 + k][h] += dv[3 + k][h]; }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a[6 + 3 * r + c][h] += dv[3 + r][h] * dv[c][h];
      a[15][h] += cnt[h];
    }
  }
  for (int k = 0; k < 16; ++k)
    acc[k] += ((a[k][0][0] + a[k][0][1]) + (a[k][0][2] + a[k][0][3])) + ((a[k][1][0] + a[k][1][1]) + (a[k][1][2] + a[k][1][3]));
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void body_avx2(const float (*unc)[kMicpUncPadded], uint32_t n_unc, const xform& Tpre, float max_dist,
                                                      double* acc) {
  body(unc, n_unc, Tpre, max_dist, acc);
}
#endif
}  // namespace micp_simd

inline void micp_undecided_sums(const float (*unc)[kMicpUncPadded], uint32_t n_unc, const xform& Tpre, float max_dist, double* acc) {
  if (n_unc == 0u) return;
#if defined(__x86_64__)
  // (RMCLHIP_NO_AVX2: the portable body on a machine that has AVX2 -- tests/test_host_logic.py checks that the two agree bit for bit)
  static const bool have_avx2 = __builtin_cpu_supports("avx2") != 0 && std::getenv("RMCLHIP_NO_AVX2") == nullptr;
  if (have_avx2) { micp_simd::body_avx2(unc, n_unc, Tpre, max_dist, acc); return; }
#endif
  micp_simd::body(unc, n_unc, Tpre, max_dist, acc);
}

// raw sums -> CrossStatistics (kernels.hip finalize_pose: IEEE divisions)
inline cstats micp_stats_from_sums(const double* acc) {
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

inline float micp_rho(const xform& T) { return 2.0f * sqrtf((T.R.x * T.R.x + T.R.y * T.R.y) + T.R.z * T.R.z); }
inline float micp_tau(const xform& T) { return sqrtf(dot_plain(T.t, T.t)); }

// is this set an exact summary for (pre-transform, max_dist)?
inline bool micp_set_covers(const MicpMomentSet& ms, const xform& Tpre, float max_dist) {
  return ms.valid && max_dist >= ms.gate_lo && max_dist <= ms.gate_hi && micp_rho(Tpre) <= ms.rho_cap && micp_tau(Tpre) <= ms.tau_cap;
}

// statistics_p2l(Tpre, dataset, model, max_dist) from a covering set
inline cstats micp_statistics_from_set(const MicpMomentSet& ms, const xform& Tpre, float max_dist) {
  double R[9], sums[16];
  micp_rotation_f64(Tpre.R, R);
  const double t[3] = {Tpre.t.x, Tpre.t.y, Tpre.t.z};
  micp_moment_sums(ms.mom, R, t, sums);
  if (ms.n_unc) micp_undecided_sums(ms.unc, ms.n_unc, Tpre, max_dist, sums);
  return micp_stats_from_sums(sums);
}

// host-side accumulation of a set from raw correspondences (the device's classification and products, in the same f64 products of
// f32 inputs): the CPU-testable twin of k_micp_moments + k_micp_publish (rmclhip_host_moment_statistics; tests/test_host_logic.py)
inline bool micp_set_from_correspondences(const float* D, const float* I, const float* N, const uint8_t* ok, uint32_t n, float gate_lo,
                                          float gate_hi, float rho_cap, float tau_cap, MicpMomentSet* ms, uint32_t* n_undecided) {
  std::memset(ms->mom, 0, sizeof(ms->mom));
  ms->gate_lo = gate_lo; ms->gate_hi = gate_hi; ms->rho_cap = rho_cap; ms->tau_cap = tau_cap;
  ms->n_unc = 0; ms->valid = false;
  uint32_t unc = 0;
  double* m = ms->mom;
  static thread_local float aos[kMicpHostMaxUnc][9];
  for (uint32_t i = 0; i < n; ++i) {
    if (ok && !ok[i]) continue;
    const f3 Di = mk3(D[3 * i], D[3 * i + 1], D[3 * i + 2]), Ii = mk3(I[3 * i], I[3 * i + 1], I[3 * i + 2]);
    const f3 Ni = mk3(N[3 * i], N[3 * i + 1], N[3 * i + 2]);
    const float spd0 = dot_plain(sub3(Ii, Di), Ni);
    const float nd = sqrtf(dot_plain(Di, Di));
    const int cls = micp_gate_class(spd0, nd, gate_lo, gate_hi, rho_cap, tau_cap);
    if (cls == 2) {
      if (unc < kMicpHostMaxUnc) {
        float* u = aos[unc];
        u[0] = Di.x; u[1] = Di.y; u[2] = Di.z; u[3] = Ii.x; u[4] = Ii.y; u[5] = Ii.z; u[6] = Ni.x; u[7] = Ni.y; u[8] = Ni.z;
      }
      ++unc;
    } else if (cls == 1) {
      const double Dd[3] = {Di.x, Di.y, Di.z}, Nd[3] = {Ni.x, Ni.y, Ni.z};
      const double sI = (Nd[0] * static_cast<double>(Ii.x) + Nd[1] * static_cast<double>(Ii.y)) + Nd[2] * static_cast<double>(Ii.z);
      const double DD[6] = {Dd[0] * Dd[0], Dd[0] * Dd[1], Dd[0] * Dd[2], Dd[1] * Dd[1], Dd[1] * Dd[2], Dd[2] * Dd[2]};
      const double NN[6] = {Nd[0] * Nd[0], Nd[0] * Nd[1], Nd[0] * Nd[2], Nd[1] * Nd[1], Nd[1] * Nd[2], Nd[2] * Nd[2]};
      m[0] += 1.0;
      for (int j = 0; j < 3; ++j) m[1 + j] += Dd[j];
      for (int k = 0; k < 6; ++k) m[4 + k] += DD[k];
      for (int a = 0; a < 3; ++a) {
        const double sn = sI * Nd[a];
        m[10 + a] += sn;
        for (int j = 0; j < 3; ++j) m[13 + 3 * a + j] += sn * Dd[j];
      }
      for (int pq = 0; pq < 6; ++pq) {
        m[22 + pq] += NN[pq];
        for (int j = 0; j < 3; ++j) m[28 + 3 * pq + j] += NN[pq] * Dd[j];
        for (int k = 0; k < 6; ++k) m[46 + 6 * pq + k] += NN[pq] * DD[k];
      }
    }
  }
  if (n_undecided) *n_undecided = unc;
  if (unc > kMicpHostMaxUnc) return false;
  ms->set_undecided(aos, unc);
  ms->valid = true;
  return true;
}

}  // namespace rmclhip
