// devmath.h -- rmagine-style POD math shared by host and device code.
//
// Operation order is part of the contract: the whole library is compiled with
// -ffp-contract=off and every fused multiply-add is an explicit fmaf(), so a value computed
// here is bit-identical on the host, on gfx950 and in the parity oracle.
// Semantics restate rmagine's Quaternion/Transform (external dependency of the reference,
// uos/rmagine >= 2.4.0): Hamilton product, q*p = q (p,0) q^-1, T*v = R v + t, ~T = (R^-1, -R^-1 t).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RM_HD __host__ __device__ __forceinline__
#else
#define RM_HD inline
#endif

namespace rmclhip {

struct f3 { float x, y, z; };
struct quat { float x, y, z, w; };
struct xform { quat R; f3 t; uint32_t stamp; };
static_assert(sizeof(xform) == 32, "Transform must be 32 B");

RM_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RM_HD f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RM_HD f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RM_HD f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RM_HD f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// rmagine Vector3::dot
RM_HD float dot_plain(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// Embree-style fused dot / cross (madd / msub chains)
RM_HD float dot_fma(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RM_HD f3 cross_fma(f3 a, f3 b) {
  return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

RM_HD quat qmul(quat a, quat b) {
  quat r;
  r.w = ((a.w * b.w - a.x * b.x) - a.y * b.y) - a.z * b.z;
  r.x = ((a.w * b.x + a.x * b.w) + a.y * b.z) - a.z * b.y;
  r.y = ((a.w * b.y - a.x * b.z) + a.y * b.w) + a.z * b.x;
  r.z = ((a.w * b.z + a.x * b.y) - a.y * b.x) + a.z * b.w;
  return r;
}
RM_HD quat qinv(quat q) { quat r; r.x = -q.x; r.y = -q.y; r.z = -q.z; r.w = q.w; return r; }
RM_HD f3 qrot(quat q, f3 p) {
  quat P; P.x = p.x; P.y = p.y; P.z = p.z; P.w = 0.0f;
  const quat PT = qmul(qmul(q, P), qinv(q));
  return mk3(PT.x, PT.y, PT.z);
}
RM_HD f3 xapply(const xform& T, f3 p) { return add3(qrot(T.R, p), T.t); }
RM_HD xform xmul(const xform& a, const xform& b) {
  xform r;
  r.t = add3(qrot(a.R, b.t), a.t);
  r.R = qmul(a.R, b.R);
  r.stamp = a.stamp;
  return r;
}
RM_HD xform xinv(const xform& a) {
  xform r;
  r.R = qinv(a.R);
  r.t = neg3(qrot(r.R, a.t));
  r.stamp = a.stamp;
  return r;
}
RM_HD xform xidentity() {
  xform r;
  r.R.x = 0.f; r.R.y = 0.f; r.R.z = 0.f; r.R.w = 1.f;
  r.t = mk3(0.f, 0.f, 0.f);
  r.stamp = 0;
  return r;
}

// rmagine Matrix3x3 <- Quaternion (row-major)
RM_HD void quat_to_mat(quat q, float* M) {
  M[0] = 2.0f * (q.w * q.w + q.x * q.x) - 1.0f; M[1] = 2.0f * (q.x * q.y - q.w * q.z); M[2] = 2.0f * (q.x * q.z + q.w * q.y);
  M[3] = 2.0f * (q.x * q.y + q.w * q.z); M[4] = 2.0f * (q.w * q.w + q.y * q.y) - 1.0f; M[5] = 2.0f * (q.y * q.z - q.w * q.x);
  M[6] = 2.0f * (q.x * q.z - q.w * q.y); M[7] = 2.0f * (q.y * q.z + q.w * q.x); M[8] = 2.0f * (q.w * q.w + q.z * q.z) - 1.0f;
}

// CrossStatistics (rmagine): covariance row-major, C(r,c) = 1/n sum (m-mm)_r (d-dm)_c
struct cstats {
  f3 dataset_mean;
  f3 model_mean;
  float covariance[9];
  uint32_t n_meas;
};
static_assert(sizeof(cstats) == 64, "CrossStatistics must be 64 B");

RM_HD cstats cs_identity() {
  cstats s;
  s.dataset_mean = mk3(0.f, 0.f, 0.f);
  s.model_mean = mk3(0.f, 0.f, 0.f);
  for (int i = 0; i < 9; ++i) s.covariance[i] = 0.f;
  s.n_meas = 0;
  return s;
}

// CrossStatistics::operator+= : count-weighted (Chan) merge
RM_HD cstats cs_merge(const cstats& a, const cstats& b) {
  cstats r;
  r.n_meas = a.n_meas + b.n_meas;
  if (r.n_meas == 0) return cs_identity();
  const float w1 = static_cast<float>(a.n_meas) / static_cast<float>(r.n_meas);
  const float w2 = static_cast<float>(b.n_meas) / static_cast<float>(r.n_meas);
  r.dataset_mean = add3(scale3(a.dataset_mean, w1), scale3(b.dataset_mean, w2));
  r.model_mean = add3(scale3(a.model_mean, w1), scale3(b.model_mean, w2));
  const f3 m1 = sub3(a.model_mean, r.model_mean), d1 = sub3(a.dataset_mean, r.dataset_mean);
  const f3 m2 = sub3(b.model_mean, r.model_mean), d2 = sub3(b.dataset_mean, r.dataset_mean);
  const float mm1[3] = {m1.x, m1.y, m1.z}, dd1[3] = {d1.x, d1.y, d1.z};
  const float mm2[3] = {m2.x, m2.y, m2.z}, dd2[3] = {d2.x, d2.y, d2.z};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const float P1 = a.covariance[3 * i + j] * w1 + b.covariance[3 * i + j] * w2;
      const float P2 = (mm1[i] * dd1[j]) * w1 + (mm2[i] * dd2[j]) * w2;
      r.covariance[3 * i + j] = P1 + P2;
    }
  return r;
}

// Transform * CrossStatistics
RM_HD cstats cs_transform(const xform& T, const cstats& s) {
  cstats r;
  r.dataset_mean = xapply(T, s.dataset_mean);
  r.model_mean = xapply(T, s.model_mean);
  float R[9], tmp[9];
  quat_to_mat(T.R, R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < 3; ++k) acc += R[3 * i + k] * s.covariance[3 * k + j];
      tmp[3 * i + j] = acc;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < 3; ++k) acc += tmp[3 * i + k] * R[3 * j + k];
      r.covariance[3 * i + j] = acc;
    }
  r.n_meas = s.n_meas;
  return r;
}

// ---- 3x3 SVD (Jacobi eigen-decomposition of A^T A, double) + Umeyama ----------------------
RM_HD void jacobi_eig3(const double* Ain, double* V, double* e) {
  double A[9];
  for (int i = 0; i < 9; ++i) { A[i] = Ain[i]; V[i] = 0.0; }
  V[0] = V[4] = V[8] = 1.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    // rounding leaves off ~ 1e-32 * dg after convergence: a tighter bound would never trigger and burn all sweeps
    if (off <= 1e-30 * dg || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  e[0] = A[0]; e[1] = A[4]; e[2] = A[8];
}

RM_HD double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// A = U diag(w) V^T, row-major; U, V orthonormal (completed for rank-deficient A)
RM_HD void svd3(const double* A, double* U, double* w, double* V) {
  double B[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += A[3 * k + i] * A[3 * k + j];
      B[3 * i + j] = acc;
    }
  double Vr[9], e[3];
  jacobi_eig3(B, Vr, e);
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (e[idx[j]] > e[idx[i]]) { const int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) V[3 * r + c] = Vr[3 * r + idx[c]];
    w[c] = e[idx[c]] > 0 ? sqrt(e[idx[c]]) : 0.0;
  }
  const double tol = 1e-9 * (w[0] > 0 ? w[0] : 1.0);
  double u[3][3];
  bool have[3] = {false, false, false};
  for (int c = 0; c < 3; ++c) {
    if (w[c] > tol) {
      for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += A[3 * r + k] * V[3 * k + c];
        u[c][r] = acc / w[c];
      }
      for (int p = 0; p < c; ++p)
        if (have[p]) {
          double d = 0;
          for (int r = 0; r < 3; ++r) d += u[c][r] * u[p][r];
          for (int r = 0; r < 3; ++r) u[c][r] -= d * u[p][r];
        }
      double nrm = 0;
      for (int r = 0; r < 3; ++r) nrm += u[c][r] * u[c][r];
      nrm = sqrt(nrm);
      if (nrm > 0) {
        for (int r = 0; r < 3; ++r) u[c][r] /= nrm;
        have[c] = true;
      }
    }
  }
  if (!have[0]) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; have[0] = true; }
  if (!have[1]) {
    int k = 0;
    if (fabs(u[0][1]) < fabs(u[0][k])) k = 1;
    if (fabs(u[0][2]) < fabs(u[0][k])) k = 2;
    double a[3] = {0, 0, 0};
    a[k] = 1;
    const double d = u[0][k];
    double nrm = 0;
    for (int r = 0; r < 3; ++r) { u[1][r] = a[r] - d * u[0][r]; nrm += u[1][r] * u[1][r]; }
    nrm = sqrt(nrm);
    for (int r = 0; r < 3; ++r) u[1][r] /= nrm;
    have[1] = true;
  }
  if (!have[2]) {
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  }
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) U[3 * r + c] = u[c][r];
}

RM_HD quat mat_to_quat(const double* R) {
  double q[4];  // x y z w (Shepperd)
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = sqrt(tr + 1.0) * 2.0;
    q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
    q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
    q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
  } else {
    const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
    q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
  }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  quat r;
  r.x = static_cast<float>(q[0] / n); r.y = static_cast<float>(q[1] / n);
  r.z = static_cast<float>(q[2] / n); r.w = static_cast<float>(q[3] / n);
  return r;
}


// Rotation of the Kabsch / Umeyama problem as a unit quaternion, without an SVD: the optimal proper rotation
// R (m ~ R d) is the eigenvector of the largest eigenvalue of Horn's symmetric, traceless 4x4 matrix K built from
// the cross-covariance (Horn 1987); that eigenvalue is the largest root of the quartic
// P(l) = l^4 + c2 l^2 + c1 l + c0 and is found by Newton's method from the upper bound sqrt(trace K^2) (monotone
// convergence; Theobald 2005, "QCP"); the eigenvector is a column of adj(K - l I).  The reflection case of the
// SVD formulation (det < 0 -> diag(1,1,-1)) needs no special handling: the eigenvector IS the best proper
// rotation.  ~350 dependent fp64 operations instead of ~900 for the scaled-Newton polar iteration used before (a
// lone lane retires one dependent instruction per ~5 cycles, so this is what the MICP step costs).  Returns false
// for (nearly) degenerate inputs -- repeated largest eigenvalue, zero matrix --, where the caller falls back to
// the Jacobi SVD, which defines the semantics.  q = (x, y, z, w), normalised.
// determinant and adjugate of a symmetric 4x4 matrix from its ten entries, by 2x2 sub-determinants (Laplace expansion);
// written with scalars only -- indexed local arrays would live in scratch memory on the device
struct sym4 {
  double k00, k01, k02, k03, k11, k12, k13, k22, k23, k33;
};
struct sym4_minors {
  double s0, s1, s2, s3, s4, s5, c0, c1, c2, c3, c4, c5;
};
RM_HD sym4_minors sym4_sub(const sym4& K) {
  sym4_minors m;
  // rows 0,1 (k10 = k01): s*; rows 2,3 (k20 = k02, k21 = k12, k30 = k03, k31 = k13, k32 = k23): c*
  m.s0 = K.k00 * K.k11 - K.k01 * K.k01; m.s1 = K.k00 * K.k12 - K.k01 * K.k02; m.s2 = K.k00 * K.k13 - K.k01 * K.k03;
  m.s3 = K.k01 * K.k12 - K.k11 * K.k02; m.s4 = K.k01 * K.k13 - K.k11 * K.k03; m.s5 = K.k02 * K.k13 - K.k12 * K.k03;
  m.c5 = K.k22 * K.k33 - K.k23 * K.k23; m.c4 = K.k12 * K.k33 - K.k13 * K.k23; m.c3 = K.k12 * K.k23 - K.k13 * K.k22;
  m.c2 = K.k02 * K.k33 - K.k03 * K.k23; m.c1 = K.k02 * K.k23 - K.k03 * K.k22; m.c0 = K.k02 * K.k13 - K.k03 * K.k12;
  return m;
}
RM_HD double sym4_det(const sym4_minors& m) {
  return m.s0 * m.c5 - m.s1 * m.c4 + m.s2 * m.c3 + m.s3 * m.c2 - m.s4 * m.c1 + m.s5 * m.c0;
}

RM_HD bool horn_quaternion(const double* C, double* q) {
  // C[3*r + c] = sum m_r d_c  =>  S_ab = sum d_a m_b = C[3*b + a]
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
  // largest eigenvalue of K = sum of the singular values of S <= sqrt(3) * |S|_F (Cauchy-Schwarz); starting Newton's
  // iteration at or above the largest root of the quartic keeps it monotone (P convex there), so this tighter bound
  // (the classic one is sqrt(trace K^2) = 2 |S|_F) saves iterations without changing the limit
  const double lam0 = sqrt(3.0 * ss) * (1.0 + 1e-12);
  double lam = lam0;
  {
    // A much closer start for the case this solver is called for (small corrections: S nearly symmetric, R nearly I):
    // k00 = trace S is the Rayleigh quotient of q = (1, 0, 0, 0), hence a LOWER bound of the largest root, O(theta^2) below
    // it.  Where the quartic is convex (3 l^2 > ss) and rising, one Newton step from a point left of the root lands right
    // of it (the tangent of a convex function lies below it), from where the iteration is monotone as before.
    const double a = K.k00, a2 = a * a;
    const double Pa = (a2 + c2) * a2 + c1 * a + c0, dPa = (4.0 * a2 + 2.0 * c2) * a + c1;
    if (a > 0.0 && 3.0 * a2 > ss && dPa > 0.0 && Pa <= 0.0) {
      const double a1 = a - Pa / dPa;
      if (a1 < lam0) lam = a1;
    }
  }
  bool converged = false;
  for (int it = 0; it < 60; ++it) {
    const double l2 = lam * lam;
    const double P = (l2 + c2) * l2 + c1 * lam + c0;
    const double dP = (4.0 * l2 + 2.0 * c2) * lam + c1;
    if (!(dP > 0.0)) break;
    const double step = P / dP;
    lam -= step;
    if (step <= 1e-14 * lam0) { converged = true; break; }
  }
  if (!converged) return false;
  K.k00 -= lam; K.k11 -= lam; K.k22 -= lam; K.k33 -= lam;
  const sym4_minors m = sym4_sub(K);
  // adj(K - l I) = const * v v^T (symmetric): diagonal entries ~ v_c^2, take the column of the largest one
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
  const double n = sqrt((v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3));
  if (!(n > 0.0)) return false;
  const double rn = 1.0 / n;
  double w = v0 * rn, x = v1 * rn, y = v2 * rn, z = v3 * rn;
  // sign convention of mat_to_quat (Shepperd): w > 0 when trace R > 0 (|w| > 1/2), else the largest of x, y, z positive
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

RM_HD xform umeyama(const cstats& s) {
  xform T = xidentity();
  if (s.n_meas == 0) return T;
  double C[9];
  for (int i = 0; i < 9; ++i) C[i] = static_cast<double>(s.covariance[i]);
  double q[4];
  if (horn_quaternion(C, q)) {
    T.R.x = static_cast<float>(q[0]); T.R.y = static_cast<float>(q[1]);
    T.R.z = static_cast<float>(q[2]); T.R.w = static_cast<float>(q[3]);
  } else {
    // Jacobi SVD: R = U diag(1, 1, sign(det U det V)) V^T (defines the semantics; degenerate inputs only)
    double U[9], w[3], V[9], R[9];
    svd3(C, U, w, V);
    double S[3] = {1, 1, 1};
    if (det3(U) * det3(V) < 0) S[2] = -1;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += U[3 * i + k] * S[k] * V[3 * j + k];
        R[3 * i + j] = acc;
      }
    T.R = mat_to_quat(R);
  }
  T.t = sub3(s.model_mean, qrot(T.R, s.dataset_mean));
  return T;
}


// ---- gate-stable moment form of the MICP iterations (kernels.hip k_micp_moments, find_kernel.hip.h find_moments_wave, and the
// host's evaluation in micp_host.h): classification of one correspondence at the identity pre-transform.  spd0 = (I - D) . N is
// the reduction's own gate value, nd = |D|.  While the pre-transform stays within (rho_cap = |2 sin(theta/2)|, tau_cap = |t|) the
// dataset point moves by at most rho_cap nd + tau_cap, and so does the gate value (|N| = 1; the 1e-4 (1 + nd) covers the f32
// rounding of both evaluations).  The gate |spd| < max_dist is asked for every max_dist in [gate_lo, gate_hi] (one value when the
// caller knows it; a band when the find speculates for the computeCrossStatistics calls that will follow it):
//   1 = gated in under every such pre-transform and max_dist  -> its contribution comes from the moments
//   0 = gated out under every one (also a NaN gate value)     -> contributes nothing
//   2 = undecided                                             -> re-evaluated per call with the reduction's own f32 arithmetic
RM_HD int micp_gate_class(float spd0, float nd, float gate_lo, float gate_hi, float rho_cap, float tau_cap) {
  const float margin = (rho_cap * nd + tau_cap) + 1e-4f * (1.0f + nd);
  const float a = fabsf(spd0);
  if (spd0 != spd0) return 0;
  if (gate_lo - a > margin) return 1;
  if (a - gate_hi > margin) return 0;
  return 2;
}

}  // namespace rmclhip
