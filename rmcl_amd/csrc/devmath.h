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


// reciprocal for the Newton iteration below: on the device one v_rcp_f64 plus two Newton-Raphson steps (full
// double accuracy, ~5 dependent instructions instead of the ~35 of an IEEE division); exact division on the host
RM_HD double polar_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

// Rotation factor of the polar decomposition A = R*H by the scaled Newton iteration X <- (g X + X^-T / g) / 2
// (Higham).  Only valid for det A > 0 (a proper rotation is wanted); returns false for reflections, rank
// deficiency or slow convergence -- the caller falls back to the Jacobi SVD.  The iteration runs on ONE lane, so
// its cost is the length of the dependent fp64 chain: the scaling factor g (any positive value works, it only
// steers convergence) is computed in fp32 and snapped to exactly 1 near convergence (so that the fixed point is
// exactly orthogonal), the inverse uses polar_rcp, and the loop stops when the step is below 1e-7: Newton
// converges quadratically, the iterate just computed is then accurate to ~1e-14.
RM_HD bool polar3(const double* A, double* R) {
  double X[9];
  double fro2 = 0.0;
  for (int i = 0; i < 9; ++i) { X[i] = A[i]; fro2 += A[i] * A[i]; }
  const double fro = sqrt(fro2);
  if (!(fro > 0.0) || !(det3(A) > 1e-9 * fro * fro * fro)) return false;
  for (int it = 0; it < 24; ++it) {
    // Y = X^-T via cofactors: inv(X) = adj(X) / det  =>  X^-T = cof(X) / det
    double Cf[9];
    Cf[0] = X[4] * X[8] - X[5] * X[7]; Cf[1] = X[5] * X[6] - X[3] * X[8]; Cf[2] = X[3] * X[7] - X[4] * X[6];
    Cf[3] = X[2] * X[7] - X[1] * X[8]; Cf[4] = X[0] * X[8] - X[2] * X[6]; Cf[5] = X[1] * X[6] - X[0] * X[7];
    Cf[6] = X[1] * X[5] - X[2] * X[4]; Cf[7] = X[2] * X[3] - X[0] * X[5]; Cf[8] = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * Cf[0] + X[1] * Cf[1] + X[2] * Cf[2];
    if (!(det > 0.0)) return false;
    const double idet = polar_rcp(det);
    double nx = 0.0, ny = 0.0;
    for (int i = 0; i < 9; ++i) { Cf[i] *= idet; nx += X[i] * X[i]; ny += Cf[i] * Cf[i]; }
    float gf = sqrtf(sqrtf(static_cast<float>(ny) / static_cast<float>(nx)));  // (|X^-1|_F / |X|_F)^(1/2)
    if (!(gf > 0.0f) || !(gf < 3.0e38f) || fabsf(gf - 1.0f) < 1.0e-3f) gf = 1.0f;
    const double a = 0.5 * static_cast<double>(gf), b = static_cast<double>(0.5f / gf);
    double diff = 0.0;
    for (int i = 0; i < 9; ++i) {
      const double xn = a * X[i] + b * Cf[i];
      const double d = xn - X[i];
      diff += d * d;
      X[i] = xn;
    }
    if (diff <= 1e-14 * 3.0) {  // |X_{k+1} - X_k|_F <= 1e-7 |R|_F  =>  |X_{k+1} - R| ~ 1e-14
      for (int i = 0; i < 9; ++i) R[i] = X[i];
      return true;
    }
  }
  return false;
}

// rm::umeyama_transform: C = U S V^T, R = U diag(1,1,sign(det U det V)) V^T, t = mm - R dm
RM_HD xform umeyama(const cstats& s) {
  xform T = xidentity();
  if (s.n_meas == 0) return T;
  double C[9], R[9];
  for (int i = 0; i < 9; ++i) C[i] = static_cast<double>(s.covariance[i]);
  // Fast path: for det(C) > 0 the Kabsch/Umeyama rotation U diag(1,1,+1) V^T IS the orthogonal polar factor of
  // C, which a scaled Newton iteration X <- (g X + X^-T / g) / 2 delivers in ~6 steps of ~60 fp64 operations --
  // an order of magnitude fewer dependent fp64 instructions than the Jacobi SVD (a lone lane issues one fp64
  // instruction per ~8 cycles: the SVD solve measured ~9 us of the 14 us k_micp_step).  Reflection (det < 0),
  // rank-deficient or slowly converging inputs take the SVD path below, which defines the semantics.
  if (!polar3(C, R)) {
    double U[9], w[3], V[9];
    svd3(C, U, w, V);
    double S[3] = {1, 1, 1};
    if (det3(U) * det3(V) < 0) S[2] = -1;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += U[3 * i + k] * S[k] * V[3 * j + k];
        R[3 * i + j] = acc;
      }
  }
  T.R = mat_to_quat(R);
  T.t = sub3(s.model_mean, qrot(T.R, s.dataset_mean));
  return T;
}

}  // namespace rmclhip
