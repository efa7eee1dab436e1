// pf_common.hip.h -- particle-filter pieces shared by the production kernels (kernels.hip: k_pf_update_v3, the closest-point
// form k_pf_update<64, 3>) and the experiments (kernels_lab.hip: the round kernels and the round-2 persistent kernel).
#pragma once
#include "traverse.hip.h"

namespace rmclhip {
namespace {

// ---------------------------------------------------------------------------------------------
// particle filter: all beams of all particles in one launch
// ---------------------------------------------------------------------------------------------
struct g1d { float mean, sigma; uint32_t n_meas; };
struct pattrs { g1d likelihood; float state_sigma[6]; };
static_assert(sizeof(pattrs) == 36, "ParticleAttributes must be 36 B");

// rm::Gaussian1D::operator+= (1-D count-weighted merge)
__device__ __forceinline__ g1d g1d_add(g1d a, g1d b) {
  g1d r;
  r.n_meas = a.n_meas + b.n_meas;
  const float w1 = static_cast<float>(a.n_meas) / static_cast<float>(r.n_meas);
  const float w2 = static_cast<float>(b.n_meas) / static_cast<float>(r.n_meas);
  r.mean = a.mean * w1 + b.mean * w2;
  const float P1 = a.sigma * w1 + b.sigma * w2;
  const float P2 = ((a.mean - r.mean) * (a.mean - r.mean)) * w1 + ((b.mean - r.mean) * (b.mean - r.mean)) * w2;
  r.sigma = P1 + P2;
  return r;
}

// Normal the point-to-plane error of evaluate_rcc is taken against: the unit face normal (OptiX program,
// BeamEvaluateProgram.cu:104-113; dwords 12..14 of the record) or, for correspondence_type 2, Embree's un-normalised
// rayhit.hit.Ng = cross(e2, e1) that the Embree updater reads (PCDSensorUpdaterEmbree.cpp:56-66; dwords 9..11).
__device__ __forceinline__ f3 pf_error_normal(const uint32_t* tris, uint32_t rec, uint32_t raw_ng) {
  const uint4 r = reinterpret_cast<const uint4*>(tris)[static_cast<size_t>(rec) * 4u + (raw_ng ? 2u : 3u)];
  return raw_ng ? mk3(asf(r.y), asf(r.z), asf(r.w)) : mk3(asf(r.x), asf(r.y), asf(r.z));
}

// kTrav: 0 = while-while traversal, per-lane stack 16 entries in LDS + scratch overflow (default)
//        1 = while-while traversal, per-lane stack entirely in LDS
//        2 = original single-loop traversal, stack in LDS (A/B)
//        3 = closest-point correspondences (correspondence_type 1): nearest-point query instead of a ray
template <int kStackDepth, int kTrav>
__global__ void __launch_bounds__(256) k_pf_update(const PfParams p) {
  // LDS: [ per-lane stacks kStackDepth*256 (kTrav != 0) | Tsm (PB xforms) | evals (PB*n_beams floats) ]
  extern __shared__ uint32_t lds_dyn[];
  uint32_t* stacks = lds_dyn;
  xform* s_Tsm = reinterpret_cast<xform*>(lds_dyn + ((kTrav == 0 || kTrav == 3) ? 16 : kStackDepth) * 256);
  float* s_eval = reinterpret_cast<float*>(s_Tsm + p.particles_per_block);

  const uint32_t PB = p.particles_per_block;
  const uint32_t p0 = blockIdx.x * PB;
  if (p0 >= p.n_particles) return;
  const uint32_t np = min(PB, p.n_particles - p0);
  if (threadIdx.x < np) s_Tsm[threadIdx.x] = xmul(p.poses[p0 + threadIdx.x], p.Tsb);
  __syncthreads();

  const float sq = p.dist_sigma * p.dist_sigma;
  const uint32_t nrays = np * p.n_beams;
  for (uint32_t r = threadIdx.x; r < ((nrays + 255u) & ~255u); r += 256u) {
    const bool live = r < nrays;
    const uint32_t rr = live ? r : 0u;
    const uint32_t pi = rr / p.n_beams, b = rr - pi * p.n_beams;
    const xform Tsm = s_Tsm[pi];
    const float* bm = p.beams + 16u * b;
    // meas_m = Tsm * meas_s (RangeMeasurement.hpp:28-42)
    const f3 dir = qrot(Tsm.R, mk3(bm[3], bm[4], bm[5]));
    const f3 org = xapply(Tsm, mk3(bm[0], bm[1], bm[2]));
    const float range = bm[6];
    const bool finite = (dir.x == dir.x) && (dir.y == dir.y) && (dir.z == dir.z);
    if (kTrav == 3) {
      // evaluate_cpc (PCDSensorUpdaterEmbree.cpp:88-95): distance of meas_m.mean() = orig + dir * range to the surface
      const f3 mean = add3(org, scale3(dir, range));
      const bool ok = (mean.x == mean.x) && (mean.y == mean.y) && (mean.z == mean.z);
      NearHit nh;
      if (p.near_grid != nullptr) {
        // round 4: the query starts from the record of the map's near-grid cell the point falls into -- an actual candidate, so the
        // result is unchanged; beam end points are often metres from any surface, where an unseeded query cannot prune
        const uint32_t sr = (live && ok) ? near_grid_record(p.near_grid, p.gn, p.gorg, p.ginv, mean) : kNone;
        const NearHit seed = near_seed_from_record(p.tris, sr, p.n_tris, mean, live && ok);
        nearest_lane_ww<16>(p.nodes, p.tris, mean, live && ok, stacks + threadIdx.x, 256u, nh, &seed);
      } else {
        nearest_lane_ww<16>(p.nodes, p.tris, mean, live && ok, stacks + threadIdx.x, 256u, nh);
      }
      if (live) {
        const float error = (nh.face != kInvalidFace) ? sqrtf(nh.d2) : __uint_as_float(0x7FC00000u);
        if (p.errors) p.errors[static_cast<size_t>(p0 + pi) * p.n_beams + b] = error;
        const float arg = -(error * error) / sq / 2;
        s_eval[rr] = static_cast<float>(exp(static_cast<double>(arg)) /
                                        sqrt(static_cast<double>(2 * sq) * 3.14159265358979323846));
      }
      continue;
    }
    RayHit h;
    const float rtf = (live && finite) ? p.ray_tfar : -1.0f;
    if (kTrav == 0) trace_lane_bf<16>(p.nodes, p.tris, org, dir, rtf, stacks + threadIdx.x, h);
    else if (kTrav == 1) trace_lane_ww<64>(p.nodes, p.tris, org, dir, rtf, stacks + threadIdx.x, 256u, h);
    else trace_lane(p.nodes, p.tris, org, dir, rtf, stacks + threadIdx.x, 256u, h);
    if (live) {
      // evaluate_rcc (PCDSensorUpdaterEmbree.cpp:18-86) with unit face normals (BeamEvaluateProgram.cu:104-113)
      const bool real_hit = (range >= p.range_min) && (range <= p.range_max);
      const bool sim_hit = (h.rec != kNone) && (!p.sim_min_range || h.t > p.range_min);
      float error;
      if (sim_hit) {
        if (real_hit) {
          const f3 n = pf_error_normal(p.tris, h.rec, p.raw_ng);
          const f3 preal = add3(org, scale3(dir, range));
          const f3 pint = add3(org, scale3(dir, h.t));
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
      const float eval = static_cast<float>(exp(static_cast<double>(arg)) /
                                            sqrt(static_cast<double>(2 * sq) * 3.14159265358979323846));
      s_eval[rr] = eval;
    }
  }
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

constexpr int kPfRows = 20;  // LDS stack rows per lane of the persistent particle-filter kernels (sentinel included)

}  // namespace
}  // namespace rmclhip
