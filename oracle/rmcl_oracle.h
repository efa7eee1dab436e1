/*
 * rmcl_oracle.h -- CPU restatement of RMCL / MICP-L's ray-casting-correspondence
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library.  The product (rmcl_amd/, librmclhip.so) never links,
 * imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (/root/reference, uos/rmcl v2.4.0) ships no
 * tests, no golden vectors and cannot be built here (its arithmetic lives in
 * the un-vendored dependency uos/rmagine `version: main` (>= 2.4.0, CMake
 * rmcl/CMakeLists.txt:62-73) and, below it, Embree 4).  This file restates
 * rmagine's / Embree's published algorithms and is anchored on the reference's
 * own call sites and in-repo restatements (cited per function).  See DESIGN.md.
 *
 * All arithmetic on the ray path is IEEE binary32 with an explicit operation
 * order (compiled with -ffp-contract=off; fused multiply-adds appear only where
 * written as fmaf()), so that the HIP kernels can reproduce it bit for bit.
 */
#ifndef RMCL_ORACLE_H
#define RMCL_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- POD types (layouts pinned by rmcl_ros/src/nodes/rmcl_localization.cpp:245-249,
 *      rmcl_ros/include/rmcl_ros/rmcl/ParticleAttributes.hpp:18-34,
 *      rmcl_ros/include/rmcl_ros/rmcl/RangeMeasurement.hpp:10-21) ---- */
typedef struct { float x, y, z; } orc_vec3;
typedef struct { float x, y, z, w; } orc_quat;
typedef struct { orc_quat R; orc_vec3 t; uint32_t stamp; } orc_transform;      /* 32 B */
typedef struct { float min, inc; uint32_t size; } orc_discrete_interval;
typedef struct { float min, max; } orc_interval;
typedef struct {
  orc_discrete_interval phi;   /* vertical   (rows,  height) */
  orc_discrete_interval theta; /* horizontal (cols,  width)  */
  orc_interval range;
} orc_spherical_model;
typedef struct {
  orc_vec3 dataset_mean;
  orc_vec3 model_mean;
  float covariance[9];         /* row-major C(r,c) = sum (m-mm)_r (d-dm)_c / n */
  uint32_t n_meas;
} orc_cross_statistics;                                                       /* 64 B */
typedef struct { float mean, sigma; uint32_t n_meas; } orc_gaussian1d;
typedef struct { orc_gaussian1d likelihood; float state_sigma[6]; } orc_particle_attributes; /* 36 B */
typedef struct { orc_vec3 orig, dir; float range; float cov[9]; } orc_range_measurement;    /* 64 B */
typedef struct {
  float dist_sigma;
  float real_hit_sim_miss_error;
  float real_miss_sim_hit_error;
  float real_miss_sim_miss_error;
  orc_interval sensor_range;
  uint32_t max_n_meas;          /* 10000, ParticleAttributes.hpp:34 */
  uint32_t correspondence_type; /* 0 = evaluate_rcc (unit normals), 1 = evaluate_cpc (PCDSensorUpdaterEmbree.cpp:211-222), 2 = evaluate_rcc with Embree's raw Ng, 3 = the OptiX program's rules (tmax 1e4, no range.min test) */
} orc_pf_params;

typedef struct {
  uint64_t nodes_visited;       /* BVH2 node visits (each 32 B in the reference tree) */
  uint64_t tris_tested;
  uint64_t rays;
} orc_counters;

typedef struct orc_mesh orc_mesh;

/* ---- transform algebra (rmagine Quaternion/Transform, used throughout the reference,
 *      e.g. micp_localization.cpp:926,963; MICPSensor.hpp:178) ---- */
orc_quat orc_quat_mult(orc_quat a, orc_quat b);
orc_quat orc_quat_inv(orc_quat q);
orc_vec3 orc_quat_rotate(orc_quat q, orc_vec3 p);
orc_transform orc_transform_mult(orc_transform a, orc_transform b);
orc_transform orc_transform_inv(orc_transform a);
orc_vec3 orc_transform_apply(orc_transform T, orc_vec3 p);
orc_quat orc_euler_to_quat(float roll, float pitch, float yaw);
orc_transform orc_transform_identity(void);

/* ---- mesh + intersector ---- */
orc_mesh* orc_mesh_create(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf,
                          uint32_t max_leaf);
void orc_mesh_destroy(orc_mesh* m);
uint32_t orc_mesh_num_nodes(const orc_mesh* m);
/* face normal exactly as the product must compute it */
void orc_mesh_face_normals(const orc_mesh* m, float* out_nf3);

/* closest hit, tie-break (min t, then min face id). returns 1 on hit */
int orc_intersect_brute(const orc_mesh* m, orc_vec3 O, orc_vec3 D, float tnear, float tfar,
                        float* t_out, uint32_t* face_out);
int orc_intersect_bvh(const orc_mesh* m, orc_vec3 O, orc_vec3 D, float tnear, float tfar,
                      float* t_out, uint32_t* face_out, orc_counters* cnt);

/* walk the PRODUCT's exported BVH4 arrays (rmcl_amd/csrc/layout.h) with the oracle's intersector */
int orc_trace_bvh4(const uint32_t* nodes, uint32_t n_nodes, const uint32_t* tris, uint32_t n_tris,
                   orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out, uint32_t* face_out);
/* directions of a spherical model, buffer order, 3 floats each */
void orc_spherical_directions(const orc_spherical_model* model, float* out);
/* oracle-side triangle records (face-id order, 15 floats: v0 e1 e2 Ng n) */
void orc_mesh_tri_records(const orc_mesh* m, float* out);

/* ---- simulate (rm::SphereSimulatorEmbree::simulate / O1DnSimulatorEmbree::simulate,
 *      call sites rmcl/src/rmcl/registration/RCCEmbree.cpp:35,98) ----
 * outputs are nullable; all in the SENSOR frame (CPCEmbree.cpp:27-41). */
int orc_simulate_spherical(const orc_mesh* m, const orc_spherical_model* model,
                           const orc_transform* Tsb, const orc_transform* Tbm, uint32_t nposes,
                           int use_bvh, int nthreads,
                           uint8_t* hits, float* ranges, float* points, float* normals,
                           uint32_t* face_ids, orc_counters* cnt);
int orc_simulate_o1dn(const orc_mesh* m, uint32_t width, uint32_t height, orc_interval range,
                      orc_vec3 orig, const float* dirs,
                      const orc_transform* Tsb, const orc_transform* Tbm, uint32_t nposes,
                      int use_bvh, int nthreads,
                      uint8_t* hits, float* ranges, float* points, float* normals,
                      uint32_t* face_ids, orc_counters* cnt);

/* RCCEmbreePinhole / RCCEmbreeOnDn (RCCEmbree.cpp:39-68,102-130): same intersector, other ray generators */
orc_vec3 orc_pinhole_direction(const float* f, const float* c, uint32_t vid, uint32_t hid);
void orc_pinhole_directions(uint32_t width, uint32_t height, const float* f, const float* c, float* out);
int orc_simulate_pinhole(const orc_mesh* m, uint32_t width, uint32_t height, orc_interval range, const float* f,
                         const float* c, const orc_transform* Tsb, const orc_transform* Tbm, uint32_t nposes,
                         int use_bvh, int nthreads, uint8_t* hits, float* ranges, float* points, float* normals,
                         uint32_t* face_ids, orc_counters* cnt);
int orc_simulate_ondn(const orc_mesh* m, uint32_t width, uint32_t height, orc_interval range, const float* origs,
                      const float* dirs, const orc_transform* Tsb, const orc_transform* Tbm, uint32_t nposes,
                      int use_bvh, int nthreads, uint8_t* hits, float* ranges, float* points, float* normals,
                      uint32_t* face_ids, orc_counters* cnt);

/* ---- rm::statistics_p2l (CorrespondencesCPU.cpp:26-30; gate pinned by MICPSensorCPU.cpp:70-84) ---- */
void orc_statistics_p2l_f32(const orc_transform* Tpre,
                            const float* dataset_points, const uint8_t* dataset_mask,
                            const float* model_points, const float* model_normals,
                            const uint8_t* model_mask, uint32_t n, float max_dist,
                            orc_cross_statistics* out);
/* double two-pass version: out fields as doubles [3 + 3 + 9] + count */
void orc_statistics_p2l_f64(const orc_transform* Tpre,
                            const float* dataset_points, const uint8_t* dataset_mask,
                            const float* model_points, const float* model_normals,
                            const uint8_t* model_mask, uint32_t n, float max_dist,
                            double* out15, uint32_t* n_out);
/* the same statistics as ONE pass of raw double sums on `nthreads` workers: bench.py's cpu_baseline form (a tuned CPU reduction), never the checker's */
void orc_statistics_p2l_fast(const orc_transform* Tpre, const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                             const float* model_normals, const uint8_t* model_mask, uint32_t n, float max_dist, int nthreads,
                             orc_cross_statistics* out);
/* CorrespondencesCPU.cpp:21-23 */
float orc_adaptive_max_dist(float max_dist, float adaptive_max_dist_min, double convergence_progress);

/* rm::CrossStatistics algebra (micp_localization.cpp:918-937, MICPSensor.hpp:182) */
orc_cross_statistics orc_cross_statistics_identity(void);
orc_cross_statistics orc_cross_statistics_merge(orc_cross_statistics a, orc_cross_statistics b);
orc_cross_statistics orc_cross_statistics_transform(orc_transform T, orc_cross_statistics s);
/* rm::umeyama_transform (micp_localization.cpp:952-953) */
orc_transform orc_umeyama_transform(const orc_cross_statistics* s);
/* 3x3 SVD (one-sided Jacobi, double): A = U diag(w) V^T, row-major */
void orc_svd3(const double* A, double* U, double* w, double* V);

/* ---- particle filter beam evaluation (PCDSensorUpdaterEmbree.cpp:18-86,197-241,290-342) ---- */
float orc_evaluate_rcc(const orc_mesh* m, const orc_range_measurement* meas_m,
                       const orc_pf_params* p, int use_bvh);
orc_gaussian1d orc_gaussian1d_add(orc_gaussian1d a, orc_gaussian1d b);
int orc_pf_update(const orc_mesh* m, const orc_transform* poses, orc_particle_attributes* attrs,
                  uint32_t n, const orc_range_measurement* beams, uint32_t nbeams,
                  const orc_transform* Tsb, const orc_pf_params* p, int use_bvh, int nthreads,
                  float* errors_out /* nullable, n*nbeams */);

/* ---- closest-point correspondences (CPCEmbree.cpp:18-44; rm::EmbreeMap::closestPoint) ---- */
/* use_bvh = 2 (accepted by orc_simulate_spherical / _o1dn / ...): the closest hit through a 4-wide collapse of the BVH2 with SSE slab
 * tests -- the CPU-baseline path of bench.py; bit-identical results (same triangle test, same tie-break) */
int orc_intersect_bvh4(const orc_mesh* m, orc_vec3 O, orc_vec3 D, float tnear, float tfar, float* t_out, uint32_t* face_out);
int orc_closest_point(const orc_mesh* m, orc_vec3 P, int use_bvh, float* d_out, orc_vec3* cp_out, uint32_t* face_out);
void orc_cpc_find(const orc_mesh* m, const orc_transform* Tsb, const orc_transform* Tbm, const float* dataset_points,
                  uint32_t n, float max_dist, int use_bvh, uint8_t* hits, float* dists, float* points, float* normals,
                  uint32_t* face_ids);

/* ---- particle-filter motion update (TFMotionUpdaterCPU.cpp:17-50,184-224; particle_motion.cu:11-34) ---- */
int orc_collision_in_between(const orc_mesh* m, orc_vec3 p1, orc_vec3 p2, int use_bvh);
void orc_pf_motion_update(const orc_mesh* m, orc_transform* poses, orc_particle_attributes* attrs, uint32_t n,
                          const orc_transform* T_bnew_bold, double forget_rate, uint32_t max_n_meas, int use_bvh);

/* ---- gladiator resampling (resampling.cu:41-219, GladiatorResamplerCPU.cpp:71-176) ----
 * The reference draws from cuRAND (GPU) / per-TBB-thread mt19937 (CPU, schedule dependent): neither stream can
 * be reproduced, so the restatement pins a counter-based generator instead: Philox4x32-10 (Salmon et al. 2011),
 * key = seed, counter = (champion index, step, draw, 0).  Normals: Box-Muller in double.  Every transcendental
 * (log, sin, cos, atan2, asin, pow) is evaluated in double and rounded to float, so that host and device agree. */
typedef struct {
  float min_noise_tx, min_noise_ty, min_noise_tz;         /* GladiatorResamplerGPU.cpp:34-40 defaults .03 .03 0 */
  float min_noise_roll, min_noise_pitch, min_noise_yaw;   /* 0 0 .01 */
  float likelihood_forget_per_meter;                      /* 0.3 */
  float likelihood_forget_per_radian;                     /* 0.2 */
  uint32_t trans_dist_metric;   /* 0 = l2norm (resampling.cu:179), 1 = l2normSquared (GladiatorResamplerCPU.cpp:156) */
} orc_gladiator_config;
typedef struct { float sum, max; } orc_likelihood_stats;
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
orc_likelihood_stats orc_likelihood_stats_compute(const orc_particle_attributes* attrs, uint32_t n);
void orc_quat_to_euler(orc_quat q, float* roll, float* pitch, float* yaw);
/* champions first .. first+count-1 fight a random enemy out of the n particles; the winners land in
 * poses_new / attrs_new [0 .. count) */
void orc_gladiator_resample(const orc_transform* poses, const orc_particle_attributes* attrs, uint32_t n,
                            orc_transform* poses_new, orc_particle_attributes* attrs_new, uint32_t first,
                            uint32_t count, const orc_gladiator_config* cfg, uint64_t seed, uint32_t step);

/* ---- residual resampling (ResidualResamplerCPU.cpp:55-203): see the restatement's header comment for what is pinned.
 * cfg: the gladiator's struct (same parameters; trans_dist_metric is ignored -- the residual resampler uses |dt|^2). */
uint32_t orc_residual_resample(const orc_transform* poses, const orc_particle_attributes* attrs, uint32_t n,
                               orc_transform* poses_new, orc_particle_attributes* attrs_new, uint32_t n_new,
                               const orc_gladiator_config* cfg, uint64_t seed, uint32_t step, uint64_t max_draws,
                               uint64_t* n_draws);

/* instrumented walk of the PRODUCT's BVH4 in the product's traversal order (nearest child first, deferred
 * children on a stack): counts inner-node visits, visits whose four children all fail, leaf visits, triangle
 * tests and the stack high-water mark.  mode bit0: cull popped entries with their stored entry distance;
 * bit1: fully sort the deferred children.  BVH-quality / traversal-policy measurements only. */
int orc_trace_bvh4_ordered(const uint32_t* nodes, const uint32_t* tris, orc_vec3 O, orc_vec3 D, float tnear, float tfar,
                           int mode, uint64_t counters[5], float* t_out, uint32_t* face_out);

/* ---- PointCloud2 wire format -> O1Dn model + dataset --------------------------------------------------
 * estimateModelAndData (rmcl_ros/src/util/conversions.cpp:869-1002: per point range = |p|, dir = p / range;
 * non-finite -> dir 0, range 0) + filter (scan_operations.cpp:41-116: skip_begin / skip_end / increment
 * sub-sampling of rows and columns) + MICPO1DnSensorCPU::unpackMessage (MICPO1DnSensorCPU.cpp:176-227:
 * point = dir * range + orig, mask = range within [range_min, range_max]).  datatype: 7 = FLOAT32, 8 = FLOAT64
 * (sensor_msgs/PointField).  Outputs have out_w * out_h entries. */
typedef struct { uint32_t skip_begin, skip_end, increment; } orc_filter1d;
int orc_pointcloud2_unpack(const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                           uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t datatype, orc_filter1d fh,
                           orc_filter1d fw, float range_min, float range_max, uint32_t* out_w, uint32_t* out_h,
                           float* dirs, float* ranges, float* points, uint8_t* mask, uint32_t* n_valid);

#ifdef __cplusplus
}
/* beam sampling (PCDSensorUpdaterEmbree.cpp:276-327): MT19937(seed), index = draw % (width * height), up to 100 retries
 * for a point without NaN; returns 0, *n_out = beams written */
int orc_sample_beams_pointcloud2(const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                                 uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t datatype, uint32_t samples,
                                 uint32_t seed, orc_range_measurement* out, uint32_t* n_out);
uint32_t orc_mt19937_draw(uint32_t seed, uint32_t n_skip);   /* the (n_skip+1)-th output of MT19937(seed): known-answer tests */


#endif
#endif
