// kernels.h -- launch interface between the C ABI (capi_*.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "devmath.h"
#include "layout.h"
#include "micp_host.h"

namespace rmclhip {

// what a launcher returns for a kernel variant that lives in librmclhip_lab.so while that library is not loaded: a value of its own,
// NOT hipErrorNotSupported -- the runtime reports that for real failures (graph capture, host-allocation flags), which must stay
// visible as what they are
constexpr hipError_t kLabMissing = static_cast<hipError_t>(2040);

enum ModelKind : uint32_t { kModelNone = 0, kModelSpherical = 1, kModelO1Dn = 2, kModelPinhole = 3, kModelOnDn = 4 };

// one find() launch: every ray of (nposes x H x W)
struct FindParams {
  const uint32_t* nodes;   // Node4[]
  const uint32_t* qnodes;  // Node4Q[] (quantised twins)
  const uint32_t* cnodes;  // Node4C[] (child-major twins)
  const uint32_t* cnodes16;  // Node16C[] (a node's grandchildren, child-major; kind 32's two-levels-per-pass descent), nullable
  const uint32_t* tris;    // TriRec[]
  uint32_t n_nodes;        // number of Node4 (the LDS-resident top of the tree copies min(kTop, n_nodes) of them)
  // spherical: [cos(phi_v) (H) | sin(phi_v) (H) | cos(theta_h) (W) | sin(theta_h) (W)], host libm values
  // o1dn:      dirs xyz (W*H*3)
  // ondn:      [origs xyz (W*H*3) | dirs xyz (W*H*3)]
  // pinhole:   none (procedural from pin_f = {fx, fy}, pin_c = {cx, cy})
  const float* model_tab;
  float pin_f[2], pin_c[2];
  uint32_t W, H;
  uint32_t tile_w_log2;    // wave = 2^tile_w_log2 x (64 >> tile_w_log2) rays of the scan image
  uint32_t tiles_x, tiles_y;
  float tfar;              // model.range.max
  f3 orig_s;               // sensor-frame ray origin (0 for spherical)
  // pose(s): Tsm = Tbm * Tsb. Single pose: by value; batch: device arrays (pose = blockIdx.y)
  xform Tsm, Tms;
  const xform* Tsm_arr;
  const xform* Tms_arr;
  uint32_t nposes;
  // outputs (nullable), sensor frame, index = pose*W*H + vid*W + hid
  uint8_t* hits;
  float* ranges;
  float* points;
  float* normals;
  uint32_t* face_ids;
  // frontier start (kinds 23 / 24): the map's frontier table and what bounds a hit's distance from any origin
  const uint32_t* frontier;      // n_frontier x 8 dwords {lo.xyz hi.x | hi.yz ref pad} (layout.h kFrontierDepth)
  uint32_t n_frontier;
  f3 scene_center;
  float scene_half_diag;
  const float* tile_planes;      // per tile of the scan image: the pyramid of its rays in the sensor frame (k_tile_planes), or null
  uint32_t frontier_max_preload; // stack entries the frontier start may leave per lane: 64 - stack_need of the tree the kind walks
  // kind 32 (traverse.hip.h frontier_descent_start): the wave stops descending when a level would leave more than descent_final_cap
  // entries (<= 64) or after descent_levels levels
  uint32_t descent_final_cap, descent_levels;
  // pose batches in world order (kernels.hip launch_batch_tile_order; nullable = pose-major): entry s = pose << 16 | (first tile / tiles per workgroup) of the s-th workgroup,
  // sorted by where the tile's central ray leaves the map's bounding box -- the launch is then ONE row of blocks (gridDim.y == 1)
  const uint32_t* tile_order;
  uint32_t n_tile_order;
  uint32_t tile_order_granule;
  uint32_t xcd_mapping;    // single scans: 0 = workgroup b computes tiles 4b.. (every XCD sees all of the image), 1 = an eighth of the image per XCD, 2 = a CU's two workgroups from the image's two halves (find_kernel.hip.h)   // workgroups of consecutive slots one XCD takes before the next XCD's turn (load balance across the XCDs)
  // diagnostics (nullable): per physical wave {s_memtime at entry, at exit (low 32 bits), s_memrealtime at entry, tile | xcc << 24}
  uint32_t* wave_clock;
  // MICP moment epilogue (launch_find_moments, k_find<..., kMom = true>): the moments of the gate-stable form (kernels.hip) are
  // formed while the correspondences are still in registers -- classification as in k_micp_moments, the 10 x 10 products
  // X^T Y of the factor vectors through v_mfma_f64_16x16x4_f64 -- instead of a second pass over the find's outputs
  const float* mom_dataset_points;
  const uint8_t* mom_dataset_mask;     // nullable
  uint32_t mom_n;                      // correspondences: min(n_dataset, W * H)
  float mom_gate_lo, mom_gate_hi, mom_rho_cap, mom_tau_cap;   // devmath.h micp_gate_class
  double* mom_partials;                // [gridDim.x][kMicpFastMoments]
  unsigned long long* mom_unc_mask;    // [4 * gridDim.x]: bit l of word t = lane l of (virtual) tile t holds an undecided correspondence
};

struct MicpState;
struct MicpCall;
enum TailMode : uint32_t { kTailNone = 0, kTailStats = 1, kTailMicp = 2, kTailBatchSolve = 3 };

struct ReduceParams {
  const float* dataset_points;
  const uint8_t* dataset_mask;   // nullable
  const float* model_points;
  const float* model_normals;
  const uint8_t* model_mask;     // nullable (k_reduce_partials only: a caller-owned view without a mask)
  uint32_t n;                    // elements per pose
  uint32_t nposes;               // model buffers hold nposes*n elements; dataset is shared
  float max_dist;
  const MicpCall* call;          // nullable: when set, max_dist / Tsb / Tbo are read from it (graph replay)
  xform Tpre;                    // used when Tpre_dev == nullptr
  const xform* Tpre_dev;         // per-pose pre-transform (device), nullable
  double* partials;              // [nposes][nblocks][16]
  uint32_t nblocks;
  // fused tail executed by the last block of each pose (kTailNone: partials only)
  uint32_t tail_mode;
  uint32_t* tickets;             // [nposes], zero before the first launch; re-armed by the kernel
  cstats* stats_out;             // kTailStats / kTailBatchSolve (nullable there)
  xform Tsb, Tbo;                // kTailMicp / kTailBatchSolve
  MicpState* state;              // kTailMicp
  xform* Tdelta_out;             // kTailBatchSolve
};

struct PfParams {
  const uint32_t* nodes;
  const uint32_t* qnodes;         // quantised twins (Node4Q), nullable
  const uint32_t* tris;
  const xform* poses;            // Tbm per particle
  void* attrs;                   // rmclhip_particle_attributes[n]
  uint32_t n_particles;
  const float* beams;            // rmclhip_range_measurement[n_beams] (16 floats each), device
  uint32_t n_beams;
  xform Tsb;
  float dist_sigma, rhsm, rmsh, rmsm, range_min, range_max;
  uint32_t max_n_meas;
  float* errors;                 // nullable [n_particles*n_beams]
  uint32_t particles_per_block;
  uint32_t raw_ng;               // correspondence_type 2: error against Embree's un-normalised Ng
  uint32_t sim_min_range;        // sim hit requires t > sensor_range.min (Embree updater :47; the OptiX program does not)
  float ray_tfar;                // inf (Embree updater :27) or 1e4 (optixTrace tmax, BeamEvaluateProgram.cu:48)
  uint32_t beams_at_origin;      // every beam starts at (0,0,0) of the sensor frame (what PCDSensorUpdaterEmbree::update builds,
                                 // :314-327): Tsm * orig is Tsm.t, the rotation of a zero vector is skipped
  uint32_t refill_thr, tail_lanes;  // schedule of the persistent-lane kernel: refill when >= refill_thr lanes of a wave are idle;
                                 // leave the node phase when <= tail_lanes lanes still descend and a lane holds a leaf
  uint32_t nb_magic;             // floor(2^32 / n_beams) + 1: ray index / n_beams = umulhi(ray, nb_magic) for ray * n_beams < 2^32
  // Round 4 -- particle-coherent mapping (rmclhip_pf_set_mapping): the block's rays are dealt out PARTICLE-minor (ray j = beam j / np
  // of particle j % np), so the 64 lanes of a wave hold the SAME beam of 64 particles -- nearly the same ray when the cloud has
  // converged and neighbouring slots hold neighbouring particles (order: slot -> particle index, sorted by a Morton key of x, y, yaw;
  // null = identity).  Same rays, same per-particle merge order: results do not depend on the mapping.
  uint32_t particle_minor;
  const uint32_t* order;         // nullable [n_particles]
  // correspondence_type 1 (closest point): the map's near grid (kernels.h NearGrid; null: unseeded queries) and the record count
  const uint32_t* near_grid;
  uint32_t gn[3];
  float gorg[3], ginv[3];
  uint32_t n_tris;
  // k_pf_update_v3: the beam errors of a workgroup wait for the dense pass in GLOBAL scratch ([n_particles * n_beams] floats, written
  // and read back by the same workgroup: L2) instead of LDS -- 16 KB less LDS per workgroup, 7 instead of 4 workgroups per CU
  // (profiles/r04_pf_occupancy.txt).  null: the LDS form of rounds 3.
  float* evals;
  // k_pf_update_v3<..., kAccum> (round 5): g^i for i = 0 .. n_beams, g = max_n_meas / (max_n_meas + 1), and 1 / (max_n_meas + 1) -- the
  // closed-form merge weights of a particle's beams (kernels.hip "order-independent likelihood accumulation"); null: the forms of rounds 3 / 4
  const double* gpow;
  double inv_max1;
};

// per-call inputs of the device-resident MICP loop: written by ONE H2D copy so that the whole loop can be a
// static hipGraph (kernel arguments never change between replays)
struct MicpCall {
  xform Tsm, Tms;   // find pose: Tom * Tbo * Tsb and its inverse
  xform Tsb, Tbo;
  float max_dist;
  float rho_cap, tau_cap;   // moment form of the loop (launch_micp_fast): bounds on the pre-transforms it may meet
  uint32_t seq;             // sequence number of this call: echoed in the completion tag the chain's last kernel publishes
  float gate_lo, gate_hi;   // moment form: the classification holds for every max_dist in [gate_lo, gate_hi] (devmath.h micp_gate_class)
  uint32_t pad[2];
};

// the part of MicpCall the moment-form kernels read, passed BY VALUE when the chain is launched directly (no hipGraph, no H2D
// copy node: kernel arguments are fresh on every launch anyway)
struct MicpCallLite {
  xform Tsb, Tbo;
  float max_dist, rho_cap, tau_cap;
  uint32_t seq;
  float gate_lo, gate_hi;   // see MicpCall
};

// MICP-L inner-loop state kept on the device between launches (correct_once)
struct MicpState {
  xform T_onew_oold;
  xform T_snew_sold;             // pre-transform for the next reduction
  cstats stats_o;                // last merged statistics, odom frame
};

// moment form of the schedule-(R) loop (kernels.hip: "gate-stable moment form"): one streaming pass + one single-workgroup
// launch for all iterations.  status->code: 0 = done (state_out valid), 1 = a pre-transform left (rho_cap, tau_cap) at
// iteration `iter`, 2 = more than 4096 uncertain correspondences; for 1 and 2 the caller runs the per-iteration form.
struct MicpFastStatus { uint32_t code, iter, n_uncertain; float max_rho, max_tau; uint32_t pad[3]; };
constexpr uint32_t kMicpFastMoments = 96;
inline uint32_t micp_fast_blocks(uint32_t n) {
  const uint32_t b = (n + 255u) / 256u;   // one correspondence per thread for small scans, up to 128 rows
  return b < 1u ? 1u : (b > 128u ? 128u : b);
}
// partials: micp_fast_blocks(n) * 96 doubles; unc_mask: ceil(n / 64) words
hipError_t launch_micp_fast(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                            const float* model_normals, const uint8_t* model_mask, uint32_t n, const MicpCall* call,
                            double* partials, unsigned long long* unc_mask, uint32_t n_iter, MicpState* state_out,
                            MicpFastStatus* status, unsigned long long* done, hipStream_t s,
                            const MicpCallLite* call_by_value = nullptr);   // non-null: `call` is ignored

// the loop launch alone, after launch_find_moments: `nblocks` partial rows, mask words in the find's tile order (decoded with
// W / tiles_x / tile_w_log2 of that find)
hipError_t launch_micp_fast_loop_tiled(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                                       const float* model_normals, const uint8_t* model_mask, uint32_t n, uint32_t nblocks,
                                       const double* partials, const unsigned long long* unc_mask, uint32_t W, uint32_t tiles_x,
                                       uint32_t tile_w_log2, uint32_t words_per_block, uint32_t n_iter, MicpState* state_out,
                                       MicpFastStatus* status, unsigned long long* done, hipStream_t s, const MicpCallLite& call_by_value,
                                       double* fold_rows, uint32_t* fold_flags);   // [kMicpFoldBlocks][96] doubles / flags, zeroed once; null: one workgroup
constexpr uint32_t kMicpFoldBlocks = 8;
// Round 4 -- the iterations on the host (micp_host.h): fold the rows like the loop launch does, then hand {82 moments, undecided
// count, the undecided correspondences' D | I | N} to the host behind one completion tag {seq, xor of the written words}.
// _tiled: after launch_find_moments (mask words in the find's tile order); _moments_publish: the separate pass + the publish.
hipError_t launch_micp_publish_tiled(const float* dataset_points, const float* model_points, const float* model_normals, uint32_t n,
                                     uint32_t nblocks, const double* partials, const unsigned long long* unc_mask, uint32_t W,
                                     uint32_t tiles_x, uint32_t tile_w_log2, uint32_t words_per_block, MicpHostBlock* host_block,
                                     unsigned long long* done, uint32_t seq, double* fold_rows, uint32_t* fold_flags, hipStream_t s);
hipError_t launch_micp_moments_publish(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                                       const float* model_normals, const uint8_t* model_mask, uint32_t n, double* partials,
                                       unsigned long long* unc_mask, const MicpCallLite& cv, MicpHostBlock* host_block,
                                       unsigned long long* done, hipStream_t s);

// N-sensor MICP loop on the device (micp_localization.cpp:900-964): per-call frames + per-sensor partials, one step launch per
// iteration merges every sensor's statistics (weighted and unweighted), solves once and hands every sensor its next
// pre-transform
constexpr uint32_t kMaxMicpSensors = 8;
struct MicpMultiCall {
  xform Tsb[kMaxMicpSensors], Tbo[kMaxMicpSensors];
  double weight[kMaxMicpSensors];              // merge_weight_multiplier
  const double* partials[kMaxMicpSensors];     // [nblocks[s]][16]
  uint32_t nblocks[kMaxMicpSensors];
  uint32_t n_sensors, seq;                     // seq: echoed in the completion tag (see MicpCall)
};
struct MicpMultiState {
  xform T_onew_oold;
  cstats merged_o, merged_weighted_o;
  xform T_snew_sold[kMaxMicpSensors];
};
// moment form of the N-sensor loop: launch_micp_moments once per sensor (its MicpCall carries max_dist and the caps), then ONE
// single-workgroup launch for all iterations of all sensors; status->code as MicpFastStatus (0 done, 1 a pre-transform of
// sensor `sensor` left its caps, 2 more than 4096 undecided correspondences in total)
struct MicpMultiFastStatus {
  uint32_t code, iter, n_uncertain, sensor;
  float max_rho[kMaxMicpSensors], max_tau[kMaxMicpSensors];
};
struct MicpMultiFastParams {
  const float* dataset_points[kMaxMicpSensors];
  const float* model_points[kMaxMicpSensors];
  const float* model_normals[kMaxMicpSensors];
  const double* partials[kMaxMicpSensors];               // [nblocks][96] of launch_micp_moments
  const unsigned long long* unc_mask[kMaxMicpSensors];
  uint32_t n[kMaxMicpSensors], nblocks[kMaxMicpSensors];
  // per-call data BY VALUE (round 3: no H2D copy nodes in the chain, and nothing the one solving lane has to fetch from global
  // memory inside the iterations): frames, merge weights, gates and caps of the sensors
  xform Tsb[kMaxMicpSensors], Tbo[kMaxMicpSensors];
  double weight[kMaxMicpSensors];
  float max_dist[kMaxMicpSensors], rho_cap[kMaxMicpSensors], tau_cap[kMaxMicpSensors];
  uint32_t n_sensors, seq;
  uint32_t n_iter;
  MicpMultiState* state_out;                              // may be host-mapped
  MicpMultiFastStatus* status;                            // may be host-mapped
  unsigned long long* done;                               // host-mapped completion tag (see kernels.hip publish_tag)
  // sensors whose find + moment pass ran on ANOTHER stream (bit s of join_mask): the loop waits for join_flags[s] == seq before it
  // touches sensor s's rows -- an in-kernel wait of ~1 us where a cross-stream event takes ~10 us to reach the waiting queue
  const uint32_t* join_flags;
  uint32_t join_mask;
};
// one lane stores `seq` to *flag with release semantics at device scope: enqueued behind a sensor's moment pass on that sensor's stream
hipError_t launch_signal_flag(uint32_t* flag, uint32_t seq, hipStream_t s);
// {seq, 0} to a pinned completion tag, behind whatever the stream holds
hipError_t launch_host_tag(unsigned long long* done, uint32_t seq, hipStream_t s);
hipError_t launch_micp_moments(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                               const float* model_normals, const uint8_t* model_mask, uint32_t n, const MicpCall* call,
                               double* partials, unsigned long long* unc_mask, hipStream_t s, const MicpCallLite* call_by_value = nullptr);
hipError_t launch_micp_multi_fast_loop(const MicpMultiFastParams& p, hipStream_t s);
hipError_t launch_micp_multi_init(const MicpMultiCall* call, MicpMultiState* state, hipStream_t s);
hipError_t launch_micp_multi_step(const MicpMultiCall* call, MicpMultiState* state, hipStream_t s);

// pose batches in world order (kernels.hip: keys + counting sort over 4096 cells of the map's bounding box)
uint32_t batch_order_scratch_dwords(uint32_t n);
hipError_t launch_batch_tile_order(const FindParams& p, ModelKind kind, uint32_t group, f3 bb_min, f3 bb_max, uint32_t* scratch, uint32_t* order,
                                   hipStream_t s);
hipError_t launch_find(const FindParams& p, ModelKind kind, int variant, hipStream_t s);
// kinds 23 / 2 with the MICP moment epilogue: grid of find_moments_blocks(p, kind) workgroups, one partial row per workgroup, one mask
// word per wave (23) or per workgroup (2: a tile is a workgroup) (p.mom_* set by the caller); followed by launch_micp_fast_loop_tiled
uint32_t find_moments_blocks(const FindParams& p, int variant);
hipError_t launch_find_moments(const FindParams& p, ModelKind kind, int variant, hipStream_t s);   // variant 23 (a mask word per wave) or 2 (per workgroup)
// the plane table of the frontier start (kinds 23 / 24): tiles_x * tiles_y * 16 floats for p's model and tiling
hipError_t launch_tile_planes(const FindParams& p, ModelKind kind, float* planes, hipStream_t s);
// diagnostics (tools/probe_find.py): per-wave step timeline of one spherical scan; probe_log: tiles x 512 dwords
hipError_t launch_find_probe(const FindParams& p, int mode, uint32_t* probe_log, hipStream_t s);
// the map's near grid (traverse.hip.h CpcParams): `cells` = one record index per cell, x fastest; org / inv = the box's corner and
// cells per metre per axis
struct NearGrid {
  const uint32_t* cells;
  uint32_t n[3];
  float org[3], inv[3];
};
// grid (nullable): seeds points without a tracking seed; cells (nullable): BUILD a grid -- the query points are the centres of cells
// [0, n) of `cells` (whose records are not read), rec_out receives their records
hipError_t launch_cpc_find(const uint32_t* nodes, const uint32_t* tris, const float* dataset_points, uint32_t n,
                           float max_dist, xform Tsm, xform Tms, uint8_t* hits, float* dists, float* points,
                           float* normals, uint32_t* face_ids, bool quad, hipStream_t s, const uint32_t* seed_rec = nullptr,
                           uint32_t* rec_out = nullptr, uint32_t n_tris = 0, float bound_d2 = 3.0e38f, const NearGrid* grid = nullptr,
                           const NearGrid* cells = nullptr, float skip_d2 = 3.0e38f);
// residual resampling (ResidualResamplerCPU.cpp:55-203) in three steps; stats: 32 bytes on the device {double sum, double max,
// u64 n * E[copies per draw], u64 draws used}
hipError_t launch_residual_prepare(const void* attrs, uint32_t n, uint32_t n_new, double* psum, float* pmax, void* stats, hipStream_t s);
hipError_t launch_residual_draws(const void* attrs, uint32_t n, uint32_t n_new, const void* stats, uint32_t n_draws, uint64_t seed,
                                 uint32_t step, uint32_t* draw_idx, uint32_t* draw_cnt, unsigned long long* incl,
                                 unsigned long long* block_tot, hipStream_t s);
hipError_t launch_residual_fill(const xform* poses, const void* attrs, const uint32_t* draw_idx, const unsigned long long* incl,
                                uint32_t n_draws, xform* poses_new, void* attrs_new, uint32_t n_new, uint32_t first, uint32_t count,
                                const float* cfg8, void* stats, uint64_t seed, uint32_t step, hipStream_t s);
hipError_t launch_gladiator_resample(const xform* poses, const void* attrs, uint32_t n, xform* poses_new, void* attrs_new,
                                     uint32_t first, uint32_t count, const float* cfg8, uint32_t trans_dist_metric,
                                     uint64_t seed, uint32_t step, hipStream_t s);
// psum / pmax: scratch of 256 entries each; out2: {sum, max} (device)
hipError_t launch_likelihood_stats(const void* attrs, uint32_t n, double* psum, float* pmax, float* out2, hipStream_t s);
// the same {sum, max} of a DENSE weight vector (the gathered likelihood.mean of a sharded cloud): bit-identical to the call above on
// attributes that hold the same n values
hipError_t launch_likelihood_stats_dense(const float* weights, uint32_t n, double* psum, float* pmax, float* out2, hipStream_t s);
hipError_t launch_pointcloud2_unpack(const uint8_t* data, uint32_t point_step, uint32_t row_step, uint32_t off_x,
                                     uint32_t off_y, uint32_t off_z, bool is_f64, uint32_t h_skip, uint32_t h_inc,
                                     uint32_t w_skip, uint32_t w_inc, uint32_t out_w, uint32_t out_h, float range_min,
                                     float range_max, float* dirs, float* points, uint8_t* mask, uint32_t* n_valid,
                                     hipStream_t s);
// the 16-wide twins of the child-major nodes (layout.h Node16C), built on the device from them: one thread per (node, entry)
hipError_t launch_build_cnodes16(const uint32_t* cnodes, uint32_t n_nodes, uint32_t* cnodes16, hipStream_t s);
hipError_t launch_compose_poses(const xform* Tbm_dev, xform Tsb, xform* Tsm_out, xform* Tms_out, uint32_t n,
                                hipStream_t s);
uint32_t reduce_num_blocks(uint32_t n, uint32_t nposes);
hipError_t launch_reduce_partials(const ReduceParams& p, hipStream_t s);
// finalize one pose's partials into CrossStatistics (writes to out, which may be host-mapped memory)
// done (nullable, host-mapped, single pose only): completion tag {seq, xor of the 16 result words}, one 8-byte store after
// `out` (kernels.hip publish_tag): the host polls the tag and VERIFIES the sum -- a flag alone is not enough, see capi_rcc.cpp wait_done
hipError_t launch_reduce_finalize(const double* partials, uint32_t nblocks, uint32_t nposes, cstats* out,
                                  unsigned long long* done, uint32_t seq, hipStream_t s);
// finalize + (Tsb*, Tbo*) + umeyama + compose; advances MicpState on the device
hipError_t launch_micp_step(const double* partials, uint32_t nblocks, xform Tsb, xform Tbo, const MicpCall* call,
                            const MicpState* state, MicpState* state_out, hipStream_t s);
// closing launch of the launch_micp_iter chain: solve of the last iteration + T_onew_oold / stats_o (the chain itself works in
// the sensor frame, kernels.hip micp_advance_sensor)
// done (nullable, host-mapped): completion tag {call->seq, xor of the state's words}, stored after the results
hipError_t launch_micp_close(const double* partials, uint32_t nblocks, const MicpCall* call, const MicpState* state,
                             MicpState* state_out, unsigned long long* done, hipStream_t s);
// state: TWO MicpState slots (ping-pong of k_micp_iter); both initialised
hipError_t launch_micp_init(MicpState* state, uint32_t* barrier, hipStream_t s);
// one launch per MICP iteration: finishes the previous iteration (finalize + solve, redundantly in every block)
// and streams the next reduction; a final launch_micp_step closes the last iteration
hipError_t launch_micp_iter(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                            const float* model_normals, const uint8_t* model_mask, uint32_t n, uint32_t nblocks,
                            const MicpCall* call, const double* partials_prev, double* partials_out,
                            const MicpState* state_in, MicpState* state_out, bool first, hipStream_t s);
// persistent loop: n_iter x (reduce, barrier, finalize + solve) in one launch; nblocks <= number of CUs (the grid
// must be co-resident); partials: 2 * nblocks * 16 doubles; barrier: one uint32, zeroed by launch_micp_init
hipError_t launch_micp_loop(const float* dataset_points, const uint8_t* dataset_mask, const float* model_points,
                            const float* model_normals, const uint8_t* model_mask, uint32_t n, uint32_t n_iter,
                            const MicpCall* call, double* partials, uint32_t* barrier, MicpState* state,
                            uint32_t nblocks, bool one_xcd, hipStream_t s);
// batch: per pose finalize + umeyama -> Tdelta (sensor->base conjugated), stats
hipError_t launch_batch_solve(const double* partials, uint32_t nblocks, uint32_t nposes, xform Tsb,
                              xform* Tdelta_out, cstats* stats_out, hipStream_t s);
hipError_t launch_dataset_from_ranges(const float* ranges, const float* model_tab, uint32_t kind, uint32_t W,
                                      uint32_t H, f3 orig, const float* pin_fc, float rmin, float rmax, float* points,
                                      uint8_t* mask, uint32_t* n_valid, hipStream_t s);
hipError_t launch_pf_update(const PfParams& p, int variant, hipStream_t s);
hipError_t launch_pf_extract_weights(const void* attrs, uint32_t n, float* weights, hipStream_t s);
hipError_t launch_pf_motion(const uint32_t* nodes, const uint32_t* tris, xform* poses, void* attrs, uint32_t n,
                            xform T_bnew_bold, double forget_rate, uint32_t max_n_meas, bool collision, hipStream_t s);

// pose-estimate moments (RmclNode::estimateStats): partials = 256 * 32 doubles of scratch, out32 = 24 sums + 8 maxima (device)
hipError_t launch_pose_moments(const xform* poses, const void* attrs, uint32_t n, int pass, double L_sum, xform Tbm,
                               double* partials, double* out32, hipStream_t s);
hipError_t launch_loopback_allreduce(const double* const* send, uint32_t world, double* recv, uint32_t count, bool is_max, hipStream_t s);
hipError_t launch_compact_shards(const float* padded, float* dense, uint32_t n_total, uint32_t world, uint32_t cap, hipStream_t s);
// the same for records of record_bytes (a multiple of 4): the padded all-gather layout of a ragged partition -> the dense cloud
hipError_t launch_compact_records(const void* padded, void* dense, uint32_t n_total, uint32_t world, uint32_t cap, uint32_t record_bytes,
                                  hipStream_t s);

}  // namespace rmclhip
