/* rmclhip_bench.h -- MEASUREMENT AIDS of librmclhip.so: timing loops around the public entry points, run inside the library so that
 * bench.py / tools/ time what a C caller sees (no Python between the calls) or what the device does (HIP events on the stream the
 * kernels run on).  NOT part of the drop-in boundary: nothing here replaces a reference interface and no integrator needs this header
 * (round 5: moved out of rmclhip.h).  Every function is exported by librmclhip.so. */
#ifndef RMCLHIP_BENCH_H
#define RMCLHIP_BENCH_H

#include "rmclhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernel timing of the last synchronous find / streaming reduction on the handle's stream (hipEvent, ms).  OPT-IN since round 4
 * (rmclhip_rcc_set_kernel_timing 1): the two hipEventRecord + hipEventElapsedTime per call cost the caller microseconds on calls
 * that take tens; without it the values stay at what the last timed call left (0 initially). */
rmclhip_status rmclhip_rcc_set_kernel_timing(rmclhip_rcc* rcc, int on);
/* host-clock time of one synchronous rmclhip_rcc_find as a C caller sees it (mean over `iters` calls after one untimed call) */
rmclhip_status rmclhip_rcc_time_find_sync(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est, uint32_t iters, float* ms_per_call);
rmclhip_status rmclhip_rcc_last_kernel_ms(rmclhip_rcc* rcc, float* find_ms, float* reduce_ms);
/* benchmarking hook: run `iters` back-to-back find launches on the handle's stream bracketed by
 * hipEvents on THAT stream; returns the mean kernel-to-kernel time per launch in ms */
rmclhip_status rmclhip_rcc_time_find(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est, uint32_t iters,
                                     float* ms_per_launch);
rmclhip_status rmclhip_rcc_time_reduce(rmclhip_rcc* rcc, const rmclhip_transform* T_snew_sold, uint32_t iters,
                                       float* ms_per_launch);
/* host-clock time of one complete synchronous rmclhip_rcc_correct_once as a C caller sees it (mean over `iters` calls
 * after one untimed call); measurement aid like time_find / time_reduce */
rmclhip_status rmclhip_rcc_time_correct_once(rmclhip_rcc* rcc, const rmclhip_transform* Tom, const rmclhip_transform* Tbo,
                                             uint32_t n_iter, double convergence_progress, int refind_each_iteration,
                                             uint32_t iters, float* ms_per_call);
/* host-clock time of the reference's UNCHANGED caller loop for one sensor (micp_localization.cpp:900-964: find once, then n_iter x
 * { computeCrossStatistics, Tsb *, Tbo *, merge, umeyama_transform, compose } on the host) through the public entry points above,
 * mean over `iters` corrections after two untimed ones; also returns the last T_onew_oold / merged statistics (nullable).  What an
 * integrator measures who keeps the node's loop instead of calling rmclhip_rcc_correct_once. */
rmclhip_status rmclhip_rcc_time_caller_loop(rmclhip_rcc* rcc, const rmclhip_transform* Tom, const rmclhip_transform* Tbo,
                                            uint32_t n_iter, double convergence_progress, uint32_t iters,
                                            rmclhip_transform* T_onew_oold_out, rmclhip_cross_statistics* merged_out,
                                            float* ms_per_call);
/* `iters` back-to-back rmclhip_rcc_find_batch launches bracketed by HIP events on the handle's stream (mean ms per launch) */
rmclhip_status rmclhip_rcc_time_find_batch(rmclhip_rcc* rcc, const rmclhip_transform* Tbm, uint32_t nposes,
                                           uint32_t iters, float* ms_per_launch);
/* `iters` back-to-back sensor updates (rmclhip_pf_update's launch) bracketed by HIP events on the updater's stream (mean ms per launch) */
rmclhip_status rmclhip_pf_time_update(rmclhip_pf* pf, const rmclhip_transform* poses_dev,
                                      rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                      const rmclhip_range_measurement* beams, uint32_t n_beams,
                                      const rmclhip_transform* Tsb, uint32_t iters, float* ms_per_launch);

/* The REFERENCE's GPU schedule of the sensor update as a comparator (PCDSensorUpdaterOptix.cpp:319-338: one launch per beam, the
 * particle attributes read and written by every launch, the stream synchronised after every beam, :337): n_beams synchronous
 * rmclhip_pf_update calls of ONE beam each (sync_each_beam != 0), or the same launches enqueued back to back with one wait at the end
 * (sync_each_beam == 0: what the schedule costs the device alone).  Host clock around the whole sequence, mean over `iters` sequences
 * after one untimed one.  Results equal the fused update's (the beams are applied in order); particle traffic is
 * n_beams x n_particles x 104 B instead of n_particles x 104 B (SURVEY 8(d) B_pf unfused). */
rmclhip_status rmclhip_pf_time_update_unfused(rmclhip_pf* pf, const rmclhip_transform* poses_dev,
                                              rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                              const rmclhip_range_measurement* beams, uint32_t n_beams,
                                              const rmclhip_transform* Tsb, int sync_each_beam, uint32_t iters, float* ms_per_sequence);

#ifdef __cplusplus
}
#endif

#endif /* RMCLHIP_BENCH_H */
