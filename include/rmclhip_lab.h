/* rmclhip_lab.h -- EXPERIMENTS of librmclhip (librmclhip_lab.so): measured-and-rejected kernel variants and in-kernel
 * instrumentation, kept buildable for A/B runs (tools/) and for the `lab` group of the GPU tests.  NOT part of the drop-in
 * boundary: nothing here replaces a reference interface, and a product build never needs this header or that library.
 *
 * Loading librmclhip_lab.so (dlopen / ctypes.CDLL, after librmclhip.so) registers its launchers with the product; from then on
 *   rmclhip_rcc_set_variant accepts the traversal kinds
 *       1  one lane per ray, branch-free step                 4  one lane per ray on the 64-B quantised nodes
 *       5  while-while step, tails of every wave finished by quads
 *       6 / 7   kind 5 + the top 85 / 341 nodes resident in LDS (north_star "LDS-staged node tiles"; flat loads)
 *       8       kind 5 + one-round-trip leaves                 9 / 10  kind 8 + the LDS top
 *       11 the round-1 branchy step                            12 branch-free step + one-round-trip leaves
 *       13 / 14 wave-uniform nodes through the scalar cache ("wavefront ballot"; 14: + one-round-trip leaves)
 *       16 / 17 branch-free step + quad-finished tails (17: + one-round-trip leaves)      20 kind 16 + the leaf trigger
 *       19 / 21 / 22  round 2's automatic choices (17 / 5 / 4 with the leaf trigger), superseded by 23 / 24 = 19 / 22 + frontier start
 *       25 kind 2 (four lanes per ray) + frontier start        26 kind 23 on the 64-B quantised nodes
 *       27 kind 23 + prefetch of the hit record's normal / face id during the traversal
 *       28 kind 23 with a software-pipelined node step (next node requested after three of the five ordering steps, top of the stack in a register)
 *       29 / 30 kind 23 with four / three of the five ordering steps of a node's children    (round 3: 25 ... 30 measured, not adopted)
 *     (spherical model only; results are bit-identical to the product's kinds, tests/test_gpu_lab.py), and
 *   rmclhip_pf_set_variant accepts the round kernels (bits 4..6 = 0) and the round-2 persistent kernel (bits 7 / 8).
 * Measurements of every kind: profiles/r02_find_variants_ab.txt, profiles/r03_find_variants_ab.txt, DESIGN.md 4. */
#ifndef RMCLHIP_LAB_H
#define RMCLHIP_LAB_H

#include "rmclhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* exported by librmclhip_lab.so */
const char* rmclhip_lab_version(void);

/* The two entry points below are exported by librmclhip.so (they need the handle's internals) but work only while
 * librmclhip_lab.so is loaded -- the instrumented kernels live there; RMCLHIP_ERR_UNSUPPORTED otherwise. */

/* tools/wave_timeline.py: one spherical find() of the current variant whose waves record their entry / exit shader clock:
 * out = n_waves x 8 dwords {s_memtime entry, exit (stores completed), s_memrealtime entry (100 MHz), tile | xcc << 24,
 * s_memtime before the traversal, after it, stores issued, 0} (all zero = wave had no tile) */
rmclhip_status rmclhip_debug_wave_clock(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est, uint32_t* out, size_t cap_dwords,
                                        uint32_t* n_waves_out);
/* the moments the LAST moment-form attempt of rmclhip_rcc_correct_once worked from (rmclhip.h: rmclhip_rcc_set_micp_fast): the 96 sums
 * of its partial rows (82 used; layout in kernels.hip) and the number of correspondences it left undecided.  Works without the
 * experiments library. */
rmclhip_status rmclhip_debug_micp_moments(rmclhip_rcc* rcc, double* totals96, uint32_t* n_rows_out, uint64_t* n_uncertain_out);
/* tools/probe_find.py: one spherical find() through an instrumented copy of the one-lane-per-ray traversal that stamps
 * s_memtime around every node / leaf step of every wave.  mode: bit 0 = one-round-trip leaves, bit 1 = LDS-resident top of the
 * tree.  log_out: n_tiles x 512 dwords (host). */
rmclhip_status rmclhip_debug_probe_find(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est, int mode, uint32_t* log_out,
                                        size_t log_cap_dwords, uint32_t* n_tiles_out);

/* A/B knobs of the cooperative descent of find kinds 32 / 31 (traverse.hip.h frontier_descent_start): the wave stops descending when a
 * level would leave more than final_cap entries (<= 64, default 64) or after (max_levels & 255) levels (default 24; 0 = kind 23's
 * frontier start with the descent's bookkeeping); bits 8..15 of max_levels, when not 0: kind 32's bound on the final leaves one ray may
 * enter before its wave starts at the root instead (default 24); bits 29..30, when not 0: 1 + the tile mapping of single scans (0 as
 * dealt, 1 an eighth of the image per XCD, 2 a CU's two workgroups from the image's halves; find_kernel.hip.h) instead of the tuned /
 * default one; bit 31 set: descend on the four-wide nodes even where the map carries the 16-wide twins (A/B).  Results do not depend on any of them.  Exported by librmclhip.so. */
rmclhip_status rmclhip_rcc_set_descent(rmclhip_rcc* rcc, uint32_t final_cap, uint32_t max_levels);

/* A/B knob (round 6, VERDICT r5 #7b): pose batches in WORLD ORDER (the default; on = 0 restores the pose-major launch).  Every batch launch
 * (find_batch, correct_batch, simulate with several poses, the sharded forms) of this operator makes one
 * key per workgroup of the launch (four neighbouring tiles of one pose; Morton code of the point where their central ray leaves the map's
 * bounding box), sorts the
 * tiles by key (counting sort over 4096 cells) and walks them in that order, `on` workgroups of consecutive tiles per XCD turn (1 = the
 * default, 64), so that what runs on an XCD at a time -- one L2 -- sees one region of the map instead of one pose's whole scan.
 * Results do not depend on it.  Exported by librmclhip.so. */
rmclhip_status rmclhip_rcc_set_batch_order(rmclhip_rcc* rcc, int on);

/* TEST knob of the loopback communicator (rmclhip_comm_create_loopback): its all-reduce adds the ranks' contributions starting at
 * `first_rank` instead of rank 0 -- the freedom a real collective library has.  Whatever must not depend on the library's order of
 * summation is tested under several rotations (tests/test_gpu_distributed.py).  Exported by librmclhip.so. */
rmclhip_status rmclhip_comm_loopback_set_reduce_rotation(rmclhip_comm* comm, uint32_t first_rank);

/* debug trace of the sharded entry points (rmclhip_pf_update_sharded, _allgather_weights, _sharded_resample*): on = 1 starts (and
 * clears) the recording, on = 0 stops it; buf (nullable) receives what was recorded before this call.  Tokens: "E<r>" rank r's part of
 * a phase was enqueued, "W<r>" the host waited for rank r, "<phase>:" labels.  A phase that lets the devices run concurrently reads
 * "E0 E1 ... W0 W1 ..."; tests/test_gpu_distributed.py asserts that.  Works without the experiments library. */
rmclhip_status rmclhip_debug_trace(int on, char* buf, size_t cap);
/* how often, in this process, a host poller found ITS sequence number in a completion tag while the result words it read did not yet add
 * up to the tag's checksum -- the event the {sequence, xor} tag guards against (tools/tag_retries.py, profiles/r05_tag_handoff.txt) */
rmclhip_status rmclhip_debug_tag_retries(unsigned long long* retries_out);

#ifdef __cplusplus
}
#endif
#endif
