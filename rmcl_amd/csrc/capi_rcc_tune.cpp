// capi_rcc_tune.cpp -- see capi_internal.h
#include "capi_internal.h"


rmclhip_status rmclhip_rcc_set_kernel_timing(rmclhip_rcc* r, int on) {
  ApiGuard guard_("rmclhip_rcc_set_kernel_timing");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_kernel_timing: null");
  r->kernel_timing = on != 0;
  r->find_timing_pending = r->reduce_timing_pending = false;
  return RMCLHIP_OK;
}

// host-clock time of one synchronous rmclhip_rcc_find as a C caller sees it (mean over `iters` calls after one untimed call)
rmclhip_status rmclhip_rcc_time_find_sync(rmclhip_rcc* r, const rmclhip_transform* Tbm_est, uint32_t iters, float* ms_per_call) {
  if (!r || !Tbm_est || !ms_per_call || iters == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_time_find_sync: bad arguments");
  if (rmclhip_status st = rmclhip_rcc_find(r, Tbm_est)) return st;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < iters; ++i)
    if (rmclhip_status st = rmclhip_rcc_find(r, Tbm_est)) return st;
  const auto t1 = std::chrono::steady_clock::now();
  *ms_per_call = static_cast<float>(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_last_kernel_ms(rmclhip_rcc* r, float* find_ms, float* reduce_ms) {
  ApiGuard guard_("rmclhip_rcc_last_kernel_ms");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_last_kernel_ms: null");
  if (r->reduce_timing_pending) {
    HIPCHK(hipSetDevice(r->ctx->device));
    HIPCHK(hipEventSynchronize(r->ev1));
    HIPCHK(hipEventElapsedTime(&r->last_reduce_ms, r->ev0, r->ev1));
    r->reduce_timing_pending = false;
  }
  if (r->find_timing_pending) {   // (find + moment epilogue + publish)
    HIPCHK(hipSetDevice(r->ctx->device));
    HIPCHK(hipEventSynchronize(r->ev1));
    HIPCHK(hipEventElapsedTime(&r->last_find_ms, r->ev0, r->ev1));
    r->find_timing_pending = false;
  }
  if (find_ms) *find_ms = r->last_find_ms;
  if (reduce_ms) *reduce_ms = r->last_reduce_ms;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_time_find(rmclhip_rcc* r, const rmclhip_transform* Tbm_est, uint32_t iters, float* ms) {
  ApiGuard guard_("rmclhip_rcc_time_find");
  if (!r || !Tbm_est || !ms || iters == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_time_find: bad arguments");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_time_find: no sensor model");
  HIPCHK(hipSetDevice(r->ctx->device));
  const xform T = to_x(Tbm_est);
  if (rmclhip_status st = find_enqueue(r, T)) return st;  // warm-up + allocation
  HIPCHK(hipStreamSynchronize(r->stream));
  r->reduce_timing_pending = false;
  HIPCHK(hipEventRecord(r->ev0, r->stream));
  for (uint32_t i = 0; i < iters; ++i)
    if (rmclhip_status st = find_enqueue(r, T)) return st;
  HIPCHK(hipEventRecord(r->ev1, r->stream));
  HIPCHK(hipStreamSynchronize(r->stream));
  float total = 0.f;
  HIPCHK(hipEventElapsedTime(&total, r->ev0, r->ev1));
  *ms = total / static_cast<float>(iters);
  return RMCLHIP_OK;
}

// candidates of the measured choice: {template kind, frontier start}; reported as kinds 2 / 23 / 24 and -- without the frontier
// start -- as round 2's numbers for the same traversals, 19 / 22
struct TuneCand { int kind; bool frontier; int reported; uint32_t descent_cap; };   // descent_cap: kind 32 only (rmclhip_rcc_set_descent), 0 otherwise
// (kind 32 = the cooperative descent below the frontier, with the wave's final list capped at 12, 32 or 64 entries: open maps want the
// long list, rooms the short one or none -- profiles/r06_descent_maps_ab.txt)
static const TuneCand kTuneSingle[8] = {{2, true, 2, 0}, {23, true, 23, 0}, {23, false, 19, 0}, {24, true, 24, 0}, {24, false, 22, 0},
                                        {32, true, 32, 12}, {32, true, 32, 32}, {32, true, 32, 64}};
static const TuneCand kTuneBatch[4] = {{23, true, 23, 0}, {23, false, 19, 0}, {24, true, 24, 0}, {24, false, 22, 0}};

rmclhip_status rmclhip_rcc_autotune(rmclhip_rcc* r, const rmclhip_transform* Tbm_est, int* chosen_kind, float* kernel_ms) {
  ApiGuard guard_("rmclhip_rcc_autotune");
  if (!r || !Tbm_est) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune: bad arguments");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune: no sensor model");
  if (r->variant != 15) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune: a traversal kind is forced (set_variant); nothing to choose");
  HIPCHK(hipSetDevice(r->ctx->device));
  // each candidate timed on THIS map, model and pose: median of 5 batches of 8 back-to-back launches
  const int saved_kind = r->tuned_kind;
  const bool saved_frontier = r->tuned_frontier;
  const uint32_t saved_cap = r->descent_final_cap;
  const TuneCand* best = nullptr;
  float best_ms = 0.f;
  for (const TuneCand& c : kTuneSingle) {
    if (c.kind == 32 && (r->kind == kModelOnDn || r->map->d_cnodes == nullptr)) continue;   // (no common pyramid / no child-major nodes: kind 23)
    r->tuned_kind = c.kind; r->tuned_frontier = c.frontier;
    if (c.descent_cap != 0u) r->descent_final_cap = c.descent_cap;
    float t[5];
    for (float& x : t) {
      if (rmclhip_status st = rmclhip_rcc_time_find(r, Tbm_est, 8, &x)) { r->tuned_kind = saved_kind; r->tuned_frontier = saved_frontier; r->descent_final_cap = saved_cap; return st; }
    }
    std::sort(t, t + 5);
    if (!best || t[2] < best_ms) { best = &c; best_ms = t[2]; }
  }
  r->tuned_kind = best->kind; r->tuned_frontier = best->frontier;
  r->descent_final_cap = best->descent_cap != 0u ? best->descent_cap : saved_cap;
  // ... then the tile shape of the winner (the rule: 16 wide x 4 tall; profiles/r03_find_tile_shapes.txt shows maps that prefer
  // 4 x 16 or 32 x 2): widths 4, 8, 32 where the image is tall enough, the plane table rebuilt for each
  if (r->tile_override == 0) {
    int best_tile = 0;
    for (int tile : {3, 4, 6}) {
      const uint32_t th = 64u >> (tile - 1);
      if (th > r->H && th > 1u) continue;            // taller than the image: lanes without rays
      r->tuned_tile = tile;
      if (rmclhip_status st = rebuild_tile_planes(r, true)) { r->tuned_tile = 0; (void)rebuild_tile_planes(r, true); return st; }
      float t[5];
      for (float& x : t) {
        if (rmclhip_status st = rmclhip_rcc_time_find(r, Tbm_est, 8, &x)) { r->tuned_tile = 0; (void)rebuild_tile_planes(r, true); return st; }
      }
      std::sort(t, t + 5);
      if (t[2] < 0.98f * best_ms) { best_tile = tile; best_ms = t[2]; }   // a 2 % margin: do not chase noise
    }
    r->tuned_tile = best_tile;
    if (rmclhip_status st = rebuild_tile_planes(r, true)) return st;
  }
  // ... and which workgroup computes which tile (find_kernel.hip.h: 0 = as dealt, the default; 1 = an eighth of the image per XCD;
  // 2 = a CU's two workgroups from the two halves of the image)
  if (r->xcd_mapping_override < 0) {
    uint32_t best_map = r->tuned_xcd_mapping;
    for (uint32_t m = 0; m < 3u; ++m) {
      if (m == best_map) continue;
      const uint32_t keep = r->tuned_xcd_mapping;
      r->tuned_xcd_mapping = m;
      float t[5];
      for (float& x : t) {
        if (rmclhip_status st = rmclhip_rcc_time_find(r, Tbm_est, 8, &x)) { r->tuned_xcd_mapping = keep; return st; }
      }
      std::sort(t, t + 5);
      if (t[2] < 0.98f * best_ms) { best_map = m; best_ms = t[2]; }
      r->tuned_xcd_mapping = keep;
    }
    r->tuned_xcd_mapping = best_map;
  }
  r->graph_dirty = true; r->fast_graph_dirty = true;
  if (chosen_kind) *chosen_kind = best->reported;
  if (kernel_ms) *kernel_ms = best_ms;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_autotune_batch(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes, int* chosen_kind, float* kernel_ms) {
  ApiGuard guard_("rmclhip_rcc_autotune_batch");
  if (!r || !Tbm || nposes < 2u) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune_batch: needs at least two poses");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune_batch: no sensor model");
  if (r->variant != 15) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune_batch: a traversal kind is forced (set_variant); nothing to choose");
  HIPCHK(hipSetDevice(r->ctx->device));
  const int saved_kind = r->tuned_batch_kind;
  const bool saved_frontier = r->tuned_batch_frontier;
  const TuneCand* best = nullptr;
  float best_ms = 0.f;
  for (const TuneCand& c : kTuneBatch) {
    r->tuned_batch_kind = c.kind; r->tuned_batch_frontier = c.frontier;
    float t[3] = {0.f, 0.f, 0.f};
    for (float& x : t) {
      if (rmclhip_status st = rmclhip_rcc_time_find_batch(r, Tbm, nposes, 3, &x)) { r->tuned_batch_kind = saved_kind; r->tuned_batch_frontier = saved_frontier; return st; }
    }
    std::sort(t, t + 3);
    if (!best || t[1] < best_ms) { best = &c; best_ms = t[1]; }
  }
  r->tuned_batch_kind = best->kind; r->tuned_batch_frontier = best->frontier;
  r->graph_dirty = true; r->fast_graph_dirty = true;
  if (chosen_kind) *chosen_kind = best->reported;
  if (kernel_ms) *kernel_ms = best_ms;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_time_reduce(rmclhip_rcc* r, const rmclhip_transform* Tpre, uint32_t iters, float* ms) {
  ApiGuard guard_("rmclhip_rcc_time_reduce");
  if (!r || !Tpre || !ms || iters == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_time_reduce: bad arguments");
  HIPCHK(hipSetDevice(r->ctx->device));
  const xform T = to_x(Tpre);
  ReduceTail tail;
  tail.mode = kTailStats;
  tail.stats_out = r->h_stats_dev + 1;
  if (rmclhip_status st = reduce_enqueue(r, T, nullptr, r->max_dist, 1, tail)) return st;
  HIPCHK(hipStreamSynchronize(r->stream));
  r->reduce_timing_pending = false;
  HIPCHK(hipEventRecord(r->ev0, r->stream));
  for (uint32_t i = 0; i < iters; ++i)
    if (rmclhip_status st = reduce_enqueue(r, T, nullptr, r->max_dist, 1, tail)) return st;
  HIPCHK(hipEventRecord(r->ev1, r->stream));
  HIPCHK(hipStreamSynchronize(r->stream));
  float total = 0.f;
  HIPCHK(hipEventElapsedTime(&total, r->ev0, r->ev1));
  *ms = total / static_cast<float>(iters);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_time_correct_once(rmclhip_rcc* r, const rmclhip_transform* Tom, const rmclhip_transform* Tbo,
                                             uint32_t n_iter, double convergence_progress, int refind_each_iteration,
                                             uint32_t iters, float* ms_per_call) {
  if (!r || !Tom || !Tbo || !ms_per_call || iters == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_time_correct_once: bad arguments");
  rmclhip_transform T;
  rmclhip_cross_statistics S;
  // one untimed call (graph capture / allocation), then `iters` complete synchronous corrections on the host clock:
  // what a C or C++ caller of this ABI sees per rmclhip_rcc_correct_once
  if (rmclhip_status st = rmclhip_rcc_correct_once(r, Tom, Tbo, n_iter, convergence_progress, refind_each_iteration, &T, &S)) return st;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < iters; ++i)
    if (rmclhip_status st = rmclhip_rcc_correct_once(r, Tom, Tbo, n_iter, convergence_progress, refind_each_iteration, &T, &S)) return st;
  const auto t1 = std::chrono::steady_clock::now();
  *ms_per_call = static_cast<float>(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  return RMCLHIP_OK;
}

// The reference's caller loop for one sensor, literally (micp_localization.cpp:900-964 with MICPSensor.hpp:146-184), through the
// PUBLIC entry points a C caller has -- find once, then per iteration computeCrossStatistics + the CrossStatistics / Transform
// algebra + umeyama_transform on the host -- on the host clock.  This is the flow an integrator gets WITHOUT replacing the node's
// loop by rmclhip_rcc_correct_once.
rmclhip_status rmclhip_rcc_time_caller_loop(rmclhip_rcc* r, const rmclhip_transform* Tom, const rmclhip_transform* Tbo, uint32_t n_iter,
                                            double convergence_progress, uint32_t iters, rmclhip_transform* T_onew_oold_out,
                                            rmclhip_cross_statistics* merged_out, float* ms_per_call) {
  if (!r || !Tom || !Tbo || !ms_per_call || iters == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_time_caller_loop: bad arguments");
  rmclhip_transform Tsb, Tsb_inv, Tbo_inv, Tbm, T_onew_oold, T_bnew_bold, T_snew_sold, T_inner, tmp;
  rmclhip_cross_statistics Cs_s, Cs_b, Cs_o, ident, merged;
  from_x(r->Tsb, &Tsb);
  from_cs(cs_identity(), &ident);
  merged = ident;
  auto once = [&]() -> rmclhip_status {
    rmclhip_status st;
    if ((st = rmclhip_transform_inv(&Tsb, &Tsb_inv))) return st;
    if ((st = rmclhip_transform_inv(Tbo, &Tbo_inv))) return st;
    if ((st = rmclhip_transform_mult(Tom, Tbo, &Tbm))) return st;             // MICPSensor.hpp:146-151
    if ((st = rmclhip_rcc_find(r, &Tbm))) return st;
    from_x(xidentity(), &T_onew_oold);
    for (uint32_t i = 0; i < n_iter; ++i) {
      if ((st = rmclhip_transform_mult(&Tbo_inv, &T_onew_oold, &tmp))) return st;      // :926
      if ((st = rmclhip_transform_mult(&tmp, Tbo, &T_bnew_bold))) return st;
      if ((st = rmclhip_transform_mult(&Tsb_inv, &T_bnew_bold, &tmp))) return st;      // MICPSensor.hpp:178
      if ((st = rmclhip_transform_mult(&tmp, &Tsb, &T_snew_sold))) return st;
      if ((st = rmclhip_rcc_compute_cross_statistics(r, &T_snew_sold, convergence_progress, &Cs_s))) return st;
      if ((st = rmclhip_cross_statistics_transform(&Tsb, &Cs_s, &Cs_b))) return st;    // MICPSensor.hpp:182
      if ((st = rmclhip_cross_statistics_transform(Tbo, &Cs_b, &Cs_o))) return st;     // :931
      if ((st = rmclhip_cross_statistics_merge(&ident, &Cs_o, &merged))) return st;    // :936
      if ((st = rmclhip_umeyama_transform(&merged, &T_inner))) return st;              // :952
      if ((st = rmclhip_transform_mult(&T_onew_oold, &T_inner, &tmp))) return st;      // :963
      T_onew_oold = tmp;
    }
    return RMCLHIP_OK;
  };
  for (int warm = 0; warm < 2; ++warm)   // (allocation; the second find learns that computeCrossStatistics calls follow a find)
    if (rmclhip_status st = once()) return st;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < iters; ++i)
    if (rmclhip_status st = once()) return st;
  const auto t1 = std::chrono::steady_clock::now();
  *ms_per_call = static_cast<float>(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  if (T_onew_oold_out) *T_onew_oold_out = T_onew_oold;
  if (merged_out) *merged_out = merged;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_variant(rmclhip_rcc* r, int variant) {
  ApiGuard guard_("rmclhip_rcc_set_variant");
  if (!r || variant < 0) return fail(RMCLHIP_ERR_INVALID, "rcc_set_variant: bad arguments");
  // bit 13 adds 16 to the traversal kind (kinds 16..31), bit 14 adds 32
  const int kind = (variant & 0xF) | (((variant >> 13) & 1) << 4) | (((variant >> 14) & 1) << 5), tile = (variant >> 4) & 0xF;
  if (kind == 3 || kind == 18 || kind > 32 || tile > 7 || (variant >> 15) != 0) return fail(RMCLHIP_ERR_INVALID, "rcc_set_variant: unknown variant");
  // (kind 1 also selects the one-lane-per-point form of the closest-point query, which the product owns)
  if (kind != 15 && kind != 1 && !find_kind_in_product(kind) && lab_hooks() == nullptr)
    return fail(RMCLHIP_ERR_UNSUPPORTED, "rcc_set_variant: this traversal kind is an experiment -- it lives in librmclhip_lab.so, "
                                         "which is not loaded (the product builds kinds 0, 2, 23, 24, 32 and the automatic rule 15)");
  r->variant = kind;
  const bool tiling_changed = r->tile_override != tile;
  r->tile_override = tile;
  r->fused_tail = ((variant >> 8) & 1) != 0;
  r->use_graph = ((variant >> 9) & 1) == 0;
  {  // bits 10..12: MICP loop form -- 0 default (one launch per iteration, solve in the prologue), 1 classic
     // (reduce + solve launches), 2..6 persistent loop kernel with 16..256 blocks
    // 7: persistent loop with 32 blocks confined to one XCD (bit 16 of loop_blocks)
    static const int kLoopBlocks[8] = {0, -1, 16, 32, 64, 128, 256, 32 | (1 << 16)};
    r->loop_blocks = kLoopBlocks[(variant >> 10) & 7];
  }
  r->graph_dirty = true; r->fast_graph_dirty = true;
  // the tile shape depends on the kind (packet: 8x8, per-ray: 16x4) and on the override: the plane table follows
  (void)tiling_changed;
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  return rebuild_tile_planes(r);
}

rmclhip_status rmclhip_rcc_set_descent(rmclhip_rcc* r, uint32_t final_cap, uint32_t max_levels) {
  ApiGuard guard_("rmclhip_rcc_set_descent");
  if (!r || final_cap > 64u) return fail(RMCLHIP_ERR_INVALID, "rcc_set_descent: final_cap <= 64");
  r->descent_final_cap = final_cap;
  r->descent_levels = max_levels & 0xFFu;
  if ((max_levels >> 8) & 0xFFu) r->descent_leaf_cap = (max_levels >> 8) & 0xFFu;   // bits 8..15 (A/B): kind 32's leaves-per-ray bound, 0 = keep
  r->xcd_mapping_override = static_cast<int>((max_levels >> 29) & 3u) - 1;   // bits 29..30 (A/B): 1 + FindParams::xcd_mapping, 0 = the tuned / default one
  r->descent_wide = (max_levels >> 31) == 0u;   // bit 31 (A/B): the four-wide nodes even where the map has the 16-wide twins
  r->graph_dirty = true; r->fast_graph_dirty = true;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_micp_fast(rmclhip_rcc* r, int mode) {
  ApiGuard guard_("rmclhip_rcc_set_micp_fast");
  if (!r || mode < 0 || mode > 4)
    return fail(RMCLHIP_ERR_INVALID, "rcc_set_micp_fast: mode must be 0 (off), 1 (automatic, iterations on the host), 2 (device loop replayed from a "
                                     "hipGraph), 3 (device loop, moments in a pass of their own) or 4 (device loop, moments in the find's epilogue)");
  drop_moment_set(r);
  r->fast_mode = mode;
  r->fast_holdoff = 0;
  r->fast_overflows = 0;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_micp_fast_info(const rmclhip_rcc* r, rmclhip_micp_fast_info* out) {
  if (!r || !out) return fail(RMCLHIP_ERR_INVALID, "rcc_micp_fast_info: null");
  *out = r->fast_info;
  out->rho_cap = r->fast_rho_cap;
  out->tau_cap = r->fast_tau_cap;
  return RMCLHIP_OK;
}

// the host side of the moment form alone (micp_host.h), no device involved: classify + accumulate the moments of the given
// correspondences exactly as k_micp_moments / the find's epilogue do, then evaluate statistics_p2l at (Tpre, max_dist) from them
rmclhip_status rmclhip_host_moment_statistics(const float* dataset_points, const float* model_points, const float* model_normals,
                                              const uint8_t* valid, uint32_t n, float gate_lo, float gate_hi, float rho_cap, float tau_cap,
                                              const rmclhip_transform* Tpre_, float max_dist, rmclhip_cross_statistics* out,
                                              uint32_t* n_undecided, int* covered) {
  if ((n != 0 && (!dataset_points || !model_points || !model_normals)) || !Tpre_ || !out)
    return fail(RMCLHIP_ERR_INVALID, "host_moment_statistics: null");
  static thread_local MicpMomentSet ms;
  uint32_t unc = 0;
  const bool fits = micp_set_from_correspondences(dataset_points, model_points, model_normals, valid, n, gate_lo, gate_hi, rho_cap, tau_cap, &ms, &unc);
  if (n_undecided) *n_undecided = unc;
  const xform Tpre = to_x(Tpre_);
  const bool cov = fits && micp_set_covers(ms, Tpre, max_dist);
  if (covered) *covered = cov ? 1 : 0;
  from_cs(cov ? micp_statistics_from_set(ms, Tpre, max_dist) : cs_identity(), out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_ccs_info(const rmclhip_rcc* r, rmclhip_ccs_info* out) {
  if (!r || !out) return fail(RMCLHIP_ERR_INVALID, "rcc_ccs_info: null");
  *out = r->ccs_info;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_find_variant(const rmclhip_rcc* r, uint32_t nposes, int* variant_out) {
  if (!r || !variant_out) return fail(RMCLHIP_ERR_INVALID, "rcc_find_variant: null");
  *variant_out = find_variant(r, nposes ? nposes : 1u);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_debug_probe_find(rmclhip_rcc* r, const rmclhip_transform* Tbm_est, int mode, uint32_t* log_out,
                                        size_t log_cap_dwords, uint32_t* n_tiles_out) {
  ApiGuard guard_("rmclhip_debug_probe_find");
  if (!r || !Tbm_est || !log_out) return fail(RMCLHIP_ERR_INVALID, "debug_probe_find: null");
  if (r->kind != kModelSpherical || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "debug_probe_find: spherical model only");
  if (lab_hooks() == nullptr) return fail(RMCLHIP_ERR_UNSUPPORTED, "debug_probe_find: the probe kernel lives in librmclhip_lab.so, which is not loaded");
  HIPCHK(hipSetDevice(r->ctx->device));
  const size_t n = static_cast<size_t>(r->W) * r->H;
  r->n_model = static_cast<uint32_t>(n);
  r->nposes_last = 1;
  if (rmclhip_status st = ensure_model_buffers(r, n)) return st;
  FindParams p;
  fill_find_params(r, p, 1);
  p.Tsm = xmul(to_x(Tbm_est), r->Tsb);
  p.Tms = xinv(p.Tsm);
  const size_t ntiles = static_cast<size_t>(p.tiles_x) * p.tiles_y, dwords = ntiles * 512u;
  if (n_tiles_out) *n_tiles_out = static_cast<uint32_t>(ntiles);
  if (log_cap_dwords < dwords) return fail(RMCLHIP_ERR_INVALID, "debug_probe_find: log buffer too small");
  uint32_t* d_log = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&d_log), dwords * sizeof(uint32_t)));
  hipError_t e = hipMemsetAsync(d_log, 0, dwords * sizeof(uint32_t), r->stream);
  // a few launches first: the timeline of a warm launch (map in L2) is the one of interest
  for (int i = 0; i < 3 && e == hipSuccess; ++i) e = launch_find_probe(p, mode, d_log, r->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(log_out, d_log, dwords * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
  (void)hipFree(d_log);
  if (e != hipSuccess) return fail(RMCLHIP_ERR_HIP, std::string("debug_probe_find: ") + hipGetErrorString(e));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_debug_micp_moments(rmclhip_rcc* r, double* totals96, uint32_t* n_rows_out, uint64_t* n_uncertain_out) {
  ApiGuard guard_("rmclhip_debug_micp_moments");
  if (!r || !totals96) return fail(RMCLHIP_ERR_INVALID, "debug_micp_moments: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  std::vector<double> rows(static_cast<size_t>(r->last_fast_rows) * kMicpFastMoments);
  std::vector<unsigned long long> words(r->last_fast_words);
  if (!rows.empty()) HIPCHK(hipMemcpy(rows.data(), r->d_fast_partials.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (!words.empty()) HIPCHK(hipMemcpy(words.data(), r->d_fast_mask.p, words.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (uint32_t k = 0; k < kMicpFastMoments; ++k) {
    double acc = 0.0;
    for (uint32_t b = 0; b < r->last_fast_rows; ++b) acc += rows[static_cast<size_t>(b) * kMicpFastMoments + k];
    totals96[k] = acc;
  }
  uint64_t bits = 0;
  for (unsigned long long w : words) bits += static_cast<uint64_t>(__builtin_popcountll(w));
  if (n_rows_out) *n_rows_out = r->last_fast_rows;
  if (n_uncertain_out) *n_uncertain_out = bits;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_debug_wave_clock(rmclhip_rcc* r, const rmclhip_transform* Tbm_est, uint32_t* out, size_t cap_dwords,
                                        uint32_t* n_waves_out) {
  ApiGuard guard_("rmclhip_debug_wave_clock");
  if (!r || !Tbm_est || !out) return fail(RMCLHIP_ERR_INVALID, "debug_wave_clock: null");
  if (r->kind != kModelSpherical || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "debug_wave_clock: spherical model only");
  if (lab_hooks() == nullptr) return fail(RMCLHIP_ERR_UNSUPPORTED, "debug_wave_clock: the clocked kernels live in librmclhip_lab.so, which is not loaded");
  HIPCHK(hipSetDevice(r->ctx->device));
  const size_t n = static_cast<size_t>(r->W) * r->H;
  r->n_model = static_cast<uint32_t>(n);
  r->nposes_last = 1;
  if (rmclhip_status st = ensure_model_buffers(r, n)) return st;
  FindParams p;
  fill_find_params(r, p, 1);
  p.Tsm = xmul(to_x(Tbm_est), r->Tsb);
  p.Tms = xinv(p.Tsm);
  const int variant = find_variant(r, 1);
  const uint32_t ntiles = p.tiles_x * p.tiles_y;
  uint32_t nblocks = (variant == 2) ? ntiles : (ntiles + 3u) / 4u;
  nblocks = (nblocks + 7u) & ~7u;
  // kind 31: 16 more words per wave (the descent's phase stamps) behind the [n_waves][8] table
  const size_t dwords = static_cast<size_t>(nblocks) * 4u * ((variant == 31 || variant == 32) ? 24u : 8u);
  if (n_waves_out) *n_waves_out = nblocks * 4u;
  if (cap_dwords < dwords) return fail(RMCLHIP_ERR_INVALID, "debug_wave_clock: buffer too small");
  uint32_t* d = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&d), dwords * sizeof(uint32_t)));
  hipError_t e = hipSuccess;
  FindParams warm = p;
  for (int i = 0; i < 5 && e == hipSuccess; ++i) e = launch_find(warm, r->kind, variant, r->stream);   // warm, un-instrumented
  if (e == hipSuccess) e = hipMemsetAsync(d, 0, dwords * sizeof(uint32_t), r->stream);
  p.wave_clock = d;
  if (e == hipSuccess) e = launch_find(p, r->kind, variant, r->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, dwords * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RMCLHIP_ERR_HIP, std::string("debug_wave_clock: ") + hipGetErrorString(e));
  return RMCLHIP_OK;
}

// A pose batch in world order (VERDICT r5 #7b): one key per wave-tile of the launch (where its central ray leaves the map's bounding
// box), a counting sort, and k_find walks the sorted list -- consecutive blocks, hence one XCD's L2, then see one region of the map.
// Three small launches per batch (~15 us); the v1 batch of the reference's benchmark gains 1 % (100 k faces) / 8 % (1 M) / 15 % (10 M) / 6 % (room),
// all included (profiles/r06_batch_order_ab.txt).  On by default; rmclhip_rcc_set_batch_order(rcc, 0) restores the pose-major launch (A/B).
RMCL_INTERNAL rmclhip_status batch_order_enqueue(rmclhip_rcc* r, FindParams& p, int variant) {
  const uint32_t ntiles = p.tiles_x * p.tiles_y, group = (variant == 2) ? 1u : 4u, ngroups = (ntiles + group - 1u) / group;
  if (r->batch_order == 0u || p.nposes < 2u || ngroups > 65535u || p.nposes > 32768u || p.Tsm_arr == nullptr) return RMCLHIP_OK;
  const uint32_t n = ngroups * p.nposes;
  HIPCHK(r->d_ord_scratch.reserve(batch_order_scratch_dwords(n))); HIPCHK(r->d_ord_vals.reserve(n));
  const BvhInfo& bi = r->map->info;
  HIPCHK(launch_batch_tile_order(p, r->kind, group, mk3(bi.bbox_min[0], bi.bbox_min[1], bi.bbox_min[2]),
                                 mk3(bi.bbox_max[0], bi.bbox_max[1], bi.bbox_max[2]), r->d_ord_scratch.p, r->d_ord_vals.p, r->stream));
  p.tile_order = r->d_ord_vals.p;
  p.n_tile_order = n;
  p.tile_order_granule = r->batch_order;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_batch_order(rmclhip_rcc* r, int on) {
  ApiGuard guard_("rmclhip_rcc_set_batch_order");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_batch_order: null");
  r->batch_order = (on <= 0) ? 0u : (on == 1 ? 64u : static_cast<uint32_t>(on));   // 1: the default granule
  return RMCLHIP_OK;
}

RMCL_INTERNAL rmclhip_status find_batch_enqueue(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes) {
  if (nposes > 32768) return fail(RMCLHIP_ERR_UNSUPPORTED, "find_batch: at most 32768 poses per call");
  const size_t n = static_cast<size_t>(r->W) * r->H;
  if (rmclhip_status st = ensure_model_buffers(r, n * nposes)) return st;
  HIPCHK(r->d_Tbm.reserve(nposes)); HIPCHK(r->d_Tsm.reserve(nposes)); HIPCHK(r->d_Tms.reserve(nposes));
  HIPCHK(hipMemcpyAsync(r->d_Tbm.p, Tbm, sizeof(xform) * nposes, hipMemcpyHostToDevice, r->stream));
  HIPCHK(launch_compose_poses(r->d_Tbm.p, r->Tsb, r->d_Tsm.p, r->d_Tms.p, nposes, r->stream));
  r->n_model = static_cast<uint32_t>(n);
  r->nposes_last = nposes;
  FindParams p;
  fill_find_params(r, p, nposes);
  p.Tsm_arr = r->d_Tsm.p;
  p.Tms_arr = r->d_Tms.p;
  const int bvariant = find_variant(r, p.nposes);
  if (rmclhip_status st = batch_order_enqueue(r, p, bvariant)) return st;
  HIPCHK(launch_find(p, r->kind, bvariant, r->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_find_batch(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes) {
  ApiGuard guard_("rmclhip_rcc_find_batch");
  if (!r || (!Tbm && nposes)) return fail(RMCLHIP_ERR_INVALID, "find_batch: null");
  if (nposes == 0 || r->kind == kModelNone || r->W == 0 || r->H == 0) return RMCLHIP_OK;
  HIPCHK(hipSetDevice(r->ctx->device));
  if (rmclhip_status st = find_batch_enqueue(r, Tbm, nposes)) return st;
  HIPCHK(hipStreamSynchronize(r->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_time_find_batch(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes,
                                           uint32_t iters, float* ms) {
  ApiGuard guard_("rmclhip_rcc_time_find_batch");
  if (!r || !Tbm || !ms || iters == 0 || nposes == 0) return fail(RMCLHIP_ERR_INVALID, "time_find_batch: bad arguments");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "time_find_batch: no sensor model");
  HIPCHK(hipSetDevice(r->ctx->device));
  if (rmclhip_status st = find_batch_enqueue(r, Tbm, nposes)) return st;
  HIPCHK(hipStreamSynchronize(r->stream));
  FindParams p;
  fill_find_params(r, p, nposes);
  p.Tsm_arr = r->d_Tsm.p;
  p.Tms_arr = r->d_Tms.p;
  r->reduce_timing_pending = false;
  HIPCHK(hipEventRecord(r->ev0, r->stream));
  const int bvariant = (find_variant(r, p.nposes) == 18) ? 17 : find_variant(r, p.nposes);
  for (uint32_t i = 0; i < iters; ++i) {
    if (rmclhip_status st = batch_order_enqueue(r, p, bvariant)) return st;   // (world order on: its keys and sort are part of every batch)
    HIPCHK(launch_find(p, r->kind, bvariant, r->stream));
  }
  HIPCHK(hipEventRecord(r->ev1, r->stream));
  HIPCHK(hipStreamSynchronize(r->stream));
  float total = 0.f;
  HIPCHK(hipEventElapsedTime(&total, r->ev0, r->ev1));
  *ms = total / static_cast<float>(iters);
  return RMCLHIP_OK;
}

