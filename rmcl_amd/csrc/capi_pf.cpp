// capi_pf.cpp -- see capi_internal.h
#include "capi_internal.h"

// ---- particle filter -------------------------------------------------------------------------------
rmclhip_status rmclhip_pf_create(rmclhip_ctx* ctx, rmclhip_map* map, rmclhip_pf** out) {
  ApiGuard guard_("rmclhip_pf_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "pf_create: out is null");
  *out = nullptr;
  if (!ctx || !map) return fail(RMCLHIP_ERR_INVALID, "pf_create: NO MAP");
  HIPCHK(hipSetDevice(ctx->device));
  rmclhip_pf* f = new rmclhip_pf();
  f->ctx = ctx;
  ctx_retain(ctx);
  f->map = map;
  rmclhip_map_retain(map);
  hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&f->ev0);
  if (e == hipSuccess) e = hipEventCreate(&f->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_beams, hipEventDisableTiming);
  if (e == hipSuccess) e = f->tag.create();
  if (e != hipSuccess) {
    rmclhip_pf_destroy(f);
    return fail(RMCLHIP_ERR_HIP, std::string("pf_create: ") + hipGetErrorString(e));
  }
  *out = f;
  return RMCLHIP_OK;
}

void rmclhip_pf_destroy(rmclhip_pf* f) {
  ApiGuard guard_("rmclhip_pf_destroy");
  if (!f) return;
  (void)hipSetDevice(f->ctx->device);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  f->d_beams.release(); f->d_evals.release(); f->d_gpow.release(); f->d_order.release(); f->tag.destroy();
  if (f->h_beams) (void)hipHostFree(f->h_beams);
  if (f->ev0) (void)hipEventDestroy(f->ev0);
  if (f->ev1) (void)hipEventDestroy(f->ev1);
  if (f->ev_beams) (void)hipEventDestroy(f->ev_beams);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  rmclhip_map_release(f->map);
  ctx_release(f->ctx);
  delete f;
}

rmclhip_status rmclhip_pf_set_params(rmclhip_pf* f, const rmclhip_pf_params* p) {
  ApiGuard guard_("rmclhip_pf_set_params");
  if (!f || !p) return fail(RMCLHIP_ERR_INVALID, "pf_set_params: null");
  if (!(p->dist_sigma > 0.f)) return fail(RMCLHIP_ERR_INVALID, "pf_set_params: dist_sigma must be > 0");
  // a NaN penalty would become a NaN beam error; the default (accumulating) update evaluates exp through fmaxf, which swallows NaN
  // into a zero likelihood where the reference's exp(NaN) poisons the particle: refuse the parameter instead
  if (p->real_hit_sim_miss_error != p->real_hit_sim_miss_error || p->real_miss_sim_hit_error != p->real_miss_sim_hit_error ||
      p->real_miss_sim_miss_error != p->real_miss_sim_miss_error || p->dist_sigma != p->dist_sigma)
    return fail(RMCLHIP_ERR_INVALID, "pf_set_params: NaN parameter");
  if (p->correspondence_type > 3u)
    return fail(RMCLHIP_ERR_INVALID, "pf_set_params: correspondence_type must be 0 (RCC), 1 (CPC), 2 (RCC, Embree rules) or 3 (RCC, OptiX rules)");
  f->params = *p;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_set_error_output(rmclhip_pf* f, float* errors_dev) {
  ApiGuard guard_("rmclhip_pf_set_error_output");
  if (!f) return fail(RMCLHIP_ERR_INVALID, "pf_set_error_output: null");
  f->errors_dev = errors_dev;
  return RMCLHIP_OK;
}

static rmclhip_status pf_upload_beams(rmclhip_pf* f, const rmclhip_range_measurement* beams, uint32_t n_beams) {
  const size_t nf = static_cast<size_t>(n_beams) * 16;
  HIPCHK(f->d_beams.reserve(nf));
  // the staging buffer may still be the source of the PREVIOUS call's copy: wait for that copy alone (its event), not for whatever else
  // the stream holds -- a motion update enqueued just before this call keeps running (rmclhip_pf_sharded_step)
  if (f->beams_copy_pending) { HIPCHK(hipEventSynchronize(f->ev_beams)); f->beams_copy_pending = false; }
  if (f->h_beams_cap < nf) {
    if (f->h_beams) (void)hipHostFree(f->h_beams);
    f->h_beams = nullptr;
    f->h_beams_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&f->h_beams), nf * sizeof(float), hipHostMallocDefault));
    f->h_beams_cap = nf;
  }
  std::memcpy(f->h_beams, beams, nf * sizeof(float));
  f->beams_at_origin = true;
  for (uint32_t b = 0; b < n_beams && f->beams_at_origin; ++b)
    f->beams_at_origin = beams[b].orig.x == 0.0f && beams[b].orig.y == 0.0f && beams[b].orig.z == 0.0f;
  HIPCHK(hipMemcpyAsync(f->d_beams.p, f->h_beams, nf * sizeof(float), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipEventRecord(f->ev_beams, f->stream));
  f->beams_copy_pending = true;
  return RMCLHIP_OK;
}

static rmclhip_status pf_enqueue(rmclhip_pf* f, const rmclhip_transform* poses, rmclhip_particle_attributes* attrs,
                                 uint32_t n, uint32_t n_beams, const rmclhip_transform* Tsb) {
  PfParams p;
  std::memset(&p, 0, sizeof(p));
  p.nodes = f->map->d_nodes;
  p.qnodes = f->pf_tree ? f->map->d_qnodes_pf : f->map->d_qnodes;
  p.tris = f->map->d_tris;
  p.poses = reinterpret_cast<const xform*>(poses);
  p.attrs = attrs;
  p.n_particles = n;
  p.beams = f->d_beams.p;
  p.n_beams = n_beams;
  p.Tsb = to_x(Tsb);
  p.dist_sigma = f->params.dist_sigma;
  p.rhsm = f->params.real_hit_sim_miss_error;
  p.rmsh = f->params.real_miss_sim_hit_error;
  p.rmsm = f->params.real_miss_sim_miss_error;
  p.range_min = f->params.sensor_range.min;
  p.range_max = f->params.sensor_range.max;
  p.max_n_meas = f->params.max_n_meas;
  p.errors = f->errors_dev;
  p.raw_ng = (f->params.correspondence_type == 2u) ? 1u : 0u;
  p.sim_min_range = (f->params.correspondence_type == 3u) ? 0u : 1u;
  p.ray_tfar = (f->params.correspondence_type == 3u) ? 1.0e4f : std::numeric_limits<float>::infinity();
  // particles per workgroup: ~2048 rays per block (measured 4-6 % faster than 4096: shorter tail per block, more
  // blocks to balance), at most 64 particles, evals must fit 32 KB of LDS
  uint32_t pb = (f->big_blocks ? 4096u : 2048u) / n_beams;
  if (pb < 1u) pb = 1u;
  if (pb > 64u) pb = 64u;
  // the accumulation form keeps 384 B of accumulators per particle of the workgroup in LDS: at most 16 particles in the default
  // (beam-minor) dealing; the particle-minor mapping below chooses its own count (32 or the caller's, <= 64: 24 KB of accumulators --
  // results do not depend on it, the accumulators are order-independent) (few beams per particle
  // would otherwise put 64 of them, 23 KB, into a workgroup: 100 000 x 16 beams 0.189 -> 0.157 ms, x 32 0.338 -> 0.273, x 64 0.522 -> 0.485;
  // the stored form 0.169 / 0.295 / 0.491 -- tools/pf_shapes_ab.py, profiles/r05_pf_forms_ab.txt)
  if (f->accum && f->params.correspondence_type != 1u && pb > 16u) pb = 16u;
  // small clouds: fewer particles per workgroup until the launch has ~4 workgroups per CU (1000 particles x 256 beams in 125 workgroups
  // left half of the chip idle), but never less than one ray per lane
  // (round 4: 1000 x 256: 0.104 -> 0.046 ms, 10 000 x 64: 0.127 -> 0.076 ms; clouds of >= 10 000 x 256 are unchanged)
  while (pb > 1u && n / pb < 1024u && static_cast<uint64_t>(pb >> 1) * n_beams >= 256u) pb >>= 1;
  p.particle_minor = 0u;
  p.order = nullptr;
  p.near_grid = nullptr;
  p.n_tris = f->map->info.n_records;   // record indices (a face that spatial splits reference k times has k records)
  for (int k = 0; k < 3; ++k) { p.gn[k] = 1u; p.gorg[k] = 0.f; p.ginv[k] = 1.f; }
  if (f->params.correspondence_type == 1u && f->cpc_grid) {
    // closest-point errors: every query starts from the near grid's record of its cell (the FULL grid: beam end points are anywhere)
    const NearGrid* grid = nullptr;
    if (rmclhip_status gst = ensure_near_grid(f->map, f->stream, true, &grid)) return gst;
    if (grid) {
      p.near_grid = grid->cells;
      for (int k = 0; k < 3; ++k) { p.gn[k] = grid->n[k]; p.gorg[k] = grid->org[k]; p.ginv[k] = grid->inv[k]; }
    }
  }
  if (f->mapping == 1) {
    // particle-minor dealing: a wave's lanes hold the same beam of `pb` consecutive slots; errors of pb x n_beams beams stay in LDS
    p.particle_minor = 1u;
    pb = f->map_ppb ? f->map_ppb : 32u;
    while (pb > 1u && static_cast<size_t>(pb) * n_beams * 4u > 96u * 1024u) pb >>= 1;
    if (f->order && f->order_n == n) p.order = f->order;
  }
  if (static_cast<size_t>(pb) * n_beams > (p.particle_minor ? 24576u : 8192u)) return fail(RMCLHIP_ERR_UNSUPPORTED, "pf_update: more than 8192 beams");
  p.particles_per_block = pb;
  p.evals = nullptr;
  p.gpow = nullptr;
  p.inv_max1 = 0.0;
  const bool accum = f->accum && f->params.correspondence_type != 1u;
  if (accum) {
    if (f->gpow_beams != n_beams || f->gpow_max != f->params.max_n_meas || f->d_gpow.p == nullptr) {
      std::vector<double> g(static_cast<size_t>(n_beams) + 1u);
      const double base = static_cast<double>(f->params.max_n_meas) / (static_cast<double>(f->params.max_n_meas) + 1.0);
      for (uint32_t i = 0; i <= n_beams; ++i) g[i] = std::pow(base, static_cast<double>(i));
      HIPCHK(hipStreamSynchronize(f->stream));   // an update in flight may still read the old table
      HIPCHK(f->d_gpow.reserve(g.size()));
      HIPCHK(upload_on(f->stream, f->d_gpow.p, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
      f->gpow_beams = n_beams; f->gpow_max = f->params.max_n_meas;
    }
    p.gpow = f->d_gpow.p;
    p.inv_max1 = 1.0 / (static_cast<double>(f->params.max_n_meas) + 1.0);
  }
  if (!accum && f->evals_global && f->params.correspondence_type != 1u) {
    // (the blocks of the last, partial workgroup included: slots are addressed from the workgroup's first particle)
    const size_t slots = (static_cast<size_t>(n) + pb - 1u) / pb * pb;
    const hipError_t re = f->d_evals.reserve(slots * n_beams);
    if (re == hipSuccess) p.evals = f->d_evals.p;
    else if (re == hipErrorOutOfMemory) (void)hipGetLastError();   // no room for the scratch: the LDS form of rounds 3, same results
    else HIPCHK(re);
  }
  p.beams_at_origin = f->beams_at_origin ? 1u : 0u;
  static const uint32_t kRefillAt[5] = {48u, 8u, 16u, 32u, 48u};
  p.refill_thr = f->refill_thr ? f->refill_thr : kRefillAt[f->refill];
  p.tail_lanes = f->tail_lanes;
  // pb * n_beams <= 8192, n_beams <= 8192: exact.  n_beams == 1 has no 32-bit magic (2^32 + 1): the kernel takes pi = ray there
  p.nb_magic = (n_beams == 1u) ? 0u : static_cast<uint32_t>((1ull << 32) / n_beams) + 1u;
  const int variant = (f->variant & 3) | ((std::max(f->map->info.stack_need, f->map->info.stack_need_pf) > 32) ? 4 : 0) | (f->params.correspondence_type == 1u ? 8 : 0) |
                      (f->refill << 4) | (f->full_nodes ? 128 : 0) | (f->legacy ? 256 : 0) | (f->pf_tree ? 0 : 1024) | (f->slot_order ? 2048 : 0) | (accum ? 4096 : 0);
  HIPCHK(launch_pf_update(p, variant, f->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_update_async(rmclhip_pf* f, const rmclhip_transform* poses, rmclhip_particle_attributes* attrs,
                                       uint32_t n, const rmclhip_range_measurement* beams, uint32_t n_beams,
                                       const rmclhip_transform* Tsb) {
  ApiGuard guard_("rmclhip_pf_update_async");
  if (!f || !Tsb) return fail(RMCLHIP_ERR_INVALID, "pf_update: null");
  if (n == 0 || n_beams == 0) return RMCLHIP_OK;
  if (!poses || !attrs || !beams) return fail(RMCLHIP_ERR_INVALID, "pf_update: null buffers");
  HIPCHK(hipSetDevice(f->ctx->device));
  if (rmclhip_status st = pf_upload_beams(f, beams, n_beams)) return st;
  return pf_enqueue(f, poses, attrs, n, n_beams, Tsb);
}

rmclhip_status rmclhip_pf_update(rmclhip_pf* f, const rmclhip_transform* poses, rmclhip_particle_attributes* attrs,
                                 uint32_t n, const rmclhip_range_measurement* beams, uint32_t n_beams,
                                 const rmclhip_transform* Tsb) {
  ApiGuard guard_("rmclhip_pf_update");
  if (rmclhip_status st = rmclhip_pf_update_async(f, poses, attrs, n, beams, n_beams, Tsb)) return st;
  if (n == 0 || n_beams == 0) return RMCLHIP_OK;
  HIPCHK(f->tag.wait_chain_end(f->ctx, f->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_sync(rmclhip_pf* f) {
  ApiGuard guard_("rmclhip_pf_sync");
  if (!f) return fail(RMCLHIP_ERR_INVALID, "pf_sync: null");
  HIPCHK(hipSetDevice(f->ctx->device));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_motion_update(rmclhip_pf* f, rmclhip_transform* poses_dev, rmclhip_particle_attributes* attrs_dev,
                                        uint32_t n, const rmclhip_transform* T_bnew_bold, double forget_rate,
                                        int check_collision) {
  ApiGuard guard_("rmclhip_pf_motion_update");
  if (!f || !T_bnew_bold) return fail(RMCLHIP_ERR_INVALID, "pf_motion_update: null");
  if (n == 0) return RMCLHIP_OK;
  if (!poses_dev || !attrs_dev) return fail(RMCLHIP_ERR_INVALID, "pf_motion_update: null buffers");
  HIPCHK(hipSetDevice(f->ctx->device));
  HIPCHK(launch_pf_motion(f->map->d_qnodes, f->map->d_tris, reinterpret_cast<xform*>(poses_dev), attrs_dev, n,
                          to_x(T_bnew_bold), forget_rate, f->params.max_n_meas, check_collision != 0, f->stream));
  HIPCHK(f->tag.wait_chain_end(f->ctx, f->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_extract_weights(rmclhip_pf* f, const rmclhip_particle_attributes* attrs, uint32_t n,
                                          float* weights_dev) {
  ApiGuard guard_("rmclhip_pf_extract_weights");
  if (!f || (!attrs && n) || (!weights_dev && n)) return fail(RMCLHIP_ERR_INVALID, "pf_extract_weights: null");
  HIPCHK(hipSetDevice(f->ctx->device));
  HIPCHK(launch_pf_extract_weights(attrs, n, weights_dev, f->stream));
  HIPCHK(f->tag.wait_chain_end(f->ctx, f->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_time_update(rmclhip_pf* f, const rmclhip_transform* poses, rmclhip_particle_attributes* attrs,
                                      uint32_t n, const rmclhip_range_measurement* beams, uint32_t n_beams,
                                      const rmclhip_transform* Tsb, uint32_t iters, float* ms) {
  ApiGuard guard_("rmclhip_pf_time_update");
  if (!f || !ms || iters == 0 || !Tsb || !poses || !attrs || !beams || n == 0 || n_beams == 0)
    return fail(RMCLHIP_ERR_INVALID, "pf_time_update: bad arguments");
  HIPCHK(hipSetDevice(f->ctx->device));
  if (rmclhip_status st = pf_upload_beams(f, beams, n_beams)) return st;
  if (rmclhip_status st = pf_enqueue(f, poses, attrs, n, n_beams, Tsb)) return st;
  HIPCHK(hipStreamSynchronize(f->stream));
  HIPCHK(hipEventRecord(f->ev0, f->stream));
  for (uint32_t i = 0; i < iters; ++i)
    if (rmclhip_status st = pf_enqueue(f, poses, attrs, n, n_beams, Tsb)) return st;
  HIPCHK(hipEventRecord(f->ev1, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  float total = 0.f;
  HIPCHK(hipEventElapsedTime(&total, f->ev0, f->ev1));
  *ms = total / static_cast<float>(iters);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_time_update_unfused(rmclhip_pf* f, const rmclhip_transform* poses, rmclhip_particle_attributes* attrs, uint32_t n,
                                              const rmclhip_range_measurement* beams, uint32_t n_beams, const rmclhip_transform* Tsb,
                                              int sync_each_beam, uint32_t iters, float* ms) {
  ApiGuard guard_("rmclhip_pf_time_update_unfused");
  if (!f || !ms || iters == 0 || !Tsb || !poses || !attrs || !beams || n == 0 || n_beams == 0)
    return fail(RMCLHIP_ERR_INVALID, "pf_time_update_unfused: bad arguments");
  auto sequence = [&]() -> rmclhip_status {
    for (uint32_t b = 0; b < n_beams; ++b) {
      const rmclhip_status st = sync_each_beam ? rmclhip_pf_update(f, poses, attrs, n, beams + b, 1u, Tsb)
                                               : rmclhip_pf_update_async(f, poses, attrs, n, beams + b, 1u, Tsb);
      if (st) return st;
    }
    return sync_each_beam ? RMCLHIP_OK : rmclhip_pf_sync(f);
  };
  if (rmclhip_status st = sequence()) return st;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < iters; ++i)
    if (rmclhip_status st = sequence()) return st;
  *ms = static_cast<float>(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_set_schedule(rmclhip_pf* f, uint32_t refill_idle_lanes, uint32_t tail_lanes) {
  ApiGuard guard_("rmclhip_pf_set_schedule");
  if (!f || refill_idle_lanes > 64u || tail_lanes > 64u) return fail(RMCLHIP_ERR_INVALID, "pf_set_schedule: bad arguments");
  f->refill_thr = refill_idle_lanes;
  f->tail_lanes = tail_lanes;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_set_mapping(rmclhip_pf* f, int mapping, uint32_t particles_per_block, const uint32_t* order_dev, uint32_t n_order) {
  ApiGuard guard_("rmclhip_pf_set_mapping");
  if (!f || mapping < 0 || (mapping & 0xFF) > 1 || (mapping >> 10) != 0 || particles_per_block > 64u) return fail(RMCLHIP_ERR_INVALID, "pf_set_mapping: bad arguments");
  f->cpc_grid = ((mapping >> 8) & 1) == 0;   // bit 8 (A/B): closest-point errors WITHOUT the near-grid seed
  f->evals_global = ((mapping >> 9) & 1) == 0;   // bit 9 (A/B): beam errors in LDS (rounds 3) instead of global scratch
  mapping &= 0xFF;
  f->mapping = mapping;
  f->map_ppb = particles_per_block;
  f->order = order_dev;
  f->order_n = order_dev ? n_order : 0u;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_set_variant(rmclhip_pf* f, int variant) {
  ApiGuard guard_("rmclhip_pf_set_variant");
  if (!f || variant < 0 || (variant & 15) > 2 || ((variant >> 4) & 7) > 4 || (variant >> 13) != 0) return fail(RMCLHIP_ERR_INVALID, "pf_set_variant: bad arguments");
  const int kind = variant & 15, refill = (variant >> 4) & 7;
  const bool full_nodes = ((variant >> 7) & 1) != 0, legacy = ((variant >> 8) & 1) != 0;
  // validate BEFORE the handle is touched: a rejected configuration must not stay behind (every later update would fail)
  if ((refill == 0 || legacy || full_nodes || kind != 0) && lab_hooks() == nullptr)
    return fail(RMCLHIP_ERR_UNSUPPORTED, "pf_set_variant: the round kernels and the round-2 persistent kernel are experiments -- they live in "
                                         "librmclhip_lab.so, which is not loaded");
  f->variant = kind;
  f->refill = refill;
  f->full_nodes = full_nodes;
  f->legacy = legacy;                         // the round-2 kernel
  f->big_blocks = ((variant >> 9) & 1) != 0;  // 4096 instead of 2048 rays per workgroup
  f->pf_tree = ((variant >> 10) & 1) == 0;    // bit 10: traverse the map's tree (leaves <= 4) instead of the filter's own
  f->slot_order = ((variant >> 11) & 1) != 0; // bit 11 (round 5, A/B): children in the ray's slot order instead of sorted by entry distance
  f->accum = ((variant >> 12) & 1) == 0;      // bit 12 (A/B): the stored form of rounds 3 / 4 (errors in scratch, dense pass, in-order chain) instead of the round-5 accumulation
  return RMCLHIP_OK;
}


// PCDSensorUpdaterEmbree::update, beam sampling (PCDSensorUpdaterEmbree.cpp:276-327), on the raw message bytes (host).
rmclhip_status rmclhip_pf_sample_beams_pointcloud2(const uint8_t* data, size_t nbytes, const rmclhip_pointcloud2_layout* L,
                                                   uint32_t samples, uint64_t seed, rmclhip_range_measurement* beams_out,
                                                   uint32_t* n_out) {
  ApiGuard guard_("rmclhip_pf_sample_beams_pointcloud2");
  if (!L || !n_out || (!beams_out && samples)) return fail(RMCLHIP_ERR_INVALID, "pf_sample_beams_pointcloud2: null");
  *n_out = 0;
  if (L->datatype != 7u && L->datatype != 8u)
    return fail(RMCLHIP_ERR_UNSUPPORTED, "pf_sample_beams_pointcloud2: Field X has unknown DataType (FLOAT32 / FLOAT64 only)");
  const uint64_t n_points = static_cast<uint64_t>(L->width) * L->height;
  if (samples == 0) return RMCLHIP_OK;
  if (n_points == 0 || !data) return fail(RMCLHIP_ERR_INVALID, "pf_sample_beams_pointcloud2: empty cloud");
  const uint32_t fsz = (L->datatype == 8u) ? 8u : 4u;
  const uint32_t max_off = std::max(L->offset_x, std::max(L->offset_y, L->offset_z));
  const uint64_t last = static_cast<uint64_t>(L->height - 1u) * L->row_step + static_cast<uint64_t>(L->width - 1u) * L->point_step + max_off + fsz;
  if (last > nbytes) return fail(RMCLHIP_ERR_INVALID, "pf_sample_beams_pointcloud2: cloud data shorter than its layout");
  // the reference draws from a function-static std::mt19937 seeded by std::random_device through a
  // std::uniform_int_distribution (implementation defined); pinned here: mt19937(seed), index = draw % n_points
  std::mt19937 gen(static_cast<uint32_t>(seed));
  auto load = [&](const uint8_t* p) -> float {
    if (fsz == 8u) { double d; std::memcpy(&d, p, 8); return static_cast<float>(d); }
    float f; std::memcpy(&f, p, 4); return f;
  };
  for (uint32_t sidx = 0; sidx < samples; ++sidx) {
    bool valid = false;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int t = 0; t < 100 && !valid; ++t) {
      const uint64_t id = static_cast<uint64_t>(gen()) % n_points;
      const uint8_t* ptr = data + (id / L->width) * L->row_step + (id % L->width) * L->point_step;
      x = load(ptr + L->offset_x); y = load(ptr + L->offset_y); z = load(ptr + L->offset_z);
      valid = (x == x) && (y == y) && (z == z);   // NaN only, like the reference (:303): +-inf passes
    }
    if (!valid) break;   // "Point invalid": the reference returns early (:306-311)
    rmclhip_range_measurement m;
    std::memset(&m, 0, sizeof(m));
    const float norm = std::sqrt((x * x + y * y) + z * z);   // rm::Vector3::l2norm
    m.dir = {x / norm, y / norm, z / norm};                  // rm::Vector3::normalize
    m.range = norm;
    m.cov[0] = m.cov[4] = m.cov[8] = 0.1f;                   // Rrs * (Identity * 0.1) * Rrs^T with Trs = Identity
    beams_out[(*n_out)++] = m;
  }
  return RMCLHIP_OK;
}

// ---- resampling --------------------------------------------------------------------------------
rmclhip_status rmclhip_resampler_create(rmclhip_ctx* ctx, rmclhip_resampler** out) {
  ApiGuard guard_("rmclhip_resampler_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "resampler_create: out is null");
  *out = nullptr;
  if (!ctx) return fail(RMCLHIP_ERR_INVALID, "resampler_create: null context");
  HIPCHK(hipSetDevice(ctx->device));
  rmclhip_resampler* r = new rmclhip_resampler();
  r->ctx = ctx;
  ctx_retain(ctx);
  hipError_t e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = r->d_psum.reserve(256);
  if (e == hipSuccess) e = r->d_pmax.reserve(256);
  if (e == hipSuccess) e = r->d_out.reserve(2);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_out), 2 * sizeof(float), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_res), 8 * sizeof(unsigned long long), hipHostMallocDefault);
  if (e == hipSuccess) e = r->tag.create();
  if (e != hipSuccess) {
    rmclhip_resampler_destroy(r);
    return fail(RMCLHIP_ERR_HIP, std::string("resampler_create: ") + hipGetErrorString(e));
  }
  *out = r;
  return RMCLHIP_OK;
}

void rmclhip_resampler_destroy(rmclhip_resampler* r) {
  ApiGuard guard_("rmclhip_resampler_destroy");
  if (!r) return;
  (void)hipSetDevice(r->ctx->device);
  if (r->stream) (void)hipStreamSynchronize(r->stream);
  r->d_psum.release();
  r->d_pmax.release();
  r->d_out.release();
  r->d_res_stats.release(); r->d_res_incl.release(); r->d_res_btot.release(); r->d_res_idx.release(); r->d_res_cnt.release();
  if (r->h_out) (void)hipHostFree(r->h_out);
  if (r->h_res) (void)hipHostFree(r->h_res);
  r->tag.destroy();
  if (r->stream) (void)hipStreamDestroy(r->stream);
  ctx_release(r->ctx);
  delete r;
}

rmclhip_status rmclhip_resampler_compute_stats(rmclhip_resampler* r, const rmclhip_particle_attributes* attrs_dev,
                                               uint32_t n, rmclhip_likelihood_stats* out) {
  ApiGuard guard_("rmclhip_resampler_compute_stats");
  if (!r || !out || (!attrs_dev && n)) return fail(RMCLHIP_ERR_INVALID, "resampler_compute_stats: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(launch_likelihood_stats(attrs_dev, n, r->d_psum.p, r->d_pmax.p, r->d_out.p, r->stream));
  HIPCHK(hipMemcpyAsync(r->h_out, r->d_out.p, 2 * sizeof(float), hipMemcpyDeviceToHost, r->stream));
  HIPCHK(hipStreamSynchronize(r->stream));
  out->sum = r->h_out[0];
  out->max = r->h_out[1];
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_resampler_compute_stats_weights(rmclhip_resampler* r, const float* weights_dev, uint32_t n, rmclhip_likelihood_stats* out) {
  ApiGuard guard_("rmclhip_resampler_compute_stats_weights");
  if (!r || !out || (!weights_dev && n)) return fail(RMCLHIP_ERR_INVALID, "resampler_compute_stats_weights: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(launch_likelihood_stats_dense(weights_dev, n, r->d_psum.p, r->d_pmax.p, r->d_out.p, r->stream));
  HIPCHK(hipMemcpyAsync(r->h_out, r->d_out.p, 2 * sizeof(float), hipMemcpyDeviceToHost, r->stream));
  HIPCHK(hipStreamSynchronize(r->stream));
  out->sum = r->h_out[0];
  out->max = r->h_out[1];
  return RMCLHIP_OK;
}

// The resamplers as ENQUEUE + WAIT (round 4): the sharded entry points enqueue every device's part before they wait for any
// (pf_sharded_resample_impl); the public single-device calls are enqueue + one wait.  `st`: the stream the work goes to (the
// resampler's own, or the communicator's stream of that device behind the all-gather of the cloud).
RMCL_INTERNAL rmclhip_status gladiator_enqueue(rmclhip_resampler* r, const rmclhip_transform* poses_dev, const rmclhip_particle_attributes* attrs_dev,
                                        uint32_t n_particles, rmclhip_transform* poses_new_dev, rmclhip_particle_attributes* attrs_new_dev,
                                        uint32_t first, uint32_t count, const rmclhip_gladiator_config* cfg, uint64_t seed, uint32_t step,
                                        hipStream_t st) {
  if (!r || !cfg) return fail(RMCLHIP_ERR_INVALID, "resampler_gladiator: null");
  if (count == 0) return RMCLHIP_OK;
  if (!poses_dev || !attrs_dev || !poses_new_dev || !attrs_new_dev || n_particles == 0)
    return fail(RMCLHIP_ERR_INVALID, "resampler_gladiator: null particle buffers");
  if (static_cast<uint64_t>(first) + count > n_particles)
    return fail(RMCLHIP_ERR_INVALID, "resampler_gladiator: champion range exceeds the particle count");
  if (cfg->trans_dist_metric > 1u) return fail(RMCLHIP_ERR_INVALID, "resampler_gladiator: trans_dist_metric must be 0 or 1");
  if (poses_new_dev == poses_dev || attrs_new_dev == attrs_dev)
    return fail(RMCLHIP_ERR_INVALID, "resampler_gladiator: the tournament is out of place (double buffers)");
  HIPCHK(hipSetDevice(r->ctx->device));
  const float c8[8] = {cfg->min_noise_tx, cfg->min_noise_ty, cfg->min_noise_tz, cfg->min_noise_roll,
                       cfg->min_noise_pitch, cfg->min_noise_yaw, cfg->likelihood_forget_per_meter,
                       cfg->likelihood_forget_per_radian};
  HIPCHK(launch_gladiator_resample(reinterpret_cast<const xform*>(poses_dev), attrs_dev, n_particles,
                                   reinterpret_cast<xform*>(poses_new_dev), attrs_new_dev, first, count, c8,
                                   cfg->trans_dist_metric, seed, step, st));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_resampler_gladiator(rmclhip_resampler* r, const rmclhip_transform* poses_dev,
                                           const rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                           rmclhip_transform* poses_new_dev, rmclhip_particle_attributes* attrs_new_dev,
                                           uint32_t first, uint32_t count, const rmclhip_gladiator_config* cfg,
                                           uint64_t seed, uint32_t step) {
  ApiGuard guard_("rmclhip_resampler_gladiator");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "resampler_gladiator: null");
  if (rmclhip_status st = gladiator_enqueue(r, poses_dev, attrs_dev, n_particles, poses_new_dev, attrs_new_dev, first, count, cfg, seed, step,
                                            r->stream))
    return st;
  if (count == 0) return RMCLHIP_OK;
  HIPCHK(r->tag.wait_chain_end(r->ctx, r->stream));
  return RMCLHIP_OK;
}

RMCL_INTERNAL rmclhip_status residual_check(ResidualJob& j) {
  j.active = false;
  if (!j.r || !j.cfg) return fail(RMCLHIP_ERR_INVALID, "resampler_residual: null");
  if (j.n_new == 0 || j.count == 0) return RMCLHIP_OK;
  if (!j.poses || !j.attrs || !j.poses_new || !j.attrs_new || j.n_particles == 0)
    return fail(RMCLHIP_ERR_INVALID, "resampler_residual: null particle buffers");
  if (static_cast<uint64_t>(j.first) + j.count > j.n_new) return fail(RMCLHIP_ERR_INVALID, "resampler_residual: slot range exceeds the new cloud");
  if (j.poses_new == j.poses || j.attrs_new == j.attrs)
    return fail(RMCLHIP_ERR_INVALID, "resampler_residual: out of place (double buffers)");
  j.active = true;
  return RMCLHIP_OK;
}
RMCL_INTERNAL rmclhip_status residual_prepare_enqueue(ResidualJob& j) {
  if (!j.active) return RMCLHIP_OK;
  rmclhip_resampler* r = j.r;
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(r->d_res_stats.reserve(4));
  // ResidualResamplerCPU.cpp:72-85
  HIPCHK(launch_residual_prepare(j.attrs, j.n_particles, j.n_new, r->d_psum.p, r->d_pmax.p, r->d_res_stats.p, j.st));
  HIPCHK(hipMemcpyAsync(r->h_res, r->d_res_stats.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, j.st));
  return RMCLHIP_OK;
}
RMCL_INTERNAL rmclhip_status residual_draws_enqueue(ResidualJob& j, bool first_try) {
  if (!j.active || j.filled) return RMCLHIP_OK;
  rmclhip_resampler* r = j.r;
  HIPCHK(hipSetDevice(r->ctx->device));
  if (first_try) {
    double sum; unsigned long long expect;
    std::memcpy(&sum, &r->h_res[0], sizeof(double));
    expect = r->h_res[2];
    if (!(sum > 0.0)) return fail(RMCLHIP_ERR_INVALID, "resampler_residual: the likelihoods sum to zero (or NaN): nothing to resample from");
    if (expect == 0ull)
      return fail(RMCLHIP_ERR_INVALID, "resampler_residual: every particle's share L / sum * N_new truncates to 0 -- no draw would ever insert "
                                       "a particle (the reference's loop, ResidualResamplerCPU.cpp:104, does not terminate on this input)");
    // a block of draws that fills the cloud with a margin: N_new / E[copies per draw] x 1.25 + 4096; doubled if it falls short
    const double per_draw = static_cast<double>(expect) / static_cast<double>(j.n_particles);
    j.want = static_cast<double>(j.n_new) / per_draw * 1.25 + 4096.0;
  } else {
    j.want *= 2.0;   // the same draws again plus as many more: the stream is a function of the draw index
  }
  const double kMaxDraws = 268435456.0;    // 2^28 draws = 4 GB of scratch: far beyond any sane input
  if (j.want > kMaxDraws) return fail(RMCLHIP_ERR_UNSUPPORTED, "resampler_residual: more than 2^28 draws would be needed to fill the cloud");
  j.n_draws = static_cast<uint32_t>(j.want);
  const uint32_t nb = (j.n_draws + 1023u) / 1024u;
  HIPCHK(r->d_res_idx.reserve(j.n_draws));
  HIPCHK(r->d_res_cnt.reserve(j.n_draws));
  HIPCHK(r->d_res_incl.reserve(j.n_draws));
  HIPCHK(r->d_res_btot.reserve(nb));
  HIPCHK(launch_residual_draws(j.attrs, j.n_particles, j.n_new, r->d_res_stats.p, j.n_draws, j.seed, j.step, r->d_res_idx.p, r->d_res_cnt.p,
                               r->d_res_incl.p, r->d_res_btot.p, j.st));
  HIPCHK(hipMemcpyAsync(&r->h_res[4], r->d_res_incl.p + (j.n_draws - 1u), sizeof(unsigned long long), hipMemcpyDeviceToHost, j.st));
  return RMCLHIP_OK;
}
RMCL_INTERNAL void residual_draws_done(ResidualJob& j) {   // after the wait that follows residual_draws_enqueue
  if (j.active && !j.filled) j.filled = j.r->h_res[4] >= j.n_new;
}
RMCL_INTERNAL rmclhip_status residual_fill_enqueue(ResidualJob& j, bool want_n_draws) {
  if (!j.active) return RMCLHIP_OK;
  rmclhip_resampler* r = j.r;
  HIPCHK(hipSetDevice(r->ctx->device));
  const float c8[8] = {j.cfg->min_noise_tx, j.cfg->min_noise_ty, j.cfg->min_noise_tz, j.cfg->min_noise_roll,
                       j.cfg->min_noise_pitch, j.cfg->min_noise_yaw, j.cfg->likelihood_forget_per_meter,
                       j.cfg->likelihood_forget_per_radian};
  HIPCHK(launch_residual_fill(reinterpret_cast<const xform*>(j.poses), j.attrs, r->d_res_idx.p, r->d_res_incl.p, j.n_draws,
                              reinterpret_cast<xform*>(j.poses_new), j.attrs_new, j.n_new, j.first, j.count, c8, r->d_res_stats.p, j.seed,
                              j.step, j.st));
  if (want_n_draws) HIPCHK(hipMemcpyAsync(r->h_res, r->d_res_stats.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, j.st));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_resampler_residual(rmclhip_resampler* r, const rmclhip_transform* poses_dev,
                                          const rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                          rmclhip_transform* poses_new_dev, rmclhip_particle_attributes* attrs_new_dev,
                                          uint32_t n_new, uint32_t first, uint32_t count, const rmclhip_gladiator_config* cfg,
                                          uint64_t seed, uint32_t step, uint64_t* n_draws_out) {
  ApiGuard guard_("rmclhip_resampler_residual");
  if (n_draws_out) *n_draws_out = 0;
  ResidualJob j;
  j.r = r; j.st = r ? r->stream : nullptr;
  j.poses = poses_dev; j.attrs = attrs_dev; j.poses_new = poses_new_dev; j.attrs_new = attrs_new_dev;
  j.n_particles = n_particles; j.n_new = n_new; j.first = first; j.count = count; j.cfg = cfg; j.seed = seed; j.step = step;
  if (rmclhip_status st = residual_check(j)) return st;
  if (!j.active) return RMCLHIP_OK;
  if (rmclhip_status st = residual_prepare_enqueue(j)) return st;
  HIPCHK(hipStreamSynchronize(j.st));
  for (bool first_try = true; !j.filled; first_try = false) {
    if (rmclhip_status st = residual_draws_enqueue(j, first_try)) return st;
    HIPCHK(hipStreamSynchronize(j.st));
    residual_draws_done(j);
  }
  const bool want = n_draws_out && static_cast<uint64_t>(first) + count == n_new;
  if (rmclhip_status st = residual_fill_enqueue(j, want)) return st;
  HIPCHK(hipStreamSynchronize(j.st));
  if (want) *n_draws_out = r->h_res[3];
  return RMCLHIP_OK;
}

