// capi_rcc.cpp -- see capi_internal.h
#include "capi_internal.h"

// ---- rcc -------------------------------------------------------------------------------------
rmclhip_status rmclhip_rcc_create(rmclhip_ctx* ctx, rmclhip_map* map, rmclhip_rcc** out) {
  ApiGuard guard_("rmclhip_rcc_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "rcc_create: out is null");
  *out = nullptr;
  if (!ctx || !map) return fail(RMCLHIP_ERR_INVALID, "rcc_create: NO MAP");
  HIPCHK(hipSetDevice(ctx->device));
  rmclhip_rcc* r = new rmclhip_rcc();
  r->ctx = ctx;
  ctx_retain(ctx);
  r->map = map;
  rmclhip_map_retain(map);
  hipError_t e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&r->ev0);
  if (e == hipSuccess) e = hipEventCreate(&r->ev1);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_stats), sizeof(cstats) * 2, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&r->h_stats_dev), r->h_stats, 0);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_state), sizeof(MicpState), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&r->h_state_dev), r->h_state, 0);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->d_state), 2 * sizeof(MicpState));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_fast_status), sizeof(MicpFastStatus), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&r->h_fast_status_dev), r->h_fast_status, 0);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_done), 2 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&r->h_done_dev), r->h_done, 0);
  if (e == hipSuccess) r->h_done[0] = r->h_done[1] = 0ull;
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_mom), sizeof(MicpHostBlock), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&r->h_mom_dev), r->h_mom, 0);
  if (e == hipSuccess) std::memset(r->h_mom, 0, sizeof(MicpHostBlock));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->d_counter), sizeof(uint32_t));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r->h_call), sizeof(MicpCall), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->d_call), sizeof(MicpCall));
  if (e != hipSuccess) {
    rmclhip_rcc_destroy(r);
    return fail(RMCLHIP_ERR_HIP, std::string("rcc_create: ") + hipGetErrorString(e));
  }
  *out = r;
  return RMCLHIP_OK;
}

void rmclhip_rcc_destroy(rmclhip_rcc* r) {
  ApiGuard guard_("rmclhip_rcc_destroy");
  if (!r) return;
  (void)hipSetDevice(r->ctx->device);
#define DBG_STEP(x)                                                                                       \
  do {                                                                                                  \
    const hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess && std::getenv("RMCLHIP_DEBUG")) std::fprintf(stderr, "[rmclhip debug] %s -> %s\n", #x, hipGetErrorString(e_)); \
  } while (0)
  if (r->stream) DBG_STEP(hipStreamSynchronize(r->stream));
  r->d_model_tab.release(); r->d_ds_points.release(); r->d_ds_mask.release();
  r->d_hits.release(); r->d_ranges.release(); r->d_points.release(); r->d_normals.release(); r->d_face_ids.release();
  r->d_partials.release(); r->d_Tbm.release(); r->d_Tsm.release(); r->d_Tms.release(); r->d_Tdelta.release();
  r->d_bstats.release();
  if (r->h_bT) DBG_STEP(hipHostFree(r->h_bT));
  if (r->h_bS) DBG_STEP(hipHostFree(r->h_bS));
  r->d_raw.release();
  DBG_STEP(hipPeekAtLastError());
  if (r->h_stats) DBG_STEP(hipHostFree(r->h_stats));
  if (r->h_state) DBG_STEP(hipHostFree(r->h_state));
  if (r->d_state) DBG_STEP(hipFree(r->d_state));
  if (r->d_loop_barrier) DBG_STEP(hipFree(r->d_loop_barrier));
  if (r->d_counter) DBG_STEP(hipFree(r->d_counter));
  if (r->d_tickets) DBG_STEP(hipFree(r->d_tickets));
  if (r->micp_exec) DBG_STEP(hipGraphExecDestroy(r->micp_exec));
  if (r->micp_graph) DBG_STEP(hipGraphDestroy(r->micp_graph));
  if (r->micp_fast_exec) DBG_STEP(hipGraphExecDestroy(r->micp_fast_exec));
  if (r->micp_fast_graph) DBG_STEP(hipGraphDestroy(r->micp_fast_graph));
  if (r->h_fast_status) DBG_STEP(hipHostFree(r->h_fast_status));
  if (r->h_done) DBG_STEP(hipHostFree(r->h_done));
  if (r->h_mom) DBG_STEP(hipHostFree(r->h_mom));
  r->d_cpc_rec.release();
  r->d_fast_partials.release(); r->d_fast_mask.release(); r->d_tile_planes.release();
  if (r->d_fold_rows) DBG_STEP(hipFree(r->d_fold_rows));
  if (r->ev_join) DBG_STEP(hipEventDestroy(r->ev_join));
  if (r->d_join_flags) DBG_STEP(hipFree(r->d_join_flags));
  r->d_multi_blob.release();
  if (r->h_multi_state) DBG_STEP(hipHostFree(r->h_multi_state));
  if (r->h_multi_status) DBG_STEP(hipHostFree(r->h_multi_status));
  if (r->h_call) DBG_STEP(hipHostFree(r->h_call));
  if (r->d_call) DBG_STEP(hipFree(r->d_call));
  if (r->ev0) DBG_STEP(hipEventDestroy(r->ev0));
  if (r->ev1) DBG_STEP(hipEventDestroy(r->ev1));
  if (r->stream) DBG_STEP(hipStreamDestroy(r->stream));
  rmclhip_map_release(r->map);
  ctx_release(r->ctx);
  delete r;
}

rmclhip_status rmclhip_rcc_set_tsb(rmclhip_rcc* r, const rmclhip_transform* Tsb) {
  ApiGuard guard_("rmclhip_rcc_set_tsb");
  if (!r || !Tsb) return fail(RMCLHIP_ERR_INVALID, "rcc_set_tsb: null");
  r->Tsb = to_x(Tsb);
  return RMCLHIP_OK;
}


rmclhip_status rmclhip_rcc_set_model_spherical(rmclhip_rcc* r, const rmclhip_spherical_model* m) {
  ApiGuard guard_("rmclhip_rcc_set_model_spherical");
  if (!r || !m) return fail(RMCLHIP_ERR_INVALID, "rcc_set_model_spherical: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  const uint32_t H = m->phi.size, W = m->theta.size;
  r->kind = kModelSpherical;
  r->graph_dirty = true; r->fast_graph_dirty = true;
  r->W = W; r->H = H;
  r->range = m->range;
  r->orig = mk3(0.f, 0.f, 0.f);
  r->tile_planes_ok = false;
  r->ang_aspect = (H > 1u && W > 1u && m->theta.inc != 0.0f && std::isfinite(m->phi.inc / m->theta.inc)) ? std::fabs(m->phi.inc / m->theta.inc) : 0.0f;
  if (W == 0 || H == 0) return RMCLHIP_OK;
  // trig tables with the host libm, exactly what rmagine's getDirection evaluates per ray:
  // phi = phi.min + float(vid) * phi.inc, theta likewise
  std::vector<float> tab(2 * static_cast<size_t>(H) + 2 * static_cast<size_t>(W));
  for (uint32_t v = 0; v < H; ++v) {
    const float phi = m->phi.min + static_cast<float>(v) * m->phi.inc;
    tab[v] = cosf(phi);
    tab[H + v] = sinf(phi);
  }
  for (uint32_t h = 0; h < W; ++h) {
    const float th = m->theta.min + static_cast<float>(h) * m->theta.inc;
    tab[2 * H + h] = cosf(th);
    tab[2 * H + W + h] = sinf(th);
  }
  HIPCHK(r->d_model_tab.reserve(tab.size()));
  HIPCHK(upload_on(r->stream, r->d_model_tab.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
  return rebuild_tile_planes(r);
}

rmclhip_status rmclhip_rcc_set_model_o1dn(rmclhip_rcc* r, uint32_t width, uint32_t height, rmclhip_interval range,
                                          rmclhip_vec3 orig, const float* dirs) {
  ApiGuard guard_("rmclhip_rcc_set_model_o1dn");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_model_o1dn: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  r->kind = kModelO1Dn;
  r->ang_aspect = 0.0f;
  r->graph_dirty = true; r->fast_graph_dirty = true;
  r->W = width; r->H = height;
  r->range = range;
  r->orig = mk3(orig.x, orig.y, orig.z);
  const size_t n = static_cast<size_t>(width) * height;
  r->tile_planes_ok = false;
  if (n == 0) return RMCLHIP_OK;
  if (!dirs) return fail(RMCLHIP_ERR_INVALID, "rcc_set_model_o1dn: dirs is null");
  HIPCHK(r->d_model_tab.reserve(3 * n));
  HIPCHK(upload_on(r->stream, r->d_model_tab.p, dirs, 3 * n * sizeof(float), hipMemcpyHostToDevice));
  return rebuild_tile_planes(r);
}

rmclhip_status rmclhip_rcc_set_model_pinhole(rmclhip_rcc* r, uint32_t width, uint32_t height, rmclhip_interval range,
                                             float fx, float fy, float cx, float cy) {
  ApiGuard guard_("rmclhip_rcc_set_model_pinhole");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_model_pinhole: null");
  if (!(fx != 0.f) || !(fy != 0.f)) return fail(RMCLHIP_ERR_INVALID, "rcc_set_model_pinhole: zero focal length");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  r->kind = kModelPinhole;
  r->ang_aspect = 0.0f;
  r->graph_dirty = true; r->fast_graph_dirty = true;
  r->W = width; r->H = height;
  r->range = range;
  r->orig = mk3(0.f, 0.f, 0.f);
  r->pin_fc[0] = fx; r->pin_fc[1] = fy; r->pin_fc[2] = cx; r->pin_fc[3] = cy;
  return rebuild_tile_planes(r);
}

rmclhip_status rmclhip_rcc_set_model_ondn(rmclhip_rcc* r, uint32_t width, uint32_t height, rmclhip_interval range,
                                          const float* origs, const float* dirs) {
  ApiGuard guard_("rmclhip_rcc_set_model_ondn");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_model_ondn: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  r->kind = kModelOnDn;
  r->ang_aspect = 0.0f;
  r->tile_planes_ok = false;
  r->tuned_kind = r->tuned_batch_kind = 0; r->tuned_frontier = r->tuned_batch_frontier = true; r->tuned_tile = 0; r->tuned_xcd_mapping = 0;
  r->graph_dirty = true; r->fast_graph_dirty = true;
  r->W = width; r->H = height;
  r->range = range;
  r->orig = mk3(0.f, 0.f, 0.f);
  const size_t n = static_cast<size_t>(width) * height;
  if (n == 0) return RMCLHIP_OK;
  if (!origs || !dirs) return fail(RMCLHIP_ERR_INVALID, "rcc_set_model_ondn: origs / dirs is null");
  HIPCHK(r->d_model_tab.reserve(6 * n));
  HIPCHK(upload_on(r->stream, r->d_model_tab.p, origs, 3 * n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(upload_on(r->stream, r->d_model_tab.p + 3 * n, dirs, 3 * n * sizeof(float), hipMemcpyHostToDevice));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_params(rmclhip_rcc* r, float max_dist, float adaptive_max_dist_min) {
  ApiGuard guard_("rmclhip_rcc_set_params");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_params: null");
  r->max_dist = max_dist;
  r->adaptive_max_dist_min = adaptive_max_dist_min;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_dataset(rmclhip_rcc* r, const float* pts, const uint8_t* mask, uint32_t n,
                                       int src_is_device) {
  ApiGuard guard_("rmclhip_rcc_set_dataset");
  if (r) drop_moment_set(r);
  if (!r || (!pts && n > 0)) return fail(RMCLHIP_ERR_INVALID, "rcc_set_dataset: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  r->n_dataset = n;
  r->cpc_rec_n = 0;   // a new dataset: the closest-point records of the old one mean nothing
  r->ds_has_mask = (mask != nullptr);
  if (n == 0) return RMCLHIP_OK;
  const hipMemcpyKind kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIPCHK(r->d_ds_points.reserve(3 * static_cast<size_t>(n)));
  HIPCHK(upload_on(r->stream, r->d_ds_points.p, pts, 3 * static_cast<size_t>(n) * sizeof(float), kind));
  if (mask) {
    HIPCHK(r->d_ds_mask.reserve(n));
    HIPCHK(upload_on(r->stream, r->d_ds_mask.p, mask, n, kind));
  }
  r->ds_pts = r->d_ds_points.p;
  r->ds_msk = mask ? r->d_ds_mask.p : nullptr;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_dataset_view(rmclhip_rcc* r, const float* pts_dev, const uint8_t* mask_dev, uint32_t n) {
  ApiGuard guard_("rmclhip_rcc_set_dataset_view");
  if (r) drop_moment_set(r);
  if (!r || (!pts_dev && n > 0)) return fail(RMCLHIP_ERR_INVALID, "rcc_set_dataset_view: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  r->n_dataset = n;
  r->cpc_rec_n = 0;
  r->ds_has_mask = (mask_dev != nullptr);
  r->ds_pts = pts_dev;
  r->ds_msk = mask_dev;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_dataset_from_ranges(rmclhip_rcc* r, const float* ranges, uint32_t n,
                                                   uint32_t* n_valid_out) {
  ApiGuard guard_("rmclhip_rcc_set_dataset_from_ranges");
  if (r) drop_moment_set(r);
  if (!r || !ranges) return fail(RMCLHIP_ERR_INVALID, "rcc_set_dataset_from_ranges: null");
  if (r->kind == kModelNone) return fail(RMCLHIP_ERR_INVALID, "rcc_set_dataset_from_ranges: no sensor model set");
  if (n != r->W * r->H) return fail(RMCLHIP_ERR_INVALID, "rcc_set_dataset_from_ranges: n != model size");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  r->n_dataset = n;
  r->cpc_rec_n = 0;
  r->ds_has_mask = true;
  if (n_valid_out) *n_valid_out = 0;
  if (n == 0) return RMCLHIP_OK;
  HIPCHK(r->d_ds_points.reserve(3 * static_cast<size_t>(n)));
  HIPCHK(r->d_ds_mask.reserve(n));
  // stage the ranges in the (not yet used) ranges model buffer region of a scratch allocation
  DevBuf<float> d_r;
  HIPCHK(d_r.reserve(n));
  hipError_t e = upload_on(r->stream, d_r.p, ranges, n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemsetAsync(r->d_counter, 0, sizeof(uint32_t), r->stream);
  if (e == hipSuccess)
    e = launch_dataset_from_ranges(d_r.p, r->d_model_tab.p, r->kind, r->W, r->H, r->orig, r->pin_fc, r->range.min, r->range.max,
                                   r->d_ds_points.p, r->d_ds_mask.p, r->d_counter, r->stream);
  uint32_t nv = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&nv, r->d_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
  d_r.release();
  r->ds_pts = r->d_ds_points.p;
  r->ds_msk = r->d_ds_mask.p;
  if (e != hipSuccess) return fail(RMCLHIP_ERR_HIP, std::string("dataset_from_ranges: ") + hipGetErrorString(e));
  if (n_valid_out) *n_valid_out = nv;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_input_pointcloud2(rmclhip_rcc* r, const uint8_t* data, size_t nbytes,
                                                 const rmclhip_pointcloud2_layout* L, const rmclhip_filter1d* fh,
                                                 const rmclhip_filter1d* fw, rmclhip_interval range, int src_is_device,
                                                 uint32_t* out_width, uint32_t* out_height, uint32_t* n_valid_out) {
  ApiGuard guard_("rmclhip_rcc_set_input_pointcloud2");
  if (r) drop_moment_set(r);
  if (!r || !L) return fail(RMCLHIP_ERR_INVALID, "rcc_set_input_pointcloud2: null");
  if (L->datatype != 7u && L->datatype != 8u)
    return fail(RMCLHIP_ERR_UNSUPPORTED, "rcc_set_input_pointcloud2: Field X has unknown DataType (FLOAT32 / FLOAT64 only)");
  const rmclhip_filter1d none{0u, 0u, 1u};
  const rmclhip_filter1d h = fh ? *fh : none, w = fw ? *fw : none;
  if (h.increment == 0u || w.increment == 0u || static_cast<uint64_t>(h.skip_begin) + h.skip_end > L->height ||
      static_cast<uint64_t>(w.skip_begin) + w.skip_end > L->width)
    return fail(RMCLHIP_ERR_INVALID, "rcc_set_input_pointcloud2: bad filter options");
  const uint32_t ow = (L->width - w.skip_begin - w.skip_end) / w.increment;
  const uint32_t oh = (L->height - h.skip_begin - h.skip_end) / h.increment;
  const size_t n = static_cast<size_t>(ow) * oh;
  const uint32_t fsz = (L->datatype == 8u) ? 8u : 4u;
  if (n) {
    if (!data) return fail(RMCLHIP_ERR_INVALID, "rcc_set_input_pointcloud2: data is null");
    const uint32_t max_off = std::max(L->offset_x, std::max(L->offset_y, L->offset_z));
    const uint64_t last = static_cast<uint64_t>((oh - 1u) * h.increment + h.skip_begin) * L->row_step +
                          static_cast<uint64_t>((ow - 1u) * w.increment + w.skip_begin) * L->point_step + max_off + fsz;
    if (last > nbytes) return fail(RMCLHIP_ERR_INVALID, "rcc_set_input_pointcloud2: cloud data shorter than its layout");
  }
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  r->kind = kModelO1Dn;
  r->ang_aspect = 0.0f;
  r->graph_dirty = true; r->fast_graph_dirty = true;
  r->W = ow; r->H = oh;
  r->range = range;
  r->orig = mk3(0.f, 0.f, 0.f);
  r->n_dataset = static_cast<uint32_t>(n);
  r->cpc_rec_n = 0;
  r->ds_has_mask = true;
  if (out_width) *out_width = ow;
  if (out_height) *out_height = oh;
  if (n_valid_out) *n_valid_out = 0;
  r->tile_planes_ok = false;
  if (n == 0) return RMCLHIP_OK;
  HIPCHK(r->d_model_tab.reserve(3 * n));
  HIPCHK(r->d_ds_points.reserve(3 * n));
  HIPCHK(r->d_ds_mask.reserve(n));
  r->ds_pts = r->d_ds_points.p;
  r->ds_msk = r->d_ds_mask.p;
  const uint8_t* d_data = data;
  if (!src_is_device) {
    HIPCHK(r->d_raw.reserve(nbytes));
    HIPCHK(hipMemcpyAsync(r->d_raw.p, data, nbytes, hipMemcpyHostToDevice, r->stream));
    d_data = r->d_raw.p;
  }
  HIPCHK(hipMemsetAsync(r->d_counter, 0, sizeof(uint32_t), r->stream));
  HIPCHK(launch_pointcloud2_unpack(d_data, L->point_step, L->row_step, L->offset_x, L->offset_y, L->offset_z, L->datatype == 8u,
                                   h.skip_begin, h.increment, w.skip_begin, w.increment, ow, oh, range.min, range.max,
                                   r->d_model_tab.p, r->d_ds_points.p, r->d_ds_mask.p, r->d_counter, r->stream));
  if (rmclhip_status st = rebuild_tile_planes(r)) return st;   // the directions just written are the O1Dn model
  uint32_t nv = 0;
  HIPCHK(hipMemcpyAsync(&nv, r->d_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
  HIPCHK(hipStreamSynchronize(r->stream));
  if (n_valid_out) *n_valid_out = nv;
  return RMCLHIP_OK;
}

static uint32_t pick_tile_w_log2(uint32_t H, bool packet, float ang_aspect = 0.0f) {
  // Spherical models whose COLUMNS are much sparser than their rows (row spacing / column spacing < 0.5: a 32 x 32 or 16 x 16 model
  // over the full circle) get tall tiles, as square in angle as 64 rays allow: 16 wide x 4 tall tiles of such a model span half the
  // horizon, and the rays of a wave share nothing (round 4, 2000 poses x 32x32: 0.47 -> 0.27 ms; profiles/r04_v1_batch_breakdown.txt).
  if (!packet && ang_aspect > 0.0f && ang_aspect < 0.5f) {
    uint32_t hp = 1;
    while (hp < H && hp < 64u) hp <<= 1;                         // tile height <= the model's (rounded up to a power of two)
    uint32_t min_twl = 0;
    while ((64u >> min_twl) > hp) ++min_twl;
    const float want = 0.5f * std::log2(64.0f * ang_aspect);     // log2 of the width that makes the tile square in angle
    int twl = static_cast<int>(std::lround(want));
    twl = std::max(twl, static_cast<int>(min_twl));
    twl = std::min(std::max(twl, 0), 4);                         // (never wider than the general rule below)
    return static_cast<uint32_t>(twl);
  }
  // The 64 rays of a wave (kind 2: of a block) are a tile of the scan image.  The wave-packet traversal (kind 0), whose rays walk
  // together, keeps round 1's square 8x8 tiles (images at least 8 rows tall; flatter tiles for 2-D scanners).  For the per-ray
  // traversals round 3 re-measured the shapes with the frontier start in place (profiles/r03_find_tile_shapes.txt): 16 wide x 4 tall
  // is faster or equal in 10 of 12 (size, map) cells of kinds 23 / 2 -- C2 16.9 -> 16.5 us (sphere), 25.3 -> 24.5 us (room) -- and
  // neutral for pose batches (kind 24).
  const uint32_t max_th = packet ? 8u : 4u;
  uint32_t th = 1;
  while (th < H && th < max_th) th <<= 1;
  uint32_t twl = 0;
  while ((64u >> twl) > th) ++twl;
  return twl;  // tile = 2^twl wide, 64 >> twl tall
}

RMCL_INTERNAL rmclhip_status ensure_model_buffers(rmclhip_rcc* r, size_t n_total) {
  drop_moment_set(r);   // every find form comes through here first
  // only what the selected bundle carries (rmclhip_rcc_set_outputs); a deselected buffer keeps what it has
  if (r->out_mask & RMCLHIP_OUT_HITS) HIPCHK(r->d_hits.reserve(n_total));
  if (r->out_mask & RMCLHIP_OUT_RANGES) HIPCHK(r->d_ranges.reserve(n_total));
  if (r->out_mask & RMCLHIP_OUT_POINTS) HIPCHK(r->d_points.reserve(3 * n_total));
  if (r->out_mask & RMCLHIP_OUT_NORMALS) HIPCHK(r->d_normals.reserve(3 * n_total));
  if (r->out_mask & RMCLHIP_OUT_FACE_IDS) HIPCHK(r->d_face_ids.reserve(n_total));
  return RMCLHIP_OK;
}

// traversal kind of a launch of `nposes` scans (tools/latency_explore.py, tools/perf_explore.py)
RMCL_INTERNAL int find_variant(const rmclhip_rcc* r, uint32_t nposes) {
  if (r->variant != 15) return r->variant;
  if (nposes == 1u && r->tuned_kind != 0) return r->tuned_kind;   // measured on this operator's own map and model (rmclhip_rcc_autotune)
  if (nposes > 1u && r->tuned_batch_kind != 0) return r->tuned_batch_kind;
  const uint64_t rays = static_cast<uint64_t>(r->W) * r->H * nposes;
  if (rays <= 57344u) return 2;   // bound by the slowest ray's fetch chain: four lanes per ray (crossover measured between
                                  // 49152 rays -- quads 13.6 / 20.8 us vs 16.4 / 23.4 -- and 65536 -- 15.9 / 26.3 vs 16.3 / 23.7)
  // one lane per ray, starting at the map's FRONTIER instead of the root (traverse.hip.h frontier_start): from 65 536 to 262 144
  // rays kind 23 is the fastest or within 3 % of it on both benchmark maps (profiles/r03_find_variants_ab.txt), which replaces
  // round 2's three brackets (19 / 21 / 22) by one; larger launches and pose batches are bound by cache-line accesses and issue
  // slots: the 64-B quantised nodes of the FILTER's tree (leaves <= 2: the per-lane triangle loop is short).  At 262 144 rays the
  // sphere prefers 24 (24.9 vs 28.1 us) and the room 23 (38.9 vs 41.1); at 524 288 both prefer 24 (40.5 / 57.8 vs 43.6 / 59.3)
  if (rays <= 262144u) return 23;  // full-precision nodes, branch-free step, one-round-trip leaves, quad-finished tails, leaf trigger
  return 24;                       // quantised nodes of the filter's tree, 16 LDS rows, leaf trigger
}

// `kind`: the traversal the caller is about to launch when it is not the automatic rule's (enqueue_find_with_moments replaces 24 by
// 23); tree, frontier table, tile shape and pre-load bound all follow THAT kind (the tables of the filter's tree under a
// walk of the map's tree start rays at wrong nodes)
RMCL_INTERNAL void fill_find_params(rmclhip_rcc* r, FindParams& p, uint32_t nposes, int kind) {
  const int v = kind >= 0 ? kind : find_variant(r, nposes);
  std::memset(&p, 0, sizeof(p));
  p.nodes = r->map->d_nodes;
  p.qnodes = r->map->d_qnodes;
  p.cnodes = r->map->d_cnodes;
  p.cnodes16 = r->descent_wide ? r->map->d_cnodes16 : nullptr;
  p.tris = r->map->d_tris;
  p.n_nodes = r->map->info.n_nodes;
  p.frontier = r->map->d_frontier;
  p.n_frontier = r->map->n_frontier;
  {
    const BvhInfo& bi = r->map->info;
    p.scene_center = mk3(0.5f * (bi.bbox_min[0] + bi.bbox_max[0]), 0.5f * (bi.bbox_min[1] + bi.bbox_max[1]), 0.5f * (bi.bbox_min[2] + bi.bbox_max[2]));
    const float dx = bi.bbox_max[0] - bi.bbox_min[0], dy = bi.bbox_max[1] - bi.bbox_min[1], dz = bi.bbox_max[2] - bi.bbox_min[2];
    p.scene_half_diag = 0.5f * std::sqrt(dx * dx + dy * dy + dz * dz) + bi.pad;
  }
  p.model_tab = r->d_model_tab.p;
  p.W = r->W; p.H = r->H;
  p.tile_w_log2 = (r->tile_override > 0) ? static_cast<uint32_t>(r->tile_override - 1)
                  : ((r->tuned_tile > 0 && v != 0) ? static_cast<uint32_t>(r->tuned_tile - 1) : pick_tile_w_log2(r->H, v == 0, r->ang_aspect));
  const uint32_t tw = 1u << p.tile_w_log2, th = 64u >> p.tile_w_log2;
  p.tiles_x = (r->W + tw - 1) / tw;
  p.tiles_y = (r->H + th - 1) / th;
  p.tfar = r->range.max;
  p.orig_s = r->orig;
  p.pin_f[0] = r->pin_fc[0]; p.pin_f[1] = r->pin_fc[1]; p.pin_c[0] = r->pin_fc[2]; p.pin_c[1] = r->pin_fc[3];
  p.nposes = nposes;
  // bundle attribute selection (rmclhip_rcc_set_outputs): the kernel skips the stores of a null output
  p.hits = (r->out_mask & RMCLHIP_OUT_HITS) ? r->d_hits.p : nullptr;
  p.ranges = (r->out_mask & RMCLHIP_OUT_RANGES) ? r->d_ranges.p : nullptr;
  p.points = (r->out_mask & RMCLHIP_OUT_POINTS) ? r->d_points.p : nullptr;
  p.normals = (r->out_mask & RMCLHIP_OUT_NORMALS) ? r->d_normals.p : nullptr;
  p.face_ids = (r->out_mask & RMCLHIP_OUT_FACE_IDS) ? r->d_face_ids.p : nullptr;
  p.tile_planes = (r->tile_planes_ok && (nposes == 1u ? r->tuned_frontier : r->tuned_batch_frontier)) ? r->d_tile_planes.p : nullptr;
  {
    // kind 24 (and its frontier-less twin 22: rays on the quantised nodes, triangles in a per-lane loop) walks the FILTER's tree --
    // the same BVH2 cut at leaves of <= 2 instead of <= 4 triangles, the same record array (layout.h): pose batches 6-10 % faster
    // (profiles/r03_find_variants_ab.txt).  That tree has its own node numbering, hence its own frontier table.
    uint32_t need = r->map->info.stack_need;
    if ((v == 24 || v == 22) && r->map->d_qnodes_pf != nullptr) {
      p.qnodes = r->map->d_qnodes_pf;
      p.frontier = r->map->d_frontier_pf;
      p.n_frontier = r->map->n_frontier_pf;
      need = r->map->info.stack_need_pf;
    }
    // The frontier start pre-loads a lane's stack (up to 19 entries for kind 23, 12 for kind 24, more for the quad kind); map_upload's
    // stack_need <= 64 bounds a descent from the ROOT only.  From the frontier the descent may still push what the tree's deepest path
    // pushes, so the start may leave at most 64 - stack_need entries (traverse.hip.h frontier_start returns the root beyond that); a
    // tree that leaves no room for even two starts every ray at the root.
    p.descent_final_cap = r->descent_final_cap;
    p.xcd_mapping = (r->xcd_mapping_override >= 0) ? static_cast<uint32_t>(r->xcd_mapping_override) : r->tuned_xcd_mapping;
    p.descent_levels = (r->descent_levels & 0xFFu) | (r->descent_leaf_cap << 8);   // (traverse.hip.h frontier_descent_start: levels | most leaves per ray << 8)
    p.frontier_max_preload = (need < 64u) ? 64u - need : 0u;
    if (p.frontier_max_preload < 2u) p.tile_planes = nullptr;
  }
}

// The frontier start's plane table belongs to (model, tiling): rebuilt -- one small launch on the handle's stream -- by whatever
// changes either (the model setters, set_variant's tile shape), never inside a find (finds are captured into graphs).
RMCL_INTERNAL rmclhip_status rebuild_tile_planes(rmclhip_rcc* r, bool keep_tuning) {
  r->tile_planes_ok = false;
  if (!keep_tuning) {
    r->tuned_kind = r->tuned_batch_kind = 0;   // a measurement belongs to the model it was taken with
    r->tuned_frontier = r->tuned_batch_frontier = true;
    r->tuned_tile = 0;
    r->tuned_xcd_mapping = 0;
  }
  if (r->kind == kModelOnDn || r->kind == kModelNone || r->W == 0 || r->H == 0) return RMCLHIP_OK;
  FindParams p;
  fill_find_params(r, p, 1);
  HIPCHK(r->d_tile_planes.reserve(static_cast<size_t>(p.tiles_x) * p.tiles_y * 16u));
  HIPCHK(launch_tile_planes(p, r->kind, r->d_tile_planes.p, r->stream));
  r->tile_planes_ok = true;
  r->graph_dirty = true; r->fast_graph_dirty = true;
  return RMCLHIP_OK;
}

static rmclhip_status enqueue_find_with_moments(rmclhip_rcc* r, const xform& Tsm, float lo, float hi, float rho_cap, float tau_cap, uint32_t seq,
                                                bool epilogue_allowed);
static inline void learn_caps(rmclhip_rcc* r, float max_rho, float max_tau);
static inline void gate_band(const rmclhip_rcc* r, float centre, float* lo, float* hi);
static inline uint32_t next_seq(rmclhip_rcc* r);
static hipError_t wait_moments(rmclhip_rcc* r, uint32_t seq, float lo, float hi, float rho_cap, float tau_cap);

// `speculate` (out, nullable): set when the find was enqueued WITH the moment epilogue + publish for the computeCrossStatistics calls
// that will follow it (r->mset_pending, r->mset_seq, r->pend_*): the reference's caller loop (micp_localization.cpp:900-964)
// alternates find() and n x computeCrossStatistics(), so a find that was followed by such calls expects them again.
RMCL_INTERNAL rmclhip_status find_enqueue(rmclhip_rcc* r, const xform& Tbm, bool* speculate) {
  const size_t n = static_cast<size_t>(r->W) * r->H;
  r->n_model = static_cast<uint32_t>(n);
  r->nposes_last = 1;
  if (rmclhip_status st = ensure_model_buffers(r, n)) return st;
  if (speculate) {
    *speculate = false;
    // what the last loop met bounds what this one may meet
    if (r->ccs_since_find != 0u) learn_caps(r, r->ccs_max_rho, r->ccs_max_tau);
    r->ccs_loop = r->ccs_since_find != 0u;
    r->ccs_since_find = 0u; r->ccs_max_rho = 0.f; r->ccs_max_tau = 0.f;
    const int fv = find_variant(r, 1);
    if (r->ccs_loop && r->fast_mode == 1 && !r->fused_tail && r->n_dataset != 0u && (fv == 23 || fv == 32 || fv == 2) && r->ccs_last_maxd == r->ccs_last_maxd &&
        micp_outputs_selected(r)) {
      gate_band(r, r->ccs_last_maxd, &r->pend_lo, &r->pend_hi);
      r->pend_rho = r->fast_rho_cap; r->pend_tau = r->fast_tau_cap;
      r->mset_seq = next_seq(r);
      if (rmclhip_status st = enqueue_find_with_moments(r, xmul(Tbm, r->Tsb), r->pend_lo, r->pend_hi, r->pend_rho, r->pend_tau, r->mset_seq, true))
        return st;
      r->mset_pending = true;
      ++r->ccs_info.speculative_finds;
      *speculate = true;
      return RMCLHIP_OK;
    }
  }
  FindParams p;
  fill_find_params(r, p, 1);
  p.Tsm = xmul(Tbm, r->Tsb);
  p.Tms = xinv(p.Tsm);
  const int variant = find_variant(r, p.nposes);
  HIPCHK(launch_find(p, r->kind, variant, r->stream));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_find_async(rmclhip_rcc* r, const rmclhip_transform* Tbm_est) {
  ApiGuard guard_("rmclhip_rcc_find_async");
  if (!r || !Tbm_est) return fail(RMCLHIP_ERR_INVALID, "rcc_find: null");
  // RCCOptix.cpp:30-34: nothing to do for an empty model
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return RMCLHIP_OK;
  HIPCHK(hipSetDevice(r->ctx->device));
  bool spec = false;
  return find_enqueue(r, to_x(Tbm_est), &spec);   // (a speculating find's publish is awaited by the first computeCrossStatistics)
}

// Wait for a handle's stream the way the context's wait mode says (rmclhip_ctx_set_wait_mode): SPIN polls hipStreamQuery, then one
// hipStreamSynchronize (immediate) keeps the runtime's own view in order; BLOCK goes to hipStreamSynchronize at once.  20 ms of polling
// at most.  Measured late in round 3 (a synchronous 128x1024 find at the C ABI, median of 200): 32.0 us either way -- on this runtime
// hipStreamSynchronize spins for short waits itself, so for a find the mode only says whose loop burns the core; it is the polled
// completion TAG of the calls that return results (wait_done: computeCrossStatistics 21.5 vs 27.3 us) that the mode really moves.
// For the particle filter's short kernels the polling loop was SLOWER than hipStreamSynchronize (likelihood statistics 19 -> 31 us):
// those entry points call hipStreamSynchronize directly.
static hipError_t stream_wait(const rmclhip_ctx* ctx, hipStream_t stream) {
  if (!ctx->wait_block.load(std::memory_order_relaxed)) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return q;
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
  }
  return hipStreamSynchronize(stream);
}

// Wait for the end of what the handle's stream holds.  SPIN mode: a one-thread launch behind the chain stores a completion tag in pinned
// memory, which the host sees ~7 us before the stream's own completion signal (a synchronous 128x1024 find: 31.6 -> 24.5 us, round 4);
// the chain has ended -- kernel boundary -- when that launch runs, so its results are complete.  BLOCK mode: hipStreamSynchronize.
static hipError_t wait_chain_end(rmclhip_rcc* r);

rmclhip_status rmclhip_rcc_find(rmclhip_rcc* r, const rmclhip_transform* Tbm_est) {
  ApiGuard guard_("rmclhip_rcc_find");
  if (!r || !Tbm_est) return fail(RMCLHIP_ERR_INVALID, "rcc_find: null");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return RMCLHIP_OK;
  HIPCHK(hipSetDevice(r->ctx->device));
  const bool timed = r->kernel_timing;
  r->reduce_timing_pending = false;   // the events are reused below
  r->find_timing_pending = false;
  if (timed) HIPCHK(hipEventRecord(r->ev0, r->stream));
  bool spec = false;
  if (rmclhip_status st = find_enqueue(r, to_x(Tbm_est), &spec)) return st;
  if (timed) HIPCHK(hipEventRecord(r->ev1, r->stream));
  if (spec) {
    // the publish launch's tag says the find before it on this stream is complete as well -- and reaches the host sooner than the
    // stream's own completion does (see wait_done); the events are read when rmclhip_rcc_last_kernel_ms asks for them
    HIPCHK(wait_moments(r, r->mset_seq, r->pend_lo, r->pend_hi, r->pend_rho, r->pend_tau));
    r->find_timing_pending = timed;
    return RMCLHIP_OK;
  }
  if (timed) {
    HIPCHK(stream_wait(r->ctx, r->stream));
    HIPCHK(hipEventElapsedTime(&r->last_find_ms, r->ev0, r->ev1));
    return RMCLHIP_OK;
  }
  // (round 5, built, measured and removed: the find storing its own completion tag from its last wave instead of the one-thread kernel
  // behind it -- 39 to 70 us per call against 24: profiles/r05_sync_find_tag.txt)
  HIPCHK(wait_chain_end(r));
  return RMCLHIP_OK;
}

// squared search radius of a bounded closest-point query: everything with sqrtf(d2) <= max_dist must stay inside it (sqrtf rounds
// to nearest: d2 <= max_dist^2 (1 + 2^-22) covers every such d2), so hits -- and every output of a point that hits -- are those of
// the unbounded search
static float cpc_bound_d2(const rmclhip_rcc* r) {
  if (!r->cpc_bounded || !(r->max_dist >= 0.0f)) return 3.0e38f;
  const double m = static_cast<double>(r->max_dist);
  const double b = m * m * (1.0 + 1.0 / 1048576.0) + 1e-30;
  return b < 3.0e38 ? static_cast<float>(b) : 3.0e38f;
}

// The map's near grid: ~2 M cubic cells over the map's box (at most 256 per axis), each holding the record closest to its centre --
// one cold closest-point launch over the cell centres, once per map (a few ms; 8 MB), under the map's mutex: operators of one map may
// be used from different threads.
// `full`: every cell gets a record (the particle filter's closest-point mode queries beam END points, metres from any surface);
// otherwise cells farther than two coarse cell diagonals from the surface get none (scan points lie near it) -- the cheap build.
RMCL_INTERNAL rmclhip_status ensure_near_grid(rmclhip_map* m, hipStream_t stream, bool full, const NearGrid** out) {
  *out = nullptr;
  std::lock_guard<std::mutex> lock(m->grid_mtx);
  if (m->grid_slot[1].ready) { *out = &m->grid_slot[1].g; return RMCLHIP_OK; }   // the full grid serves every caller
  rmclhip_map::GridSlot& slot = m->grid_slot[full ? 1 : 0];
  if (slot.ready) { *out = &slot.g; return RMCLHIP_OK; }
  if (slot.failed) return RMCLHIP_OK;
  float ext[3];
  double vol = 1.0;
  for (int k = 0; k < 3; ++k) {
    ext[k] = std::max(m->info.bbox_max[k] - m->info.bbox_min[k], 1e-3f);
    ext[k] *= 1.02f;   // a thin margin: points of a scan lie ON the surface, i.e. on the box's faces
    vol *= ext[k];
  }
  if (!(vol > 0.0) || !std::isfinite(vol)) { slot.failed = true; return RMCLHIP_OK; }
  const float cell = static_cast<float>(std::cbrt(vol / 2.0e6));
  NearGrid g = {};
  size_t total = 1;
  for (int k = 0; k < 3; ++k) {
    g.n[k] = std::max(1u, std::min(256u, static_cast<uint32_t>(std::ceil(ext[k] / cell))));
    g.org[k] = 0.5f * (m->info.bbox_min[k] + m->info.bbox_max[k]) - 0.5f * ext[k];
    g.inv[k] = static_cast<float>(g.n[k]) / ext[k];
    total *= g.n[k];
  }
  uint32_t* d_cells = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_cells), total * sizeof(uint32_t));
  if (e != hipSuccess) { slot.failed = true; (void)hipGetLastError(); return RMCLHIP_OK; }   // (a map too large for the table simply runs without it)
  // coarse to fine: a grid of a quarter of the resolution first (its cells far from any surface are the expensive, unbounded queries:
  // 64 x fewer of them), then the full grid with every cell seeded from its coarse parent
  // (slot 0: cells farther than two coarse cell diagonals from the surface get no record: a query point there runs unseeded, as before)
  const float cdiag = 4.0f * cell * 1.7320508f;
  const float skip_d2 = full ? 3.0e38f : (2.0f * cdiag) * (2.0f * cdiag);
  NearGrid c = g;
  size_t ctotal = 1;
  for (int k = 0; k < 3; ++k) { c.n[k] = (g.n[k] + 3u) / 4u; c.inv[k] = g.inv[k] * static_cast<float>(c.n[k]) / static_cast<float>(g.n[k]); ctotal *= c.n[k]; }
  uint32_t* d_coarse = nullptr;
  e = hipMalloc(reinterpret_cast<void**>(&d_coarse), ctotal * sizeof(uint32_t));
  if (e == hipSuccess)
    e = launch_cpc_find(m->d_nodes, m->d_tris, nullptr, static_cast<uint32_t>(ctotal), 0.f, xidentity(), xidentity(), nullptr, nullptr, nullptr, nullptr,
                        nullptr, false, stream, nullptr, d_coarse, m->info.n_records, 3.0e38f, nullptr, &c);
  c.cells = d_coarse;
  if (e == hipSuccess)
    e = launch_cpc_find(m->d_nodes, m->d_tris, nullptr, static_cast<uint32_t>(total), 0.f, xidentity(), xidentity(), nullptr, nullptr, nullptr, nullptr,
                        nullptr, false, stream, nullptr, d_cells, m->info.n_records, 3.0e38f, &c, &g, skip_d2);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (d_coarse) (void)hipFree(d_coarse);
  if (e != hipSuccess) {
    (void)hipFree(d_cells);
    slot.failed = true;
    return fail(RMCLHIP_ERR_HIP, std::string("near grid: ") + hipGetErrorString(e));
  }
  g.cells = d_cells;
  slot.g = g;
  slot.ready = true;
  m->bytes += total * sizeof(uint32_t);
  *out = &slot.g;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_find_cpc(rmclhip_rcc* r, const rmclhip_transform* Tbm_est) {
  ApiGuard guard_("rmclhip_rcc_find_cpc");
  if (!r || !Tbm_est) return fail(RMCLHIP_ERR_INVALID, "rcc_find_cpc: null");
  if (r->n_dataset == 0) return RMCLHIP_OK;
  HIPCHK(hipSetDevice(r->ctx->device));
  // CPCEmbree.cpp:20-25: model buffers are sized like the DATASET (grow-only)
  const size_t n = r->n_dataset;
  if (rmclhip_status st = ensure_model_buffers(r, n)) return st;
  r->n_model = r->n_dataset;
  r->nposes_last = 1;
  const xform Tsm = xmul(to_x(Tbm_est), r->Tsb);
  const bool quad = (r->variant == 15) ? true : (r->variant == 2);  // four lanes per point read the child-major nodes
  // tracking: the record every point was closest to in the previous call of this operator bounds this call's search (same
  // results; rmclhip_rcc_set_cpc_tracking).  The records belong to one dataset of one size: anything else starts cold.
  const uint32_t* seed = nullptr;
  if (r->cpc_tracking) {
    HIPCHK(r->d_cpc_rec.reserve(n));
    if (r->d_cpc_rec.p != r->cpc_rec_ptr) { r->cpc_rec_ptr = r->d_cpc_rec.p; r->cpc_rec_n = 0; }   // (re)allocated
    if (r->cpc_rec_n == r->n_dataset && r->cpc_rec_pts == r->ds_pts) seed = r->d_cpc_rec.p;
  }
  const NearGrid* grid = nullptr;
  if (r->cpc_grid) { if (rmclhip_status gst = ensure_near_grid(r->map, r->stream, false, &grid)) return gst; }
  HIPCHK(launch_cpc_find(quad ? r->map->d_cnodes : r->map->d_nodes, r->map->d_tris, r->ds_pts, r->n_dataset,
                         r->max_dist, Tsm, xinv(Tsm), (r->out_mask & RMCLHIP_OUT_HITS) ? r->d_hits.p : nullptr,
                         (r->out_mask & RMCLHIP_OUT_RANGES) ? r->d_ranges.p : nullptr, (r->out_mask & RMCLHIP_OUT_POINTS) ? r->d_points.p : nullptr,
                         (r->out_mask & RMCLHIP_OUT_NORMALS) ? r->d_normals.p : nullptr,
                         (r->out_mask & RMCLHIP_OUT_FACE_IDS) ? r->d_face_ids.p : nullptr, quad, r->stream, seed, r->cpc_tracking ? r->d_cpc_rec.p : nullptr, r->map->info.n_records,
                         cpc_bound_d2(r), grid));
  if (r->cpc_tracking) { r->cpc_rec_n = r->n_dataset; r->cpc_rec_pts = r->ds_pts; }
  HIPCHK(wait_chain_end(r));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_cpc_tracking(rmclhip_rcc* r, int on) {
  ApiGuard guard_("rmclhip_rcc_set_cpc_tracking");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_cpc_tracking: null");
  r->cpc_tracking = on != 0;
  r->cpc_rec_n = 0;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_cpc_grid(rmclhip_rcc* r, int on) {
  ApiGuard guard_("rmclhip_rcc_set_cpc_grid");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_cpc_grid: null");
  r->cpc_grid = on != 0;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_set_cpc_bounded(rmclhip_rcc* r, int on) {
  ApiGuard guard_("rmclhip_rcc_set_cpc_bounded");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_cpc_bounded: null");
  r->cpc_bounded = on != 0;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_sync(rmclhip_rcc* r) {
  ApiGuard guard_("rmclhip_rcc_sync");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_sync: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  // SPIN mode: a one-thread launch behind whatever the stream holds stores a completion tag that the host sees several microseconds
  // before the stream's own completion signal (what the synchronous find does since round 4); BLOCK mode: hipStreamSynchronize
  if (hipStreamQuery(r->stream) == hipSuccess) return RMCLHIP_OK;
  HIPCHK(wait_chain_end(r));
  return RMCLHIP_OK;
}


RMCL_INTERNAL rmclhip_status reduce_enqueue(rmclhip_rcc* r, const xform& Tpre, const xform* Tpre_dev, float max_dist,
                                     uint32_t nposes, const ReduceTail& tail) {
  if (!micp_outputs_selected(r)) return fail(RMCLHIP_ERR_INVALID, kNeedMicpOutputs);
  const uint32_t n = (r->n_dataset < r->n_model) ? r->n_dataset : r->n_model;
  if (n == 0) return fail(RMCLHIP_ERR_INVALID, "computeCrossStatistics: empty dataset or model (call find first)");
  if (r->n_model != n && nposes > 1) return fail(RMCLHIP_ERR_INVALID, "batch reduction needs dataset size == model size");
  const uint32_t nb = reduce_num_blocks(n, nposes);
  HIPCHK(r->d_partials.reserve(static_cast<size_t>(nposes) * nb * 16));
  if (r->tickets_cap < nposes) {
    if (r->d_tickets) (void)hipFree(r->d_tickets);
    r->d_tickets = nullptr;
    r->tickets_cap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&r->d_tickets), sizeof(uint32_t) * nposes));
    HIPCHK(hipMemset(r->d_tickets, 0, sizeof(uint32_t) * nposes));
    r->tickets_cap = nposes;
  }
  ReduceParams p;
  std::memset(&p, 0, sizeof(p));
  p.dataset_points = r->ds_pts;
  p.dataset_mask = r->ds_has_mask ? r->ds_msk : nullptr;
  p.model_points = r->d_points.p;
  p.model_normals = r->d_normals.p;
  p.model_mask = r->d_hits.p;
  p.n = n;
  p.nposes = nposes;
  p.max_dist = max_dist;
  p.Tpre = Tpre;
  p.Tpre_dev = Tpre_dev;
  p.partials = r->d_partials.p;
  p.nblocks = nb;
  p.tickets = r->d_tickets;
  p.call = tail.call;
  p.Tsb = r->Tsb;
  p.Tbo = tail.Tbo;
  p.state = tail.state;
  p.stats_out = tail.stats_out;
  p.Tdelta_out = tail.Tdelta_out;
  p.tail_mode = r->fused_tail ? tail.mode : static_cast<uint32_t>(kTailNone);
  HIPCHK(launch_reduce_partials(p, r->stream));
  if (!r->fused_tail) {
    if (tail.mode == kTailStats) HIPCHK(launch_reduce_finalize(r->d_partials.p, nb, nposes, tail.stats_out, tail.done, tail.seq, r->stream));
    else if (tail.mode == kTailMicp) HIPCHK(launch_micp_step(r->d_partials.p, nb, r->Tsb, tail.Tbo, tail.call, tail.state, tail.state, r->stream));
    else if (tail.mode == kTailBatchSolve)
      HIPCHK(launch_batch_solve(r->d_partials.p, nb, nposes, r->Tsb, tail.Tdelta_out, tail.stats_out, r->stream));
  }
  return RMCLHIP_OK;
}

// Wait for the completion tag the LAST kernel of a chain stores in host-mapped memory after its results (kernels.hip
// publish_tag), instead of hipStreamSynchronize: the tag arrives ~9 us before the stream's completion signal has made its way
// through the runtime (measured on the MICP loop: 84 -> 75 us per correction).
// The tag carries {sequence number of the call, xor of every result word}: the result is accepted only when the sequence number is
// this call's AND the words the host reads add up to the tag's sum; otherwise polling continues.  (Round 2 polled a flag of constant
// value that the host cleared before each launch and read a stale result 1 call in ~10^4; the per-call sequence number is what cured
// it -- the checksum has not rejected a block since, profiles/r05_tag_handoff.txt -- and the xor remains as a belt.)  20 ms without an
// acceptable tag, or wait mode "block" (rmclhip_ctx_set_wait_mode), falls back to the stream.
static inline uint32_t xor_host(const void* p, size_t bytes) {
  const volatile uint32_t* w = static_cast<const volatile uint32_t*>(p);
  uint32_t x = 0;
  for (size_t i = 0; i < bytes / 4; ++i) x ^= w[i];
  return x;
}
// how often a poller saw ITS sequence number in the tag while the result words did not (yet) add up to the tag's sum: the event the
// checksum exists for (rmclhip_debug_tag_retries; profiles/r05_tag_handoff.txt)
RMCL_INTERNAL std::atomic<unsigned long long> g_tag_sum_retries{0};

static inline uint32_t next_seq(rmclhip_rcc* r) {
  if (++r->done_seq == 0u) r->done_seq = 1u;
  return r->done_seq;
}
static inline uint32_t done_sum(const DoneCheck& c) {
  uint32_t x = xor_host(c.base, c.base_bytes);
  if (c.code == nullptr || *c.code == 0u)
    for (int k = 0; k < 3; ++k) if (c.extra[k]) x ^= xor_host(c.extra[k], c.extra_bytes[k]);
  return x;
}
static hipError_t wait_done(const rmclhip_ctx* ctx, volatile const unsigned long long* tag, uint32_t seq, const DoneCheck& chk,
                            hipStream_t stream) {
  if (ctx->wait_block.load(std::memory_order_relaxed)) return hipStreamSynchronize(stream);
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
  for (uint32_t spins = 0;; ++spins) {
    const unsigned long long t = *tag;
    if (static_cast<uint32_t>(t) == seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      if (done_sum(chk) == static_cast<uint32_t>(t >> 32)) return hipSuccess;
      g_tag_sum_retries.fetch_add(1, std::memory_order_relaxed);   // this call's tag, but the words read do not add up to it (yet)
    }
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() > t_end) return hipStreamSynchronize(stream);
  }
}

static hipError_t wait_chain_end(rmclhip_rcc* r) {
  if (r->ctx->wait_block.load(std::memory_order_relaxed)) return hipStreamSynchronize(r->stream);
  const uint32_t seq = next_seq(r);
  if (const hipError_t e = launch_host_tag(r->h_done_dev, seq, r->stream)) return e;
  DoneCheck none;
  return wait_done(r->ctx, r->h_done, seq, none, r->stream);
}

static float adaptive_max_dist(const rmclhip_rcc* r, double p) {
  // CorrespondencesCPU.cpp:21-23 (float operands, double arithmetic, float store)
  return static_cast<float>(static_cast<double>(r->max_dist) * (1.0 - p) +
                            static_cast<double>(r->adaptive_max_dist_min) * p);
}

// ---- the gate-stable moments on the host (micp_host.h; kernels.hip k_micp_publish) -----------------------------------------------
// The band of max_dist' values a speculating find classifies for: max_dist' = max_dist (1 - p) + adaptive_max_dist_min p moves with
// the node's convergence_progress_ from one correction to the next (micp_localization.cpp:988-1007), the find does not know the
// next value, so it takes +-8 % around the last one, clipped to what the two parameters allow.  A max_dist' outside the band
// costs the first computeCrossStatistics of that correction one moment pass of its own (what every call cost before round 4).
static inline void gate_band(const rmclhip_rcc* r, float centre, float* lo, float* hi) {
  const float a = std::min(r->max_dist, r->adaptive_max_dist_min), b = std::max(r->max_dist, r->adaptive_max_dist_min);
  *lo = std::max(a, 0.92f * centre);
  *hi = std::min(b, 1.08f * centre);
  if (!(*lo <= centre)) *lo = centre;   // (centre outside [a, b]: parameters changed since; also NaN-safe)
  if (!(*hi >= centre)) *hi = centre;
}

// wait for the tag of a publish launch and take a verified copy of the block (see wait_done for why the sum is checked)
static hipError_t wait_moments(rmclhip_rcc* r, uint32_t seq, float lo, float hi, float rho_cap, float tau_cap) {
  const MicpHostBlock* hb = r->h_mom;
  auto block_sum = [hb]() -> uint32_t {
    uint32_t x = xor_host(hb->mom, sizeof(hb->mom));
    const uint32_t code = *reinterpret_cast<const volatile uint32_t*>(&hb->code);
    const uint32_t n = *reinterpret_cast<const volatile uint32_t*>(&hb->n_uncertain);
    x ^= code ^ n;
    if (n <= kMicpHostMaxUnc) x ^= xor_host(hb->unc, static_cast<size_t>(n) * 9u * sizeof(float));   // (written whenever they fit)
    return x;
  };
  hipError_t e = hipSuccess;
  bool unverified = false;
  if (r->ctx->wait_block.load(std::memory_order_relaxed)) e = hipStreamSynchronize(r->stream);
  else {
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
    volatile const unsigned long long* tag = r->h_done;
    for (uint32_t spins = 0;; ++spins) {
      const unsigned long long t = *tag;
      if (static_cast<uint32_t>(t) == seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        if (block_sum() == static_cast<uint32_t>(t >> 32)) break;
        g_tag_sum_retries.fetch_add(1, std::memory_order_relaxed);
      }
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#endif
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() > t_end) {
        // 20 ms without this sequence number: either the device is that slow, or a later launch of this handle has already
        // replaced the tag (h_done is shared by every tagged launch).  Drain the stream, then accept the block only if the tag
        // still is this publish's and its sum matches; otherwise the caller takes the streaming reduction
        e = hipStreamSynchronize(r->stream);
        const unsigned long long t2 = *tag;
        std::atomic_thread_fence(std::memory_order_acquire);
        unverified = !(static_cast<uint32_t>(t2) == seq && block_sum() == static_cast<uint32_t>(t2 >> 32));
        break;
      }
    }
  }
  r->mset_pending = false;
  if (e != hipSuccess || unverified) { r->mset.valid = false; return e; }
  MicpMomentSet& ms = r->mset;
  std::memcpy(ms.mom, hb->mom, sizeof(ms.mom));
  ms.gate_lo = lo; ms.gate_hi = hi; ms.rho_cap = rho_cap; ms.tau_cap = tau_cap;
  ms.n_unc = hb->n_uncertain;
  ms.valid = (hb->code == 0u && ms.n_unc <= kMicpHostMaxUnc);
  if (ms.valid) ms.set_undecided(hb->unc, ms.n_unc);   // (component-major copy, padded for the eight-lane sums)
  return hipSuccess;
}

// caps of the next moment set from the pre-transforms the last loop met (as rmclhip_rcc_correct_once learns them)
static inline void learn_caps(rmclhip_rcc* r, float max_rho, float max_tau) {
  r->fast_rho_cap = std::max(0.002f, std::max(2.0f * max_rho, 0.9f * r->fast_rho_cap));
  r->fast_tau_cap = std::max(0.005f, std::max(2.0f * max_tau, 0.9f * r->fast_tau_cap));
  r->fast_info.rho_cap = r->fast_rho_cap;
  r->fast_info.tau_cap = r->fast_tau_cap;
}

// find + moment epilogue + publish on the handle's stream (kinds 23 / 2), or find + moment pass + publish (any other kind);
// the caller waits with wait_moments(seq, ...)
static rmclhip_status enqueue_find_with_moments(rmclhip_rcc* r, const xform& Tsm, float lo, float hi, float rho_cap, float tau_cap, uint32_t seq,
                                                bool epilogue_allowed) {
  const uint32_t nred = (r->n_dataset < r->n_model) ? r->n_dataset : r->n_model;
  int fv = find_variant(r, 1);
  // scans above 262 144 rays would take kind 24, which has no moment epilogue: the separate moment pass over half a million
  // correspondences costs more (~40 us) than kind 23 loses against kind 24 there (~3 us) -- a correction of a 256 x 2048 scan 93 -> 6x us
  if (epilogue_allowed && fv == 24 && r->variant == 15) fv = 23;
  FindParams fp;
  fill_find_params(r, fp, 1, fv);   // the tree and tables of the kind that RUNS
  fp.Tsm = Tsm;
  fp.Tms = xinv(Tsm);
  r->last_moment_find_kind = fv;
  r->last_moment_find_tiled = epilogue_allowed && (fv == 23 || fv == 32 || fv == 2);
  if (epilogue_allowed && (fv == 23 || fv == 32 || fv == 2)) {
    const uint32_t nb = find_moments_blocks(fp, fv), wpb = (fv == 2) ? 1u : 4u;   // mask words per workgroup
    HIPCHK(r->d_fast_partials.reserve(static_cast<size_t>(nb) * kMicpFastMoments));
    HIPCHK(r->d_fast_mask.reserve(static_cast<size_t>(nb) * wpb));
    fp.mom_dataset_points = r->ds_pts;
    fp.mom_dataset_mask = r->ds_has_mask ? r->ds_msk : nullptr;
    fp.mom_n = nred;
    fp.mom_gate_lo = lo; fp.mom_gate_hi = hi; fp.mom_rho_cap = rho_cap; fp.mom_tau_cap = tau_cap;
    fp.mom_partials = r->d_fast_partials.p;
    fp.mom_unc_mask = r->d_fast_mask.p;
    if (!r->d_fold_rows) {
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&r->d_fold_rows), kMicpFoldBlocks * kMicpFastMoments * sizeof(double) + kMicpFoldBlocks * sizeof(uint32_t)));
      HIPCHK(hipMemset(r->d_fold_rows, 0, kMicpFoldBlocks * kMicpFastMoments * sizeof(double) + kMicpFoldBlocks * sizeof(uint32_t)));
      HIPCHK(hipDeviceSynchronize());
      r->d_fold_flags = reinterpret_cast<uint32_t*>(r->d_fold_rows + kMicpFoldBlocks * kMicpFastMoments);
    }
    r->last_fast_rows = nb; r->last_fast_words = wpb * nb;
    HIPCHK(launch_find_moments(fp, r->kind, fv, r->stream));
    HIPCHK(launch_micp_publish_tiled(r->ds_pts, r->d_points.p, r->d_normals.p, nred, nb, r->d_fast_partials.p, r->d_fast_mask.p, r->W,
                                     fp.tiles_x, fp.tile_w_log2, wpb, r->h_mom_dev, r->h_done_dev, seq, r->d_fold_rows, r->d_fold_flags,
                                     r->stream));
  } else {
    HIPCHK(r->d_fast_partials.reserve(static_cast<size_t>(micp_fast_blocks(nred)) * kMicpFastMoments));
    HIPCHK(r->d_fast_mask.reserve((static_cast<size_t>(nred) + 63u) / 64u));
    r->last_fast_rows = micp_fast_blocks(nred); r->last_fast_words = (nred + 63u) / 64u;
    MicpCallLite cl{};
    cl.gate_lo = lo; cl.gate_hi = hi; cl.max_dist = hi; cl.rho_cap = rho_cap; cl.tau_cap = tau_cap; cl.seq = seq;
    HIPCHK(launch_find(fp, r->kind, fv, r->stream));
    HIPCHK(launch_micp_moments_publish(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p, nred,
                                       r->d_fast_partials.p, r->d_fast_mask.p, cl, r->h_mom_dev, r->h_done_dev, r->stream));
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_compute_cross_statistics(rmclhip_rcc* r, const rmclhip_transform* T_snew_sold,
                                                    double convergence_progress, rmclhip_cross_statistics* out) {
  ApiGuard guard_("rmclhip_rcc_compute_cross_statistics");
  if (!r || !T_snew_sold || !out) return fail(RMCLHIP_ERR_INVALID, "computeCrossStatistics: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  if (r->nposes_last != 1) return fail(RMCLHIP_ERR_INVALID, "computeCrossStatistics: last find was a batch");
  if (!micp_outputs_selected(r)) return fail(RMCLHIP_ERR_INVALID, kNeedMicpOutputs);
  // ---- from the moments of this find's correspondences, when a set covers (pre-transform, max_dist'): no launch, no wait.
  // The reference's caller (micp_localization.cpp:915-964) calls this once per sensor and iteration on FIXED correspondences.
  {
    const xform Tpre = to_x(T_snew_sold);
    const float maxd = adaptive_max_dist(r, convergence_progress);
    const float rho = micp_rho(Tpre), tau = micp_tau(Tpre);
    ++r->ccs_since_find;
    r->ccs_last_maxd = maxd;
    r->ccs_max_rho = std::max(r->ccs_max_rho, rho);
    r->ccs_max_tau = std::max(r->ccs_max_tau, tau);
    const uint32_t nred = (r->n_dataset < r->n_model) ? r->n_dataset : r->n_model;
    if (r->fast_mode != 0 && !r->fused_tail && nred != 0 && maxd == maxd) {
      ++r->ccs_info.calls;
      if (r->mset_pending) HIPCHK(wait_moments(r, r->mset_seq, r->pend_lo, r->pend_hi, r->pend_rho, r->pend_tau));
      if (!micp_set_covers(r->mset, Tpre, maxd) && r->mset_passes < 2u && rho == rho && tau == tau) {
        // no covering set (the find did not speculate, or max_dist' / the pre-transform left what it speculated for): ONE moment
        // pass over the find's outputs now -- costs what the streaming reduction below costs -- serves the rest of the loop
        ++r->mset_passes;
        ++r->ccs_info.passes;
        float lo, hi;
        gate_band(r, maxd, &lo, &hi);
        const float rho_cap = std::max(r->fast_rho_cap, 2.0f * rho), tau_cap = std::max(r->fast_tau_cap, 2.0f * tau);
        HIPCHK(r->d_fast_partials.reserve(static_cast<size_t>(micp_fast_blocks(nred)) * kMicpFastMoments));
        HIPCHK(r->d_fast_mask.reserve((static_cast<size_t>(nred) + 63u) / 64u));
        r->last_fast_rows = micp_fast_blocks(nred); r->last_fast_words = (nred + 63u) / 64u;
        MicpCallLite cl{};
        cl.gate_lo = lo; cl.gate_hi = hi; cl.max_dist = maxd; cl.rho_cap = rho_cap; cl.tau_cap = tau_cap; cl.seq = next_seq(r);
        HIPCHK(launch_micp_moments_publish(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p, nred,
                                           r->d_fast_partials.p, r->d_fast_mask.p, cl, r->h_mom_dev, r->h_done_dev, r->stream));
        HIPCHK(wait_moments(r, cl.seq, lo, hi, rho_cap, tau_cap));
        if (!r->mset.valid) r->mset_passes = 2u;   // too many undecided correspondences: a second pass would find as many
      }
      if (micp_set_covers(r->mset, Tpre, maxd)) {
        ++r->ccs_info.from_moments;
        from_cs(micp_statistics_from_set(r->mset, Tpre, maxd), out);
        return RMCLHIP_OK;
      }
    }
  }
  ReduceTail tail;
  tail.mode = kTailStats;
  tail.stats_out = r->h_stats_dev;  // host-mapped: the finalize launch writes the 64-B result straight to the host
  const bool polled = !r->fused_tail;
  if (polled) { tail.done = r->h_done_dev; tail.seq = next_seq(r); }
  const bool timed = r->kernel_timing;
  r->find_timing_pending = false;   // the events are reused
  r->reduce_timing_pending = false;
  if (timed) HIPCHK(hipEventRecord(r->ev0, r->stream));
  if (rmclhip_status st = reduce_enqueue(r, to_x(T_snew_sold), nullptr, adaptive_max_dist(r, convergence_progress), 1, tail))
    return st;
  if (timed) HIPCHK(hipEventRecord(r->ev1, r->stream));
  if (polled) {
    DoneCheck chk; chk.base = &r->h_stats[0]; chk.base_bytes = sizeof(cstats);
    HIPCHK(wait_done(r->ctx, r->h_done, tail.seq, chk, r->stream));
    r->reduce_timing_pending = timed;   // the events are read when rmclhip_rcc_last_kernel_ms asks for them
  } else {
    HIPCHK(hipStreamSynchronize(r->stream));
    if (timed) HIPCHK(hipEventElapsedTime(&r->last_reduce_ms, r->ev0, r->ev1));
  }
  from_cs(r->h_stats[0], out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_download(rmclhip_rcc* r, uint8_t* hits, float* ranges, float* points, float* normals,
                                    uint32_t* face_ids) {
  ApiGuard guard_("rmclhip_rcc_download");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_download: null");
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(hipStreamSynchronize(r->stream));
  const size_t n = static_cast<size_t>(r->n_model) * (r->nposes_last ? r->nposes_last : 1);
  if (n == 0) return RMCLHIP_OK;
  // an attribute that no find of this size has written yet (deselected all along: rmclhip_rcc_set_outputs) has nothing to hand out
  if ((hits && r->d_hits.cap < n) || (ranges && r->d_ranges.cap < n) || (points && r->d_points.cap < 3 * n) ||
      (normals && r->d_normals.cap < 3 * n) || (face_ids && r->d_face_ids.cap < n))
    return fail(RMCLHIP_ERR_INVALID, "rcc_download: a requested attribute was never selected for a find of this size (rmclhip_rcc_set_outputs)");
  if (hits) HIPCHK(hipMemcpy(hits, r->d_hits.p, n, hipMemcpyDeviceToHost));
  if (ranges) HIPCHK(hipMemcpy(ranges, r->d_ranges.p, n * sizeof(float), hipMemcpyDeviceToHost));
  if (points) HIPCHK(hipMemcpy(points, r->d_points.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
  if (normals) HIPCHK(hipMemcpy(normals, r->d_normals.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
  if (face_ids) HIPCHK(hipMemcpy(face_ids, r->d_face_ids.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_device_views(rmclhip_rcc* r, const uint8_t** hits, const float** ranges,
                                        const float** points, const float** normals, const uint32_t** face_ids,
                                        uint32_t* n) {
  ApiGuard guard_("rmclhip_rcc_device_views");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_device_views: null");
  // (a buffer too small for the last find belongs to an attribute no find of that size selected: no view of it)
  const size_t nt = static_cast<size_t>(r->n_model) * (r->nposes_last ? r->nposes_last : 1);
  if (hits) *hits = (r->d_hits.cap >= nt) ? r->d_hits.p : nullptr;
  if (ranges) *ranges = (r->d_ranges.cap >= nt) ? r->d_ranges.p : nullptr;
  if (points) *points = (r->d_points.cap >= 3 * nt) ? r->d_points.p : nullptr;
  if (normals) *normals = (r->d_normals.cap >= 3 * nt) ? r->d_normals.p : nullptr;
  if (face_ids) *face_ids = (r->d_face_ids.cap >= nt) ? r->d_face_ids.p : nullptr;
  if (n) *n = static_cast<uint32_t>(nt);
  return RMCLHIP_OK;
}

// ---- the rmagine-level Simulator interface (rmclhip.h) ----------------------------------------------------------------------
rmclhip_status rmclhip_rcc_set_outputs(rmclhip_rcc* r, uint32_t mask) {
  ApiGuard guard_("rmclhip_rcc_set_outputs");
  if (!r) return fail(RMCLHIP_ERR_INVALID, "rcc_set_outputs: null");
  if (mask == 0u || (mask & ~RMCLHIP_OUT_ALL) != 0u) return fail(RMCLHIP_ERR_INVALID, "rcc_set_outputs: mask must name at least one of the five attributes and nothing else");
  if (mask != r->out_mask) {
    r->out_mask = mask;
    drop_moment_set(r);   // (the set summarises buffers the next find may no longer write)
    r->graph_dirty = true; r->fast_graph_dirty = true;   // the captured chains hold the find's output pointers by value
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_get_outputs(const rmclhip_rcc* r, uint32_t* mask) {
  if (!r || !mask) return fail(RMCLHIP_ERR_INVALID, "rcc_get_outputs: null");
  *mask = r->out_mask;
  return RMCLHIP_OK;
}

// Simulator::simulate into the caller's bundle: the find kernel with the caller's pointers as its outputs.  One pose from the host goes by
// value (no copy, no compose launch: what rmclhip_rcc_find does); batches and device-resident poses are composed with Tsb on the device.
static rmclhip_status simulate_enqueue(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes, int Tbm_is_device,
                                       const rmclhip_bundle_views* out) {
  if (nposes > 32768) return fail(RMCLHIP_ERR_UNSUPPORTED, "simulate: at most 32768 poses per call");
  FindParams p;
  fill_find_params(r, p, nposes);
  p.hits = out->hits_dev; p.ranges = out->ranges_dev; p.points = out->points_xyz_dev; p.normals = out->normals_xyz_dev;
  p.face_ids = out->face_ids_dev;
  if (nposes == 1u && !Tbm_is_device) {
    p.Tsm = xmul(to_x(Tbm), r->Tsb);
    p.Tms = xinv(p.Tsm);
  } else {
    HIPCHK(r->d_Tsm.reserve(nposes)); HIPCHK(r->d_Tms.reserve(nposes));
    const xform* src = reinterpret_cast<const xform*>(Tbm);
    if (!Tbm_is_device) {
      HIPCHK(r->d_Tbm.reserve(nposes));
      // (pageable source: the copy has staged the caller's poses when it returns; the stream orders the launches behind the DMA)
      HIPCHK(hipMemcpyAsync(r->d_Tbm.p, Tbm, sizeof(xform) * nposes, hipMemcpyHostToDevice, r->stream));
      src = r->d_Tbm.p;
    }
    HIPCHK(launch_compose_poses(src, r->Tsb, r->d_Tsm.p, r->d_Tms.p, nposes, r->stream));
    p.Tsm_arr = r->d_Tsm.p;
    p.Tms_arr = r->d_Tms.p;
  }
  if (rmclhip_status st = batch_order_enqueue(r, p, find_variant(r, nposes))) return st;   // (batches: world order)
  HIPCHK(launch_find(p, r->kind, find_variant(r, nposes), r->stream));
  return RMCLHIP_OK;
}

static rmclhip_status simulate_check(const char* who, rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes,
                                     const rmclhip_bundle_views* out, bool* nothing_to_do) {
  *nothing_to_do = false;
  if (!r || !out || (!Tbm && nposes)) return fail(RMCLHIP_ERR_INVALID, std::string(who) + ": null");
  if (nposes == 0 || r->kind == kModelNone || r->W == 0 || r->H == 0) { *nothing_to_do = true; return RMCLHIP_OK; }
  if (!out->hits_dev && !out->ranges_dev && !out->points_xyz_dev && !out->normals_xyz_dev && !out->face_ids_dev)
    return fail(RMCLHIP_ERR_INVALID, std::string(who) + ": the bundle carries no attribute");
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_simulate_async(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes, int Tbm_is_device,
                                          const rmclhip_bundle_views* out) {
  ApiGuard guard_("rmclhip_rcc_simulate_async");
  bool nothing = false;
  if (rmclhip_status st = simulate_check("simulate_async", r, Tbm, nposes, out, &nothing)) return st;
  if (nothing) return RMCLHIP_OK;
  HIPCHK(hipSetDevice(r->ctx->device));
  return simulate_enqueue(r, Tbm, nposes, Tbm_is_device, out);
}

rmclhip_status rmclhip_rcc_simulate(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes, int Tbm_is_device,
                                    const rmclhip_bundle_views* out) {
  ApiGuard guard_("rmclhip_rcc_simulate");
  bool nothing = false;
  if (rmclhip_status st = simulate_check("simulate", r, Tbm, nposes, out, &nothing)) return st;
  if (nothing) return RMCLHIP_OK;
  HIPCHK(hipSetDevice(r->ctx->device));
  if (rmclhip_status st = simulate_enqueue(r, Tbm, nposes, Tbm_is_device, out)) return st;
  HIPCHK(wait_chain_end(r));
  return RMCLHIP_OK;
}

// rm::statistics_p2l on caller-owned device views: the operator's streaming reduction with the context's own scratch
rmclhip_status rmclhip_statistics_p2l(rmclhip_ctx* ctx, const rmclhip_transform* Tpre, const float* dataset_points,
                                      const uint8_t* dataset_mask, const float* model_points, const float* model_normals,
                                      const uint8_t* model_mask, uint32_t n, float max_dist, rmclhip_cross_statistics* out) {
  ApiGuard guard_("rmclhip_statistics_p2l");
  if (!ctx || !Tpre || !out) return fail(RMCLHIP_ERR_INVALID, "statistics_p2l: null");
  if (n == 0) { from_cs(cs_identity(), out); return RMCLHIP_OK; }   // CrossStatistics::Identity(): nothing measured
  if (!dataset_points || !model_points || !model_normals) return fail(RMCLHIP_ERR_INVALID, "statistics_p2l: null view");
  std::lock_guard<std::mutex> lock(ctx->p2l_mtx);
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->p2l_stream == nullptr) {
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ctx->p2l_h_stats), sizeof(cstats), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->p2l_h_stats_dev), ctx->p2l_h_stats, 0);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&ctx->p2l_h_done), sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->p2l_h_done_dev), ctx->p2l_h_done, 0);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->p2l_tickets), sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(ctx->p2l_tickets, 0, sizeof(uint32_t));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      if (ctx->p2l_h_stats) (void)hipHostFree(ctx->p2l_h_stats);
      if (ctx->p2l_h_done) (void)hipHostFree(ctx->p2l_h_done);
      if (ctx->p2l_tickets) (void)hipFree(ctx->p2l_tickets);
      ctx->p2l_h_stats = nullptr; ctx->p2l_h_done = nullptr; ctx->p2l_tickets = nullptr;
      (void)hipStreamDestroy(s);
      return fail(RMCLHIP_ERR_HIP, std::string("statistics_p2l: ") + hipGetErrorString(e));
    }
    *ctx->p2l_h_done = 0ull;
    ctx->p2l_stream = s;   // last: the destructor frees the scratch iff the stream exists
  }
  const uint32_t nb = reduce_num_blocks(n, 1);
  if (ctx->p2l_partials_cap < static_cast<size_t>(nb) * 16u) {
    if (ctx->p2l_partials) (void)hipFree(ctx->p2l_partials);
    ctx->p2l_partials = nullptr; ctx->p2l_partials_cap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&ctx->p2l_partials), static_cast<size_t>(nb) * 16u * sizeof(double)));
    ctx->p2l_partials_cap = static_cast<size_t>(nb) * 16u;
  }
  ReduceParams p;
  std::memset(&p, 0, sizeof(p));
  p.dataset_points = dataset_points; p.dataset_mask = dataset_mask;
  p.model_points = model_points; p.model_normals = model_normals; p.model_mask = model_mask;
  p.n = n; p.nposes = 1; p.max_dist = max_dist;
  p.Tpre = to_x(Tpre);
  p.partials = ctx->p2l_partials; p.nblocks = nb;
  p.tickets = ctx->p2l_tickets;
  p.Tsb = xidentity(); p.Tbo = xidentity();
  p.tail_mode = static_cast<uint32_t>(kTailNone);
  if (++ctx->p2l_seq == 0u) ctx->p2l_seq = 1u;
  HIPCHK(launch_reduce_partials(p, ctx->p2l_stream));
  HIPCHK(launch_reduce_finalize(ctx->p2l_partials, nb, 1, ctx->p2l_h_stats_dev, ctx->p2l_h_done_dev, ctx->p2l_seq, ctx->p2l_stream));
  DoneCheck chk; chk.base = ctx->p2l_h_stats; chk.base_bytes = sizeof(cstats);
  HIPCHK(wait_done(ctx, ctx->p2l_h_done, ctx->p2l_seq, chk, ctx->p2l_stream));
  from_cs(*ctx->p2l_h_stats, out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_correct_once(rmclhip_rcc* r, const rmclhip_transform* Tom_, const rmclhip_transform* Tbo_,
                                        uint32_t n_iter, double convergence_progress, int refind_each_iteration,
                                        rmclhip_transform* T_out, rmclhip_cross_statistics* stats_out) {
  ApiGuard guard_("rmclhip_rcc_correct_once");
  if (!r || !Tom_ || !Tbo_ || !T_out) return fail(RMCLHIP_ERR_INVALID, "correct_once: null");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "correct_once: no sensor model");
  if (!micp_outputs_selected(r)) return fail(RMCLHIP_ERR_INVALID, kNeedMicpOutputs);
  HIPCHK(hipSetDevice(r->ctx->device));
  const xform Tom = to_x(Tom_), Tbo = to_x(Tbo_);
  const float maxd = adaptive_max_dist(r, convergence_progress);
  if (!refind_each_iteration) {
    // schedule (R), micp_localization.cpp:900-964: 1 find, n_iter x (reduce + solve); nothing returns
    // to the host until the end: the pre-transform of iteration i+1 is produced on the device.
    // The launch chain is captured ONCE into a hipGraph (the kernels read the per-call pose / frames /
    // max_dist from d_call, refreshed by the graph's first node), so a correction costs one graph launch
    // instead of 2 + 2*n_iter host launches (~5 us each, which left the GPU idle between these tiny kernels).
    const size_t n = static_cast<size_t>(r->W) * r->H;
    r->n_model = static_cast<uint32_t>(n);
    r->nposes_last = 1;
    if (rmclhip_status st = ensure_model_buffers(r, n)) return st;
    const uint32_t nred = (r->n_dataset < r->n_model) ? r->n_dataset : r->n_model;
    if (nred == 0) return fail(RMCLHIP_ERR_INVALID, "correct_once: empty dataset");
    HIPCHK(r->d_partials.reserve(std::max<size_t>(static_cast<size_t>(reduce_num_blocks(nred, 1)) * 32, 2u * 256u * 16u)));
    if (!r->d_loop_barrier) {
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&r->d_loop_barrier), sizeof(uint32_t)));
      HIPCHK(hipMemset(r->d_loop_barrier, 0, sizeof(uint32_t)));
    }
    {
      ReduceTail none;  // allocates the ticket buffer outside the capture
      (void)none;
      if (r->tickets_cap < 1) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&r->d_tickets), sizeof(uint32_t)));
        HIPCHK(hipMemset(r->d_tickets, 0, sizeof(uint32_t)));
        r->tickets_cap = 1;
      }
    }
    r->h_call->Tsm = xmul(xmul(Tom, Tbo), r->Tsb);
    r->h_call->Tms = xinv(r->h_call->Tsm);
    r->h_call->Tsb = r->Tsb;
    r->h_call->Tbo = Tbo;
    r->h_call->max_dist = maxd;
    r->h_call->gate_lo = maxd; r->h_call->gate_hi = maxd;
    r->h_call->rho_cap = r->fast_rho_cap;
    r->h_call->tau_cap = r->fast_tau_cap;
    r->h_call->seq = next_seq(r);
    // ---- moment form first (kernels.hip "gate-stable moment form"); any outcome other than "done" falls through to the
    // per-iteration form below, which recomputes the correction from scratch
    // (the device loops need >= 2 iterations to pay for their moment pass; the host form -- two launches, no reduction launch at all -- serves 1 as well)
    const bool fast_eligible = r->fast_mode != 0 && r->loop_blocks == 0 && !r->fused_tail && n_iter >= ((r->fast_mode == 1) ? 1u : 2u);
    bool fast_tried = false;
    if (fast_eligible && r->fast_holdoff > 0u) --r->fast_holdoff;
    else if (fast_eligible) {
      fast_tried = true;
      HIPCHK(r->d_fast_partials.reserve(static_cast<size_t>(micp_fast_blocks(nred)) * kMicpFastMoments));
      HIPCHK(r->d_fast_mask.reserve((static_cast<size_t>(nred) + 63u) / 64u));
      rmclhip_rcc::MicpKey key;
      std::memset(&key, 0, sizeof(key));
      key.n_iter = n_iter; key.W = r->W; key.H = r->H; key.n_dataset = r->n_dataset;
      key.kind = static_cast<int>(r->kind); key.variant = r->variant; key.tile = r->tile_override;
      key.fused = 0; key.has_mask = r->ds_has_mask ? 1 : 0;
      key.ptrs[0] = r->d_points.p; key.ptrs[1] = r->ds_pts; key.ptrs[2] = r->d_fast_partials.p;
      key.ptrs[3] = r->d_model_tab.p; key.ptrs[4] = r->ds_msk; key.ptrs[5] = r->d_fast_mask.p;
      MicpFastStatus fs{};
      bool fs_ready = false;     // the host ran the iterations: `fs` (and r->h_state) are final, nothing to wait for
      if (!r->use_graph || r->fast_mode != 2) {
        // direct launches (fast_mode 2 replays find + moment pass + device loop from a hipGraph, A/B) with their per-call data BY VALUE --
        // no H2D copy node, no graph launch (a graph replay costs the host 10-16 us whatever it holds)
        FindParams fp;
        fill_find_params(r, fp, 1);
        fp.Tsm = r->h_call->Tsm;
        fp.Tms = r->h_call->Tms;
        MicpCallLite cl;
        cl.Tsb = r->Tsb; cl.Tbo = Tbo; cl.max_dist = maxd; cl.rho_cap = r->fast_rho_cap; cl.tau_cap = r->fast_tau_cap;
        cl.gate_lo = maxd; cl.gate_hi = maxd;
        cl.seq = r->h_call->seq;
        const int fv = find_variant(r, 1);
        const bool tiled = r->fast_mode != 3 && (fv == 23 || fv == 32 || fv == 2);   // the find forms the moments in its epilogue
        const uint32_t nb = tiled ? find_moments_blocks(fp, fv) : micp_fast_blocks(nred), wpb = (fv == 2) ? 1u : 4u;
        bool device_loop = r->fast_mode != 1;
        if (r->fast_mode == 1) {
          // ---- round 4 default: TWO launches (find with the moment epilogue; fold + publish), the iterations on the HOST from the 82
          // moments + the undecided correspondences (micp_host.h): ~0.5 us per iteration instead of ~2.7 us of one lane's f64 chain
          if (rmclhip_status st = enqueue_find_with_moments(r, r->h_call->Tsm, maxd, maxd, r->fast_rho_cap, r->fast_tau_cap, cl.seq, true)) return st;
          HIPCHK(wait_moments(r, cl.seq, maxd, maxd, r->fast_rho_cap, r->fast_tau_cap));
          if (r->mset.valid) {
            xform T_s = xidentity();
            cstats last = cs_identity();
            fs.code = 0u; fs.n_uncertain = r->mset.n_unc;
            for (uint32_t it = 0; it < n_iter; ++it) {
              const float rho = micp_rho(T_s), tau = micp_tau(T_s);
              fs.max_rho = std::max(fs.max_rho, rho);
              fs.max_tau = std::max(fs.max_tau, tau);
              if (!(rho <= r->mset.rho_cap) || !(tau <= r->mset.tau_cap)) { fs.code = 1u; fs.iter = it; break; }
              last = micp_statistics_from_set(r->mset, T_s, maxd);
              T_s = xmul(T_s, umeyama(last));   // kernels.hip micp_advance_sensor
            }
            if (fs.code == 0u) {
              // kernels.hip micp_close_sensor
              fs.iter = n_iter;
              const xform Tso = xmul(Tbo, r->Tsb);
              r->h_state->T_snew_sold = T_s;
              r->h_state->T_onew_oold = xmul(xmul(Tso, T_s), xinv(Tso));
              r->h_state->stats_o = cs_merge(cs_identity(), cs_transform(Tbo, cs_transform(r->Tsb, last)));
            }
            fs_ready = true;
          } else {
            // more undecided correspondences than the host takes: the device loop on the rows the find left (a sequence number of its own:
            // the publish launch used this one for its hand-over flags and its tag)
            device_loop = true;
            cl.seq = r->h_call->seq = next_seq(r);
            // kind and row layout are the ones enqueue_find_with_moments actually ran (it replaces kind 24 by 23 to get the epilogue):
            // the loop folds THOSE rows instead of paying a moment pass of its own
            const int ufv = r->last_moment_find_kind;
            FindParams ufp;
            fill_find_params(r, ufp, 1, ufv);
            const uint32_t unb = find_moments_blocks(ufp, ufv), uwpb = (ufv == 2) ? 1u : 4u;
            if (r->last_moment_find_tiled)
              HIPCHK(launch_micp_fast_loop_tiled(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p,
                                                 nred, unb, r->d_fast_partials.p, r->d_fast_mask.p, r->W, ufp.tiles_x, ufp.tile_w_log2, uwpb, n_iter,
                                                 r->h_state_dev, r->h_fast_status_dev, r->h_done_dev, r->stream, cl, r->d_fold_rows, r->d_fold_flags));
            else
              HIPCHK(launch_micp_fast(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p, nred,
                                      nullptr, r->d_fast_partials.p, r->d_fast_mask.p, n_iter, r->h_state_dev, r->h_fast_status_dev,
                                      r->h_done_dev, r->stream, &cl));
          }
        } else if (tiled) {
          // (fast_mode 4, round 3's default) TWO kernels: the find forms the moments in its epilogue (find_kernel.hip.h: the 10 x 10 factor
          // products of its 64 correspondences per wave through f64 MFMA), one partial row per workgroup; the loop launch folds them and runs
          // every iteration on the device
          HIPCHK(r->d_fast_partials.reserve(static_cast<size_t>(nb) * kMicpFastMoments));
          HIPCHK(r->d_fast_mask.reserve(static_cast<size_t>(nb) * wpb));
          fp.mom_dataset_points = r->ds_pts;
          fp.mom_dataset_mask = r->ds_has_mask ? r->ds_msk : nullptr;
          fp.mom_n = nred;
          fp.mom_gate_lo = maxd; fp.mom_gate_hi = maxd; fp.mom_rho_cap = r->fast_rho_cap; fp.mom_tau_cap = r->fast_tau_cap;
          fp.mom_partials = r->d_fast_partials.p;
          fp.mom_unc_mask = r->d_fast_mask.p;
          if (!r->d_fold_rows) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&r->d_fold_rows), kMicpFoldBlocks * kMicpFastMoments * sizeof(double) + kMicpFoldBlocks * sizeof(uint32_t)));
            HIPCHK(hipMemset(r->d_fold_rows, 0, kMicpFoldBlocks * kMicpFastMoments * sizeof(double) + kMicpFoldBlocks * sizeof(uint32_t)));
            HIPCHK(hipDeviceSynchronize());
            r->d_fold_flags = reinterpret_cast<uint32_t*>(r->d_fold_rows + kMicpFoldBlocks * kMicpFastMoments);
          }
          r->last_fast_rows = nb; r->last_fast_words = wpb * nb;
          HIPCHK(launch_find_moments(fp, r->kind, fv, r->stream));
          HIPCHK(launch_micp_fast_loop_tiled(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p,
                                             nred, nb, r->d_fast_partials.p, r->d_fast_mask.p, r->W, fp.tiles_x, fp.tile_w_log2, wpb, n_iter,
                                             r->h_state_dev, r->h_fast_status_dev, r->h_done_dev, r->stream, cl, r->d_fold_rows, r->d_fold_flags));
        } else {
          r->last_fast_rows = micp_fast_blocks(nred); r->last_fast_words = (nred + 63u) / 64u;
          HIPCHK(launch_find(fp, r->kind, fv, r->stream));
          HIPCHK(launch_micp_fast(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p, nred,
                                  nullptr, r->d_fast_partials.p, r->d_fast_mask.p, n_iter, r->h_state_dev, r->h_fast_status_dev,
                                  r->h_done_dev, r->stream, &cl));
        }
        (void)device_loop;
      } else if (!r->micp_fast_exec || r->fast_graph_dirty || !(key == r->micp_fast_key)) {
        // the previous call returned on its completion tag, which precedes the stream's own completion: let the last node
        // retire before its executable graph is destroyed
        HIPCHK(hipStreamSynchronize(r->stream));
        if (r->micp_fast_exec) { (void)hipGraphExecDestroy(r->micp_fast_exec); r->micp_fast_exec = nullptr; }
        if (r->micp_fast_graph) { (void)hipGraphDestroy(r->micp_fast_graph); r->micp_fast_graph = nullptr; }
        HIPCHK(hipStreamBeginCapture(r->stream, hipStreamCaptureModeThreadLocal));
        r->capturing = true;
        hipError_t le = hipMemcpyAsync(r->d_call, r->h_call, sizeof(MicpCall), hipMemcpyHostToDevice, r->stream);
        if (le == hipSuccess) {
          FindParams fp;
          fill_find_params(r, fp, 1);
          fp.Tsm_arr = &r->d_call->Tsm;
          fp.Tms_arr = &r->d_call->Tms;
          le = launch_find(fp, r->kind, find_variant(r, 1), r->stream);
        }
        if (le == hipSuccess)
          le = launch_micp_fast(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p, nred,
                                r->d_call, r->d_fast_partials.p, r->d_fast_mask.p, n_iter, r->h_state_dev, r->h_fast_status_dev,
                                r->h_done_dev, r->stream);
        r->capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t ce = hipStreamEndCapture(r->stream, &g);
        if (le != hipSuccess) { if (g) (void)hipGraphDestroy(g); return fail(RMCLHIP_ERR_HIP, std::string("micp fast capture: ") + hipGetErrorString(le)); }
        if (ce != hipSuccess) return fail(RMCLHIP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
        r->micp_fast_graph = g;
        HIPCHK(hipGraphInstantiate(&r->micp_fast_exec, g, nullptr, nullptr, 0));
        r->micp_fast_key = key;
        r->fast_graph_dirty = false;
      }
      if (r->use_graph && r->fast_mode == 2) HIPCHK(hipGraphLaunch(r->micp_fast_exec, r->stream));
      if (!fs_ready) {
        // sum of the tag: the status block, plus the state block when the loop ran to its end (code 0)
        DoneCheck chk; chk.base = r->h_fast_status; chk.base_bytes = sizeof(MicpFastStatus); chk.code = &r->h_fast_status->code;
        chk.extra[0] = r->h_state; chk.extra_bytes[0] = sizeof(MicpState);
        HIPCHK(wait_done(r->ctx, r->h_done, r->h_call->seq, chk, r->stream));
        fs = *r->h_fast_status;
      } else {
        ++r->fast_info.host_loops;
      }
      r->last_fast = fs;
      r->fast_info.attempts++;
      r->fast_info.last_code = fs.code;
      r->fast_info.last_uncertain = fs.n_uncertain;
      r->fast_info.last_rho = fs.max_rho;
      r->fast_info.last_tau = fs.max_tau;
      r->fast_info.last_setup_clocks = fs.code == 0u ? fs.pad[0] : 0u;
      r->fast_info.last_loop_clocks = fs.code == 0u ? fs.pad[1] : 0u;
      if (fs.code == 0u) {
        r->fast_info.done++;
        r->fast_overflows = 0;
        r->fast_rho_cap = std::max(0.002f, std::max(2.0f * fs.max_rho, 0.9f * r->fast_rho_cap));
        r->fast_tau_cap = std::max(0.005f, std::max(2.0f * fs.max_tau, 0.9f * r->fast_tau_cap));
        r->fast_info.rho_cap = r->fast_rho_cap;
        r->fast_info.tau_cap = r->fast_tau_cap;
        from_x(r->h_state->T_onew_oold, T_out);
        if (stats_out) from_cs(r->h_state->stats_o, stats_out);
        return RMCLHIP_OK;
      }
      if (fs.code != 1u && fs.code != 2u) return fail(RMCLHIP_ERR_HIP, "correct_once: the moment-form loop did not report a status");
      if (fs.code == 2u) r->fast_info.overflows++; else r->fast_info.cap_exits++;
    }
    // the one-launch-per-iteration chain ends with k_micp_close, which publishes a completion tag the host polls (a fresh
    // sequence number: the moment-form attempt above may have published one for this call already)
    const bool polled = r->loop_blocks == 0 && !r->fused_tail && n_iter > 0;
    r->h_call->seq = next_seq(r);
    auto enqueue_chain = [&]() -> rmclhip_status {
      HIPCHK(hipMemcpyAsync(r->d_call, r->h_call, sizeof(MicpCall), hipMemcpyHostToDevice, r->stream));
      FindParams p;
      fill_find_params(r, p, 1);
      p.Tsm_arr = &r->d_call->Tsm;
      p.Tms_arr = &r->d_call->Tms;
      const int fvariant = find_variant(r, p.nposes);
      HIPCHK(launch_find(p, r->kind, fvariant, r->stream));
      const bool iter_form = r->loop_blocks == 0 && !r->fused_tail && n_iter > 0;
      if (!iter_form) HIPCHK(launch_micp_init(r->d_state, r->d_loop_barrier, r->stream));  // k_micp_iter initialises itself
      MicpState* final_state = r->d_state;
      const uint8_t* dmask = r->ds_has_mask ? r->ds_msk : nullptr;
      if (r->loop_blocks > 0) {
        // persistent loop: every iteration inside ONE launch (k_micp_loop); A/B only -- a device-wide barrier
        // across the 8 XCDs costs more than the launch boundaries it replaces
        HIPCHK(launch_micp_loop(r->ds_pts, dmask, r->d_points.p, r->d_normals.p, r->d_hits.p, nred, n_iter,
                                r->d_call, r->d_partials.p, r->d_loop_barrier, r->d_state,
                                static_cast<uint32_t>(r->loop_blocks & 0xFFFF), (r->loop_blocks >> 16) != 0, r->stream));
      } else if (iter_form) {
        // default: ONE launch per iteration (k_micp_iter solves the previous iteration in its prologue) + one
        // closing solve: n_iter + 1 launches instead of 2 * n_iter
        const uint32_t nb = reduce_num_blocks(nred, 1);
        double* part[2] = {r->d_partials.p, r->d_partials.p + static_cast<size_t>(nb) * 16};
        for (uint32_t i = 0; i < n_iter; ++i)
          HIPCHK(launch_micp_iter(r->ds_pts, dmask, r->d_points.p, r->d_normals.p, r->d_hits.p, nred, nb, r->d_call,
                                  part[(i + 1u) & 1u], part[i & 1u], r->d_state + (i & 1u), r->d_state + ((i + 1u) & 1u),
                                  i == 0, r->stream));
        // the closing step writes the result straight into host-mapped memory (no copy node)
        HIPCHK(launch_micp_close(part[(n_iter - 1u) & 1u], nb, r->d_call, r->d_state + (n_iter & 1u), r->h_state_dev,
                                 r->h_done_dev, r->stream));
        final_state = nullptr;
      } else
      for (uint32_t i = 0; i < n_iter; ++i) {
        ReduceTail tail;
        tail.mode = kTailMicp;
        tail.Tbo = Tbo;
        tail.state = r->d_state;
        tail.call = r->d_call;
        if (rmclhip_status st = reduce_enqueue(r, xidentity(), &r->d_state->T_snew_sold, maxd, 1, tail)) return st;
      }
      if (final_state) HIPCHK(hipMemcpyAsync(r->h_state, final_state, sizeof(MicpState), hipMemcpyDeviceToHost, r->stream));
      return RMCLHIP_OK;
    };
    if (r->use_graph) {
      rmclhip_rcc::MicpKey key;
      std::memset(&key, 0, sizeof(key));
      key.n_iter = n_iter; key.W = r->W; key.H = r->H; key.n_dataset = r->n_dataset;
      key.kind = static_cast<int>(r->kind); key.variant = r->variant; key.tile = r->tile_override;
      key.fused = (r->fused_tail ? 1 : 0) | (r->loop_blocks << 1); key.has_mask = r->ds_has_mask ? 1 : 0;
      key.ptrs[0] = r->d_points.p; key.ptrs[1] = r->ds_pts; key.ptrs[2] = r->d_partials.p;
      key.ptrs[3] = r->d_model_tab.p; key.ptrs[4] = r->ds_msk; key.ptrs[5] = r->d_hits.p;
      if (!r->micp_exec || r->graph_dirty || !(key == r->micp_key)) {
        HIPCHK(hipStreamSynchronize(r->stream));   // see the moment-form graph above
        if (r->micp_exec) { (void)hipGraphExecDestroy(r->micp_exec); r->micp_exec = nullptr; }
        if (r->micp_graph) { (void)hipGraphDestroy(r->micp_graph); r->micp_graph = nullptr; }
        HIPCHK(hipStreamBeginCapture(r->stream, hipStreamCaptureModeThreadLocal));
        r->capturing = true;
        const rmclhip_status cst = enqueue_chain();
        r->capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t ce = hipStreamEndCapture(r->stream, &g);
        if (cst != RMCLHIP_OK) { if (g) (void)hipGraphDestroy(g); return cst; }
        if (ce != hipSuccess) return fail(RMCLHIP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
        r->micp_graph = g;
        HIPCHK(hipGraphInstantiate(&r->micp_exec, g, nullptr, nullptr, 0));
        r->micp_key = key;
        r->graph_dirty = false;
      }
      HIPCHK(hipGraphLaunch(r->micp_exec, r->stream));
    } else {
      if (rmclhip_status st = enqueue_chain()) return st;
    }
    if (polled) {
      DoneCheck chk; chk.base = r->h_state; chk.base_bytes = sizeof(MicpState);
      HIPCHK(wait_done(r->ctx, r->h_done, r->h_call->seq, chk, r->stream));
    }
    else HIPCHK(hipStreamSynchronize(r->stream));
    if (fast_tried) {
      // the pre-transform this correction ended with bounds the next attempt (iterates approach it monotonically in the
      // usual case; an attempt that still leaves the caps costs one more fallback and doubles them)
      const xform Tb = xmul(xmul(xinv(Tbo), r->h_state->T_onew_oold), Tbo);
      const xform Ts = xmul(xmul(xinv(r->Tsb), Tb), r->Tsb);
      const float rho = 2.0f * std::sqrt(Ts.R.x * Ts.R.x + Ts.R.y * Ts.R.y + Ts.R.z * Ts.R.z);
      const float tau = std::sqrt(Ts.t.x * Ts.t.x + Ts.t.y * Ts.t.y + Ts.t.z * Ts.t.z);
      if (r->last_fast.code == 2u) {
        // too many uncertain correspondences: tighter caps, and stop trying when that does not help either
        r->fast_rho_cap = std::max(0.002f, 1.25f * rho);
        r->fast_tau_cap = std::max(0.005f, 1.25f * tau);
        if (++r->fast_overflows >= 2u) { r->fast_holdoff = 32u; r->fast_overflows = 0; }
      } else {
        r->fast_rho_cap = std::max(0.002f, std::max(2.0f * rho, 2.0f * r->last_fast.max_rho));
        r->fast_tau_cap = std::max(0.005f, std::max(2.0f * tau, 2.0f * r->last_fast.max_tau));
      }
      r->fast_info.rho_cap = r->fast_rho_cap;
      r->fast_info.tau_cap = r->fast_tau_cap;
    }
    from_x(r->h_state->T_onew_oold, T_out);
    if (stats_out) from_cs(r->h_state->stats_o, stats_out);
    return RMCLHIP_OK;
  }
  // schedule (B), lidar_corrector_embree_benchmark.cpp:127-135: re-raycast from the corrected pose every iteration
  xform T_onew_oold = xidentity();
  cstats last = cs_identity();
  for (uint32_t i = 0; i < n_iter; ++i) {
    const xform Tom_cur = xmul(Tom, T_onew_oold);
    if (rmclhip_status st = find_enqueue(r, xmul(Tom_cur, Tbo))) return st;
    ReduceTail tail;
    tail.mode = kTailStats;
    tail.stats_out = r->h_stats_dev;
    const bool polled = !r->fused_tail;
    if (polled) { tail.done = r->h_done_dev; tail.seq = next_seq(r); }
    if (rmclhip_status st = reduce_enqueue(r, xidentity(), nullptr, maxd, 1, tail)) return st;
    if (polled) {
      DoneCheck chk; chk.base = &r->h_stats[0]; chk.base_bytes = sizeof(cstats);
      HIPCHK(wait_done(r->ctx, r->h_done, tail.seq, chk, r->stream));
    }
    else HIPCHK(hipStreamSynchronize(r->stream));
    const cstats Cs_o = cs_transform(Tbo, cs_transform(r->Tsb, r->h_stats[0]));
    last = cs_merge(cs_identity(), Cs_o);
    T_onew_oold = xmul(T_onew_oold, umeyama(last));
  }
  from_x(T_onew_oold, T_out);
  if (stats_out) from_cs(last, stats_out);
  return RMCLHIP_OK;
}


// MICPLocalizationNode::correctOnce inner loop for N sensors on one device (micp_localization.cpp:900-964): one find per
// sensor, then per iteration one reduction per sensor and ONE step launch that merges, solves and hands out the next
// pre-transforms; nothing returns to the host until the loop is over (the host form costs one synchronisation per sensor
// and iteration).
rmclhip_status rmclhip_micp_correct_once(rmclhip_rcc* const* sensors, uint32_t n_sensors, const rmclhip_transform* Tom_,
                                         const rmclhip_transform* Tbo_, const double* merge_weight_multiplier, uint32_t n_iter,
                                         double convergence_progress, rmclhip_transform* T_out,
                                         rmclhip_cross_statistics* merged_out) {
  ApiGuard guard_("rmclhip_micp_correct_once");
  if (!sensors || !Tom_ || !Tbo_ || !T_out || n_sensors == 0) return fail(RMCLHIP_ERR_INVALID, "micp_correct_once: null");
  if (n_sensors > kMaxMicpSensors) return fail(RMCLHIP_ERR_UNSUPPORTED, "micp_correct_once: at most 8 sensors");
  rmclhip_rcc* r0 = sensors[0];
  for (uint32_t s = 0; s < n_sensors; ++s) {
    rmclhip_rcc* r = sensors[s];
    if (!r) return fail(RMCLHIP_ERR_INVALID, "micp_correct_once: null sensor");
    if (r->ctx->device != r0->ctx->device) return fail(RMCLHIP_ERR_INVALID, "micp_correct_once: sensors live on different devices");
    if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "micp_correct_once: sensor without a model");
    if (r->n_dataset == 0) return fail(RMCLHIP_ERR_INVALID, "micp_correct_once: sensor without a dataset");
    if (!micp_outputs_selected(r)) return fail(RMCLHIP_ERR_INVALID, kNeedMicpOutputs);
  }
  HIPCHK(hipSetDevice(r0->ctx->device));
  hipStream_t st = r0->stream;
  static thread_local MicpMultiCall h_call;
  std::memset(&h_call, 0, sizeof(h_call));
  const xform Tom = to_x(Tom_);
  for (uint32_t s = 0; s < n_sensors; ++s) {
    rmclhip_rcc* r = sensors[s];
    // (no synchronisation with the sensor's own stream: its find is enqueued ON that stream, behind whatever it still holds, and the
    // loop touches the sensor only behind that find -- flag or event; a hipStreamSynchronize per sensor cost ~9 us each here)
    const size_t n = static_cast<size_t>(r->W) * r->H;
    r->n_model = static_cast<uint32_t>(n);
    r->nposes_last = 1;
    if (rmclhip_status e = ensure_model_buffers(r, n)) return e;
    const uint32_t nred = (r->n_dataset < r->n_model) ? r->n_dataset : r->n_model;
    const uint32_t nb = reduce_num_blocks(nred, 1);
    HIPCHK(r->d_partials.reserve(std::max<size_t>(static_cast<size_t>(nb) * 32, 2u * 256u * 16u)));
    h_call.Tsb[s] = r->Tsb;
    h_call.Tbo[s] = to_x(Tbo_ + s);
    h_call.weight[s] = merge_weight_multiplier ? merge_weight_multiplier[s] : 1.0;
    h_call.partials[s] = r->d_partials.p;
    h_call.nblocks[s] = nb;
  }
  h_call.n_sensors = n_sensors;
  h_call.seq = next_seq(r0);
  // call + state live with the first sensor and persist between calls (an allocation per call cost more than the loop)
  HIPCHK(r0->d_multi_blob.reserve(sizeof(MicpMultiCall) + sizeof(MicpMultiState)));
  if (!r0->h_multi_state) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&r0->h_multi_state), sizeof(MicpMultiState), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r0->h_multi_state_dev), r0->h_multi_state, 0));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&r0->h_multi_status), sizeof(MicpMultiFastStatus), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r0->h_multi_status_dev), r0->h_multi_status, 0));
  }
  MicpMultiCall* d_call = reinterpret_cast<MicpMultiCall*>(r0->d_multi_blob.p);
  MicpMultiState* d_state = reinterpret_cast<MicpMultiState*>(r0->d_multi_blob.p + sizeof(MicpMultiCall));
  hipError_t e = hipSuccess;
  // ---- moment form first (kernels.hip k_micp_multi_fast_loop): every sensor's caps are the ones its own corrections learnt
  bool fast_eligible = n_iter >= 2u;
  for (uint32_t s = 0; s < n_sensors; ++s) fast_eligible = fast_eligible && sensors[s]->fast_mode != 0;
  bool fast_tried = false;
  if (fast_eligible && r0->multi_holdoff > 0u) --r0->multi_holdoff;
  else if (fast_eligible) fast_tried = true;
  // ---- round 4: every sensor's find forms its moments and publishes them to the host (its own stream, its own block and tag), the
  // host merges and solves (micp_host.h; same frame-by-frame order as k_micp_multi_step).  Any sensor with too many undecided
  // correspondences or a pre-transform outside its caps: the device forms below, from scratch.
  bool device_fast = fast_tried;   // the device's moment loop is tried (after the host form, when that is on and did not finish)
  bool host_form = fast_tried;
  for (uint32_t s = 0; s < n_sensors; ++s) host_form = host_form && sensors[s]->fast_mode == 1 && !sensors[s]->fused_tail;
  if (host_form) {
    float maxd[kMaxMicpSensors];
    uint32_t seqs[kMaxMicpSensors];
    for (uint32_t s = 0; s < n_sensors; ++s) {
      rmclhip_rcc* r = sensors[s];
      maxd[s] = adaptive_max_dist(r, convergence_progress);
      seqs[s] = (s == 0u) ? h_call.seq : next_seq(r);
      if (rmclhip_status e2 = enqueue_find_with_moments(r, xmul(xmul(Tom, h_call.Tbo[s]), r->Tsb), maxd[s], maxd[s], r->fast_rho_cap, r->fast_tau_cap,
                                                        seqs[s], true))
        return e2;
    }
    bool all_valid = true;
    for (uint32_t s = 0; s < n_sensors; ++s) {
      rmclhip_rcc* r = sensors[s];
      HIPCHK(wait_moments(r, seqs[s], maxd[s], maxd[s], r->fast_rho_cap, r->fast_tau_cap));
      all_valid = all_valid && r->mset.valid;
    }
    MicpMultiFastStatus hs;
    std::memset(&hs, 0, sizeof(hs));
    hs.code = all_valid ? 0u : 2u;
    xform T_onew_oold = xidentity(), T_s[kMaxMicpSensors];
    cstats merged = cs_identity(), merged_w = cs_identity();
    for (uint32_t s = 0; s < n_sensors; ++s) { T_s[s] = xidentity(); hs.n_uncertain += sensors[s]->mset.n_unc; }
    for (uint32_t it = 0; it < n_iter && hs.code == 0u; ++it) {
      merged = cs_identity(); merged_w = cs_identity();
      for (uint32_t s = 0; s < n_sensors; ++s) {
        const rmclhip_rcc* r = sensors[s];
        const float rho = micp_rho(T_s[s]), tau = micp_tau(T_s[s]);
        hs.max_rho[s] = std::max(hs.max_rho[s], rho);
        hs.max_tau[s] = std::max(hs.max_tau[s], tau);
        if (!(rho <= r->mset.rho_cap) || !(tau <= r->mset.tau_cap)) { hs.code = 1u; hs.iter = it; hs.sensor = s; break; }
        // micp_localization.cpp:926-937 with MICPSensor.hpp:178-182
        const cstats stats_s = micp_statistics_from_set(r->mset, T_s[s], maxd[s]);
        const cstats Cs_o = cs_transform(h_call.Tbo[s], cs_transform(r->Tsb, stats_s));
        cstats Cs_w = Cs_o;
        Cs_w.n_meas = static_cast<uint32_t>(static_cast<double>(Cs_w.n_meas) * h_call.weight[s]);
        merged = cs_merge(merged, Cs_o);
        merged_w = cs_merge(merged_w, Cs_w);
      }
      if (hs.code != 0u) break;
      T_onew_oold = xmul(T_onew_oold, umeyama(merged_w));   // :952-963
      for (uint32_t s = 0; s < n_sensors; ++s) {
        const xform T_bnew_bold = xmul(xmul(xinv(h_call.Tbo[s]), T_onew_oold), h_call.Tbo[s]);
        T_s[s] = xmul(xmul(xinv(sensors[s]->Tsb), T_bnew_bold), sensors[s]->Tsb);
      }
    }
    for (uint32_t s = 0; s < n_sensors && hs.code != 2u; ++s) {   // (code 2: the device loop below is this call's attempt)
      rmclhip_rcc* r = sensors[s];
      r->fast_info.attempts++;
      r->fast_info.last_code = hs.code;
      r->fast_info.last_uncertain = hs.n_uncertain;
      r->fast_info.last_rho = hs.max_rho[s];
      r->fast_info.last_tau = hs.max_tau[s];
      r->fast_info.last_setup_clocks = r->fast_info.last_loop_clocks = 0u;
    }
    if (hs.code == 0u) {
      r0->multi_overflows = 0;
      for (uint32_t s = 0; s < n_sensors; ++s) {
        rmclhip_rcc* r = sensors[s];
        r->fast_info.done++;
        r->fast_info.host_loops++;
        learn_caps(r, hs.max_rho[s], hs.max_tau[s]);
      }
      from_x(T_onew_oold, T_out);
      if (merged_out) from_cs(merged, merged_out);
      return RMCLHIP_OK;
    }
    // not served on the host: the device forms take over, from scratch (the moment sets belong to finds that are about to be redone).
    // A pre-transform that left its caps would leave them in the device's moment loop as well: straight to the per-iteration form,
    // whose end learns the caps from this status; too many undecided correspondences for the host (> kMicpHostMaxUnc = 1024 in a sensor): the device's
    // moment loop takes up to 4096.
    if (hs.code == 1u) {
      device_fast = false;
      *r0->h_multi_status = hs;
      for (uint32_t s = 0; s < n_sensors; ++s) sensors[s]->fast_info.cap_exits++;
    }
    for (uint32_t s = 0; s < n_sensors; ++s) drop_moment_set(sensors[s]);
    h_call.seq = next_seq(r0);
  }
  // sensor->setTom(Tom); sensor->findCorrespondences()  (:900-909): Tbm = Tom * Tbo.  The sensors' finds (and moment passes) do not
  // depend on each other: sensor 0's go to the stream the loop runs on, every other sensor's to ITS OWN stream, joined by an event
  // before the loop -- one scan leaves the chip partly idle (bench.py extras.find_two_operators_in_flight_*), a second sensor's scan
  // fills it (round 3: everything sat on one stream)
  MicpMultiFastParams fp;
  std::memset(&fp, 0, sizeof(fp));
  // A cross-stream EVENT takes ~10 us to reach the waiting queue (measured: the loop started 10-12 us after its last input), so in the
  // moment form the join is a flag: a one-lane kernel behind the sensor's moment pass stores the call's sequence number, the loop
  // kernel -- launched without waiting -- polls it before it touches that sensor's rows.  The per-iteration form (fallback) records
  // an event on every other sensor's stream and waits for it.
  if (n_sensors > 1u && !r0->d_join_flags) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&r0->d_join_flags), kMaxMicpSensors * sizeof(uint32_t)));
    HIPCHK(hipMemset(r0->d_join_flags, 0, kMaxMicpSensors * sizeof(uint32_t)));
    HIPCHK(hipDeviceSynchronize());
  }
  // every find first (the host needs ~4 us per launch: the second sensor's scan should not wait behind the first sensor's moment pass
  // being enqueued), then the moment passes and the flags
  for (uint32_t s = 0; s < n_sensors && e == hipSuccess; ++s) {
    rmclhip_rcc* r = sensors[s];
    FindParams p;
    fill_find_params(r, p, 1);
    p.Tsm = xmul(xmul(Tom, h_call.Tbo[s]), r->Tsb);
    p.Tms = xinv(p.Tsm);
    e = launch_find(p, r->kind, find_variant(r, 1), (s == 0u) ? st : r->stream);
  }
  for (uint32_t s = 0; s < n_sensors && e == hipSuccess && device_fast; ++s) {
    rmclhip_rcc* r = sensors[s];
    hipStream_t fs = (s == 0u) ? st : r->stream;
    const uint32_t nred = (r->n_dataset < r->n_model) ? r->n_dataset : r->n_model;
    HIPCHK(r->d_fast_partials.reserve(static_cast<size_t>(micp_fast_blocks(nred)) * kMicpFastMoments));
    HIPCHK(r->d_fast_mask.reserve((static_cast<size_t>(nred) + 63u) / 64u));
    // per-call data by value: no H2D copy node per sensor (4.4 us each in the kernel trace of round 2's chain)
    MicpCallLite cl;
    cl.Tsb = r->Tsb; cl.Tbo = h_call.Tbo[s]; cl.max_dist = adaptive_max_dist(r, convergence_progress);
    cl.gate_lo = cl.max_dist; cl.gate_hi = cl.max_dist;
    cl.rho_cap = r->fast_rho_cap; cl.tau_cap = r->fast_tau_cap; cl.seq = h_call.seq;
    HIPCHK(launch_micp_moments(r->ds_pts, r->ds_has_mask ? r->ds_msk : nullptr, r->d_points.p, r->d_normals.p, r->d_hits.p, nred,
                               nullptr, r->d_fast_partials.p, r->d_fast_mask.p, fs, &cl));
    fp.dataset_points[s] = r->ds_pts; fp.model_points[s] = r->d_points.p; fp.model_normals[s] = r->d_normals.p;
    fp.partials[s] = r->d_fast_partials.p; fp.unc_mask[s] = r->d_fast_mask.p;
    fp.n[s] = nred; fp.nblocks[s] = micp_fast_blocks(nred);
    fp.Tsb[s] = cl.Tsb; fp.Tbo[s] = cl.Tbo; fp.weight[s] = h_call.weight[s];
    fp.max_dist[s] = cl.max_dist; fp.rho_cap[s] = cl.rho_cap; fp.tau_cap[s] = cl.tau_cap;
    if (s != 0u) {
      HIPCHK(launch_signal_flag(r0->d_join_flags + s, h_call.seq, fs));
      fp.join_mask |= 1u << s;
    }
  }
  fp.join_flags = r0->d_join_flags;
  if (e != hipSuccess) return fail(RMCLHIP_ERR_HIP, std::string("micp_correct_once: ") + hipGetErrorString(e));
  if (device_fast) {
    fp.n_sensors = n_sensors;
    fp.seq = h_call.seq;
    fp.n_iter = n_iter;
    fp.state_out = r0->h_multi_state_dev;
    fp.status = r0->h_multi_status_dev;
    fp.done = r0->h_done_dev + 1;
    HIPCHK(launch_micp_multi_fast_loop(fp, st));
    {
      DoneCheck chk; chk.base = r0->h_multi_status; chk.base_bytes = sizeof(MicpMultiFastStatus); chk.code = &r0->h_multi_status->code;
      chk.extra[0] = &r0->h_multi_state->T_onew_oold; chk.extra_bytes[0] = sizeof(xform);
      chk.extra[1] = &r0->h_multi_state->merged_o; chk.extra_bytes[1] = sizeof(cstats);
      chk.extra[2] = &r0->h_multi_state->merged_weighted_o; chk.extra_bytes[2] = sizeof(cstats);
      HIPCHK(wait_done(r0->ctx, r0->h_done + 1, h_call.seq, chk, st));
    }
    const MicpMultiFastStatus fs = *r0->h_multi_status;
    for (uint32_t s = 0; s < n_sensors; ++s) {
      rmclhip_rcc* r = sensors[s];
      r->fast_info.attempts++;
      r->fast_info.last_code = fs.code;
      r->fast_info.last_uncertain = fs.n_uncertain;
      r->fast_info.last_rho = fs.max_rho[s];
      r->fast_info.last_tau = fs.max_tau[s];
    }
    if (fs.code == 0u) {
      r0->multi_overflows = 0;
      for (uint32_t s = 0; s < n_sensors; ++s) {
        rmclhip_rcc* r = sensors[s];
        r->fast_info.done++;
        r->fast_rho_cap = std::max(0.002f, std::max(2.0f * fs.max_rho[s], 0.9f * r->fast_rho_cap));
        r->fast_tau_cap = std::max(0.005f, std::max(2.0f * fs.max_tau[s], 0.9f * r->fast_tau_cap));
      }
      from_x(r0->h_multi_state->T_onew_oold, T_out);
      if (merged_out) from_cs(r0->h_multi_state->merged_o, merged_out);
      return RMCLHIP_OK;
    }
    if (fs.code != 1u && fs.code != 2u) return fail(RMCLHIP_ERR_HIP, "micp_correct_once: the moment-form loop did not report a status");
    for (uint32_t s = 0; s < n_sensors; ++s) {
      if (fs.code == 2u) sensors[s]->fast_info.overflows++; else sensors[s]->fast_info.cap_exits++;
    }
  }
  // per-iteration form (fallback, or the moment form is off): the call block goes to the device.  The other sensors' finds ran on
  // their own streams: this stream waits for their events first.
  for (uint32_t s = 1; s < n_sensors; ++s) {
    rmclhip_rcc* r = sensors[s];
    if (!r->ev_join) HIPCHK(hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming));
    HIPCHK(hipEventRecord(r->ev_join, r->stream));   // (recorded here, not per call: the moment form never needs it)
    HIPCHK(hipStreamWaitEvent(st, r->ev_join, 0));
  }
  e = hipMemcpyAsync(d_call, &h_call, sizeof(h_call), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = launch_micp_multi_init(d_call, d_state, st);
  for (uint32_t it = 0; it < n_iter && e == hipSuccess; ++it) {
    for (uint32_t s = 0; s < n_sensors && e == hipSuccess; ++s) {
      rmclhip_rcc* r = sensors[s];
      const uint32_t nred = (r->n_dataset < r->n_model) ? r->n_dataset : r->n_model;
      ReduceParams rp;
      std::memset(&rp, 0, sizeof(rp));
      rp.dataset_points = r->ds_pts;
      rp.dataset_mask = r->ds_has_mask ? r->ds_msk : nullptr;
      rp.model_points = r->d_points.p; rp.model_normals = r->d_normals.p; rp.model_mask = r->d_hits.p;
      rp.n = nred; rp.nposes = 1;
      rp.max_dist = adaptive_max_dist(r, convergence_progress);
      rp.Tpre = xidentity();
      rp.Tpre_dev = &d_state->T_snew_sold[s];
      rp.partials = r->d_partials.p;
      rp.nblocks = h_call.nblocks[s];
      rp.tail_mode = kTailNone;
      e = launch_reduce_partials(rp, st);
    }
    if (e == hipSuccess) e = launch_micp_multi_step(d_call, d_state, st);
  }
  MicpMultiState h_state;
  if (e == hipSuccess) e = hipMemcpyAsync(&h_state, d_state, sizeof(h_state), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(RMCLHIP_ERR_HIP, std::string("micp_correct_once: ") + hipGetErrorString(e));
  if (fast_tried) {
    // caps for the next attempt from the pre-transforms this correction ended with (see rmclhip_rcc_correct_once)
    const bool overflow = r0->h_multi_status->code == 2u;
    for (uint32_t s = 0; s < n_sensors; ++s) {
      rmclhip_rcc* r = sensors[s];
      const xform Ts = h_state.T_snew_sold[s];
      const float rho = 2.0f * std::sqrt(Ts.R.x * Ts.R.x + Ts.R.y * Ts.R.y + Ts.R.z * Ts.R.z);
      const float tau = std::sqrt(Ts.t.x * Ts.t.x + Ts.t.y * Ts.t.y + Ts.t.z * Ts.t.z);
      if (overflow) {
        r->fast_rho_cap = std::max(0.002f, 1.25f * rho);
        r->fast_tau_cap = std::max(0.005f, 1.25f * tau);
      } else {
        r->fast_rho_cap = std::max(0.002f, std::max(2.0f * rho, 2.0f * r0->h_multi_status->max_rho[s]));
        r->fast_tau_cap = std::max(0.005f, std::max(2.0f * tau, 2.0f * r0->h_multi_status->max_tau[s]));
      }
    }
    if (overflow && ++r0->multi_overflows >= 2u) { r0->multi_holdoff = 32u; r0->multi_overflows = 0; }
  }
  from_x(h_state.T_onew_oold, T_out);
  if (merged_out) from_cs(h_state.merged_o, merged_out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_correct_batch(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes,
                                         rmclhip_transform* Tdelta_out, rmclhip_cross_statistics* stats_out) {
  ApiGuard guard_("rmclhip_rcc_correct_batch");
  if (!r || !Tbm || !Tdelta_out) return fail(RMCLHIP_ERR_INVALID, "correct_batch: null");
  if (nposes == 0) return RMCLHIP_OK;
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "correct_batch: no sensor model");
  if (nposes > 32768) return fail(RMCLHIP_ERR_UNSUPPORTED, "correct_batch: at most 32768 poses per call");
  if (!micp_outputs_selected(r)) return fail(RMCLHIP_ERR_INVALID, kNeedMicpOutputs);
  HIPCHK(hipSetDevice(r->ctx->device));
  const size_t n = static_cast<size_t>(r->W) * r->H;
  if (r->n_dataset != n) return fail(RMCLHIP_ERR_INVALID, "correct_batch: dataset size != model size");
  if (nposes > r->h_batch_cap) {
    HIPCHK(hipStreamSynchronize(r->stream));
    if (r->h_bT) (void)hipHostFree(r->h_bT);
    if (r->h_bS) (void)hipHostFree(r->h_bS);
    r->h_bT = nullptr; r->h_bS = nullptr; r->h_batch_cap = 0;
    const uint32_t cap = std::max(nposes, 64u);
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&r->h_bT), sizeof(xform) * cap, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r->h_bT_dev), r->h_bT, 0));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&r->h_bS), sizeof(cstats) * cap, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r->h_bS_dev), r->h_bS, 0));
    r->h_batch_cap = cap;
  }
  if (rmclhip_status st = find_batch_enqueue(r, Tbm, nposes)) return st;
  ReduceTail tail;
  tail.mode = kTailBatchSolve;
  tail.Tdelta_out = r->h_bT_dev;
  tail.stats_out = r->h_bS_dev;
  if (rmclhip_status st = reduce_enqueue(r, xidentity(), nullptr, r->max_dist, nposes, tail)) return st;
  HIPCHK(wait_chain_end(r));
  std::memcpy(Tdelta_out, r->h_bT, sizeof(xform) * nposes);
  if (stats_out) std::memcpy(stats_out, r->h_bS, sizeof(cstats) * nposes);
  return RMCLHIP_OK;
}

