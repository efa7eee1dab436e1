// capi_internal.h -- what the translation units of the C ABI (include/rmclhip.h) share: the handle structs, the error plumbing and the
// internal functions one file defines and another calls.  Round 5 split the former capi.cpp (4 300 lines) by handle type:
//   capi_map.cpp       context, map / scene upload, host-side algebra, device memory helpers
//   capi_rcc.cpp       the correspondence operator: models, datasets, find, reduction, the MICP corrections (one and N sensors), pose batches
//   capi_rcc_tune.cpp  tuning knobs, autotune, measurement aids (include/rmclhip_bench.h) and diagnostics (include/rmclhip_lab.h) of that operator
//   capi_pf.cpp        particle-filter sensor update, motion update, resamplers
//   capi_multi.cpp     several devices in one process: sharded pose batches, communicators (RCCL / loopback), the sharded filter
// Host-side orchestration only: device memory, streams, launches.  There is no CPU compute path: without a HIP device every compute
// entry point fails with RMCLHIP_ERR_NO_DEVICE.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library is resolved with dlopen when the first communicator is created
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "../../include/rmclhip.h"
#include "../../include/rmclhip_bench.h"
#include "../../include/rmclhip_lab.h"
#include "bvh_build.h"
#include "devmath.h"
#include "kernels.h"
#include "lab_hooks.h"

using namespace rmclhip;

static_assert(sizeof(rmclhip_transform) == sizeof(xform), "Transform layout");
static_assert(sizeof(rmclhip_cross_statistics) == sizeof(cstats), "CrossStatistics layout");
static_assert(sizeof(rmclhip_particle_attributes) == 36, "ParticleAttributes layout");
static_assert(sizeof(rmclhip_range_measurement) == 64, "RangeMeasurement layout");
static_assert(sizeof(rmclhip_spherical_model) == 32, "SphericalModel layout");

// internal symbols shared by the translation units of the library: never exported
#define RMCL_INTERNAL __attribute__((visibility("hidden")))

extern RMCL_INTERNAL thread_local std::string g_err;   // rmclhip_last_error (capi_map.cpp)

inline rmclhip_status fail(rmclhip_status st, const std::string& msg) {
  g_err = msg;
  return st;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ == kLabMissing)                                                                        \
      return fail(RMCLHIP_ERR_UNSUPPORTED, std::string(#expr) + ": this kernel variant is an experiment that lives in " \
                  "librmclhip_lab.so, which is not loaded (include/rmclhip_lab.h)");              \
    if (e_ != hipSuccess)                                                                         \
      return fail(RMCLHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)

inline xform to_x(const rmclhip_transform* T) {
  xform r;
  std::memcpy(&r, T, sizeof(r));
  return r;
}
inline void from_x(const xform& x, rmclhip_transform* T) { std::memcpy(T, &x, sizeof(x)); }
inline cstats to_cs(const rmclhip_cross_statistics* s) {
  cstats r;
  std::memcpy(&r, s, sizeof(r));
  return r;
}
inline void from_cs(const cstats& c, rmclhip_cross_statistics* s) { std::memcpy(s, &c, sizeof(c)); }

// RMCLHIP_DEBUG=1: report which API call leaves a HIP error behind
struct ApiGuard {
  const char* name;
  explicit ApiGuard(const char* n) : name(n) {}
  ~ApiGuard() {
    static const bool on = std::getenv("RMCLHIP_DEBUG") != nullptr;
    if (on) {
      const hipError_t e = hipPeekAtLastError();
      if (e != hipSuccess) std::fprintf(stderr, "[rmclhip debug] %s leaves HIP error: %s\n", name, hipGetErrorString(e));
    }
  }
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  // grow-only (RCCEmbree.cpp:28-33)
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};


// Upload on the handle's OWN stream, then wait for it.  A plain hipMemcpy runs on the null stream, which the handles'
// hipStreamNonBlocking streams do not synchronise with: a copy from pageable memory may return once the data is staged,
// and a kernel enqueued on the handle's stream right afterwards is then not ordered behind the DMA.  (Observed as a
// 1-in-200 deviation of 5e-7 rad in a correction issued immediately after set_dataset, tools/flaky_g5.py.)
inline hipError_t upload_on(hipStream_t s, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (bytes == 0) return hipSuccess;
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  return e;
}

struct rmclhip_ctx {
  int device = 0;
  hipDeviceProp_t props;
  // map / rcc / pf / resampler handles keep a pointer to their context: each holds a reference, and
  // rmclhip_ctx_destroy only drops the creator's, so destroying the context before its children is safe
  std::atomic<int> refs{1};
  std::atomic<int> wait_block{0};   // rmclhip_ctx_set_wait_mode: 0 = spin on the completion tag, 1 = block in hipStreamSynchronize
  // rmclhip_statistics_p2l (the free function on caller-owned views): stream, partial rows, host-mapped result + tag, created by the
  // first call under the mutex, which also serialises the calls of one context
  std::mutex p2l_mtx;
  hipStream_t p2l_stream = nullptr;
  double* p2l_partials = nullptr;
  size_t p2l_partials_cap = 0;          // doubles
  cstats* p2l_h_stats = nullptr;        // pinned, host-mapped: [0] the result
  cstats* p2l_h_stats_dev = nullptr;
  unsigned long long* p2l_h_done = nullptr;
  unsigned long long* p2l_h_done_dev = nullptr;
  uint32_t* p2l_tickets = nullptr;      // (the fused tail's arrival counter: unused by this path, the kernel's parameter block wants one)
  uint32_t p2l_seq = 0;
  ~rmclhip_ctx();                       // capi_map.cpp
};

inline void ctx_retain(rmclhip_ctx* c) { c->refs.fetch_add(1); }
inline void ctx_release(rmclhip_ctx* c) {
  if (c && c->refs.fetch_sub(1) == 1) delete c;
}

struct rmclhip_map {
  rmclhip_ctx* ctx = nullptr;
  std::atomic<int> refs{1};
  BvhInfo info;
  uint32_t* d_nodes = nullptr;
  uint32_t* d_qnodes = nullptr;  // Node4Q twins
  uint32_t* d_frontier = nullptr;   // frontier table (layout.h kFrontierDepth): n_frontier x 8 dwords {lo.xyz hi.x | hi.yz ref pad}
  uint32_t n_frontier = 0;
  uint32_t* d_frontier_pf = nullptr;   // ... of the filter's tree (find kind 24 walks d_qnodes_pf)
  uint32_t n_frontier_pf = 0;
  uint32_t* d_qnodes_pf = nullptr;  // Node4Q array of the particle filter's own tree (leaves <= kPfLeafTris, same records)
  uint32_t* d_cnodes = nullptr;  // Node4C twins
  uint32_t* d_cnodes16 = nullptr;  // Node16C twins (maps of up to kMaxNodes16 nodes; find kind 32 descends two levels per pass on them), else null
  uint32_t* d_tris = nullptr;
  uint64_t bytes = 0;
  // near grid of the closest-point queries (kernels.h NearGrid): built on the first rmclhip_rcc_find_cpc of any operator of this map
  // near grids (ensure_near_grid): slot 0 = cells near the surface only (scan points), slot 1 = every cell (the filter's beam ends).
  // A slot is built once under the mutex and never changes or moves afterwards -- other operators' launches may be reading it --
  // and both live until the map is released.
  std::mutex grid_mtx;
  struct GridSlot { bool ready = false, failed = false; NearGrid g = {}; };
  GridSlot grid_slot[2];
  std::vector<uint32_t> scene_first_face;  // map_create_scene: first global face id of every instance, + the total (else empty)
};

struct rmclhip_rcc {
  rmclhip_ctx* ctx = nullptr;
  rmclhip_map* map = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  xform Tsb = xidentity();
  // model
  ModelKind kind = kModelNone;
  uint32_t W = 0, H = 0;
  rmclhip_interval range{0.f, 0.f};
  f3 orig{0.f, 0.f, 0.f};
  float pin_fc[4] = {1.f, 1.f, 0.f, 0.f};  // pinhole fx, fy, cx, cy
  DevBuf<float> d_model_tab;
  // params
  float max_dist = 1.0f, adaptive_max_dist_min = 1.0f;
  // dataset
  DevBuf<float> d_ds_points;
  DevBuf<uint8_t> d_ds_mask;
  // what the kernels read: the handle's own copies above, or device memory borrowed from the caller
  // (rmclhip_rcc_set_dataset_view: Correspondences_::dataset lives in the caller's rm::Memory<.., VRAM_HIP>)
  const float* ds_pts = nullptr;
  const uint8_t* ds_msk = nullptr;
  uint32_t n_dataset = 0;
  bool ds_has_mask = false;
  // model buffers
  DevBuf<uint8_t> d_hits;
  DevBuf<float> d_ranges, d_points, d_normals;
  DevBuf<uint32_t> d_face_ids;
  uint32_t n_model = 0;      // per pose
  uint32_t nposes_last = 0;
  bool descent_wide = true;   // ... on the 16-wide twins when the map has them (A/B: rmclhip_rcc_set_descent's max_levels bit 31 clears it)
  uint32_t descent_final_cap = 64, descent_levels = 24;   // kind 32's cooperative descent (rmclhip_rcc_set_descent, include/rmclhip_lab.h)
  // pose batches in world order (kernels.hip launch_batch_tile_order; A/B knob rmclhip_rcc_set_batch_order): keys | sorted keys, values | sorted values, sort scratch
  int xcd_mapping_override = -1;   // A/B knob (rmclhip_rcc_set_descent bits 29..30): -1 = the tuned / default mapping
  uint32_t tuned_xcd_mapping = 0;  // rmclhip_rcc_autotune's choice among 0 / 1 / 2 (FindParams::xcd_mapping); 0 = the default
  uint32_t batch_order = 64;   // 0 = pose-major; else the granule (workgroups per XCD turn)
  DevBuf<uint32_t> d_ord_scratch, d_ord_vals;
  uint32_t descent_leaf_cap = 24;   // kind 32: a wave one of whose rays enters more final leaves than this starts at the root (lab: rmclhip_rcc_set_descent)
  uint32_t out_mask = RMCLHIP_OUT_ALL;   // rmclhip_rcc_set_outputs: which model buffers find / find_batch write
  // reduction
  DevBuf<double> d_partials;
  cstats* h_stats = nullptr;       // pinned, host-mapped
  cstats* h_stats_dev = nullptr;   // device alias of h_stats
  MicpState* d_state = nullptr;
  MicpState* h_state = nullptr;    // pinned, host-mapped
  MicpState* h_state_dev = nullptr;  // device alias of h_state
  uint32_t* d_counter = nullptr;
  uint32_t* d_tickets = nullptr;
  uint32_t* d_loop_barrier = nullptr;  // counter of the persistent-loop grid barrier   // one arrival counter per pose for the fused reduction tail
  // device-resident MICP loop as a static hipGraph: per-call inputs travel in one 256-B H2D copy
  MicpCall* h_call = nullptr;      // pinned
  MicpCall* d_call = nullptr;
  hipGraphExec_t micp_exec = nullptr;
  hipGraph_t micp_graph = nullptr;
  struct MicpKey {
    uint32_t n_iter = 0, W = 0, H = 0, n_dataset = 0;
    int kind = 0, variant = 0, tile = 0, fused = 0, has_mask = 0;
    const void* ptrs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool operator==(const MicpKey& o) const { return std::memcmp(this, &o, sizeof(MicpKey)) == 0; }
  } micp_key, micp_fast_key;
  bool use_graph = true;
  bool fast_graph_dirty = true;    // same, for the moment-form graph
  bool graph_dirty = true;         // set by setModel / set_variant: by-value launch arguments changed
  size_t tickets_cap = 0;
  int loop_blocks = 0;             // MICP loop form (schedule R): 0 one launch per iteration (k_micp_iter), -1 classic
                                   // reduce + solve launches, > 0 persistent k_micp_loop with this many blocks
  bool fused_tail = false;         // true: last-block tail inside the reduction kernel (measured slower, A/B only)
  // moment form of the schedule-(R) loop (launch_micp_fast): tried first when the previous corrections say the gate
  // decisions are stable; the per-iteration form above is the fallback and the reference for the result
  int fast_mode = 1;               // rmclhip_rcc_set_micp_fast: 0 off, 1 automatic with the iterations on the host (default), 2 / 3 / 4 device loops (A/B)
  DevBuf<double> d_fast_partials;
  DevBuf<unsigned long long> d_fast_mask;
  double* d_fold_rows = nullptr;      // hand-over area of the loop launch's folding workgroups (kernels.h: kMicpFoldBlocks)
  uint32_t* d_join_flags = nullptr;   // (first sensor of rmclhip_micp_correct_once) the other sensors' "my rows are complete" words
  hipEvent_t ev_join = nullptr;       // rmclhip_micp_correct_once: this sensor's find (+ moment pass) ran on its own stream; the loop's stream waits for it
  uint32_t* d_fold_flags = nullptr;
  uint32_t last_fast_rows = 0, last_fast_words = 0;   // partial rows / mask words of the last moment-form attempt (diagnostics)
  MicpFastStatus* h_fast_status = nullptr;      // pinned, host-mapped
  MicpFastStatus* h_fast_status_dev = nullptr;
  unsigned long long* h_done = nullptr;         // pinned, host-mapped completion tags: [0] this handle's chains, [1] the N-sensor loop
  unsigned long long* h_done_dev = nullptr;
  uint32_t done_seq = 0;                        // sequence number of the last polled call (never 0 in a tag)
  // closest-point correspondences, tracking: record index per dataset point of the previous find_cpc (see rmclhip_rcc_find_cpc)
  DevBuf<uint32_t> d_cpc_rec;
  const uint32_t* cpc_rec_ptr = nullptr;
  const float* cpc_rec_pts = nullptr;           // the dataset the records were computed for
  uint32_t cpc_rec_n = 0;
  bool cpc_tracking = true;
  bool cpc_bounded = false;        // rmclhip_rcc_set_cpc_bounded: search only within max_dist
  bool cpc_grid = true;            // rmclhip_rcc_set_cpc_grid: points without a tracking seed start from the map's near grid
  hipGraphExec_t micp_fast_exec = nullptr;
  hipGraph_t micp_fast_graph = nullptr;
  float fast_rho_cap = 0.02f, fast_tau_cap = 0.1f;   // bounds on |2 sin(theta/2)| and |t| of the pre-transforms
  uint32_t fast_holdoff = 0;       // corrections to skip the attempt for (after repeated overflows)
  uint32_t fast_overflows = 0;     // consecutive
  rmclhip_micp_fast_info fast_info = {};
  // Round 4 -- iterations on the host (micp_host.h): what k_micp_publish hands over, and the host's verified copy of it
  MicpHostBlock* h_mom = nullptr;      // pinned, host-mapped
  MicpHostBlock* h_mom_dev = nullptr;
  MicpMomentSet mset;                  // valid for the model buffers + dataset it was formed from; dropped by whatever changes either
  bool mset_pending = false;           // a publish is in flight on the stream (speculating find): its tag carries mset_seq
  uint32_t mset_seq = 0;
  float pend_lo = 0.f, pend_hi = 0.f, pend_rho = 0.f, pend_tau = 0.f;   // band and caps the in-flight set is formed for
  uint32_t mset_passes = 0;            // moment passes computeCrossStatistics ran since the last find (at most 2)
  MicpFastStatus last_fast = {};       // outcome of the last moment-form attempt, whichever side ran the iterations
  // the reference's unchanged caller loop (micp_localization.cpp:900-964): find(), then computeCrossStatistics() per iteration
  uint32_t ccs_since_find = 0;         // computeCrossStatistics calls since the last find
  bool ccs_loop = false;               // the last find was followed by such calls: the next find forms the moments in its epilogue
  float ccs_last_maxd = 0.f;           // max_dist' of the last call (the band of the next speculation is centred on it)
  float ccs_max_rho = 0.f, ccs_max_tau = 0.f;   // largest pre-transform since the last find
  rmclhip_ccs_info ccs_info = {};
  // N-sensor loop (rmclhip_micp_correct_once): call block + state of the first sensor, kept between calls
  DevBuf<uint8_t> d_multi_blob;
  MicpMultiState* h_multi_state = nullptr;          // pinned, host-mapped
  MicpMultiState* h_multi_state_dev = nullptr;
  MicpMultiFastStatus* h_multi_status = nullptr;    // pinned, host-mapped
  MicpMultiFastStatus* h_multi_status_dev = nullptr;
  uint32_t multi_holdoff = 0, multi_overflows = 0;
  // batch
  DevBuf<uint8_t> d_raw;           // staged PointCloud2 bytes (set_input_pointcloud2)
  DevBuf<xform> d_Tbm, d_Tsm, d_Tms, d_Tdelta;
  DevBuf<cstats> d_bstats;
  // correct_batch's results leave through pinned, host-mapped staging (grow-only): the solve launch writes them there, the call
  // returns on its completion tag and copies them out -- no device-to-host copy launches, no stream synchronisation
  xform* h_bT = nullptr; xform* h_bT_dev = nullptr; cstats* h_bS = nullptr; cstats* h_bS_dev = nullptr; uint32_t h_batch_cap = 0;
  bool capturing = false;          // inside hipStreamBeginCapture: no synchronisation allowed
  int variant = 15;       // traversal kind: 0 wave-packet, 1 one lane per ray (while-while), 2 four lanes per ray
                          // (quad-cooperative), 15 automatic: quad while the launch is bound by the slowest ray's
                          // chain of dependent fetches (few rays in flight), one lane per ray once the chip is full
  int tile_override = 0;  // 1 + log2(tile width), 0 = automatic
  // rmclhip_rcc_autotune[_batch]: the kind measured fastest for the current (map, model), for single scans / pose batches (0 = the
  // rule), and whether its rays start at the frontier (kinds 23 / 24 without it are round 2's kinds 19 / 22)
  int tuned_kind = 0, tuned_batch_kind = 0;
  int tuned_tile = 0;              // 1 + log2(tile width) measured best by rmclhip_rcc_autotune (0 = the rule's shape)
  int last_moment_find_kind = 0;   // what enqueue_find_with_moments launched last (it may replace the rule's 24 by 23) ...
  bool last_moment_find_tiled = false;   // ... and whether it left one moment row per workgroup (epilogue) or per 1024 elements (pass)
  bool tuned_frontier = true, tuned_batch_frontier = true;
  DevBuf<float> d_tile_planes;     // plane table of the frontier start for the current (model, tiling): 16 floats per tile
  bool tile_planes_ok = false;
  float ang_aspect = 0.0f;         // spherical models: |row spacing / column spacing| in angle (0: unknown -- the other models)
  float last_find_ms = 0.f, last_reduce_ms = 0.f;
  bool reduce_timing_pending = false;
  bool find_timing_pending = false;   // a speculating find returned on its tag: ev0 / ev1 still hold its timing
  bool kernel_timing = false;         // rmclhip_rcc_set_kernel_timing: bracket find / computeCrossStatistics with HIP events (two
                                      // hipEventRecord + one hipEventElapsedTime per call: opt-in since round 4)
};

// A pinned, host-mapped completion tag of a handle whose synchronous calls launch kernels and return nothing through the host (the
// filter's update / motion update, the tournament): a one-thread launch behind the chain stores {seq, 0}, the host polls it -- ~7 us
// sooner than the stream's own completion signal reaches hipStreamSynchronize (measured on the synchronous find, round 4).
struct ChainTag {
  unsigned long long* h = nullptr;
  unsigned long long* d = nullptr;
  uint32_t seq = 0;
  hipError_t create() {
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h), sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0);
    if (e == hipSuccess) *h = 0ull;
    return e;
  }
  void destroy() { if (h) (void)hipHostFree(h); h = nullptr; d = nullptr; }
  // wait for the end of what `stream` holds (BLOCK wait mode, or no tag: hipStreamSynchronize)
  hipError_t wait_chain_end(const rmclhip_ctx* ctx, hipStream_t stream) {
    if (h == nullptr || ctx->wait_block.load(std::memory_order_relaxed)) return hipStreamSynchronize(stream);
    seq = (seq == 0xFFFFFFFFu) ? 1u : seq + 1u;
    if (const hipError_t e = launch_host_tag(d, seq, stream)) return e;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
    volatile const unsigned long long* tag = h;
    for (uint32_t spins = 0;; ++spins) {
      if (*tag == static_cast<unsigned long long>(seq)) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#endif
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() > t_end) return hipStreamSynchronize(stream);
    }
  }
};

// computeCrossStatistics and every correction read {hits, points, normals} of the operator's model buffers (Correspondences.hpp:81-85)
inline bool micp_outputs_selected(const rmclhip_rcc* r) { return (r->out_mask & RMCLHIP_OUT_MICP) == RMCLHIP_OUT_MICP; }
constexpr const char* kNeedMicpOutputs =
    "the operator's {hits, points, normals} outputs are deselected (rmclhip_rcc_set_outputs): nothing to reduce";

// whatever is about to rewrite the model buffers or the dataset: the published moments summarise the old ones
static inline void drop_moment_set(rmclhip_rcc* r) {
  r->mset.valid = false;
  r->mset_pending = false;
  r->mset_passes = 0;
}

struct rmclhip_pf {
  rmclhip_ctx* ctx = nullptr;
  rmclhip_map* map = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  rmclhip_pf_params params{2.0f, 100.0f, 100.0f, 0.0f, {0.05f, 80.0f}, 10000u, 0u};
  DevBuf<float> d_beams;
  float* h_beams = nullptr;  // pinned staging
  hipEvent_t ev_beams = nullptr;      // behind the last copy out of h_beams
  bool beams_copy_pending = false;
  size_t h_beams_cap = 0;
  float* errors_dev = nullptr;
  int variant = 0;
  uint32_t refill_thr = 0, tail_lanes = 8;  // schedule knobs of the round-3 kernel (0: from `refill`); rmclhip_pf_set_schedule
  // rmclhip_pf_set_mapping: 0 beam-minor blocks of ~2048 rays (default), 1 particle-minor blocks (measured neutral, kept for A/B); nothing else is accepted
  bool cpc_grid = true;            // correspondence_type 1: seed every closest-point query from the map's near grid (A/B: rmclhip_pf_set_mapping bit 8 clears it)
  ChainTag tag;                    // completion tag of the synchronous kernel-only calls (update, motion update, extract_weights)
  bool slot_order = false;         // rmclhip_pf_set_variant bit 11
  bool accum = true;               // order-independent likelihood accumulation (round 5 default: no error scratch, no in-order chain); rmclhip_pf_set_variant bit 12 clears it (A/B: rounds 3 / 4)
  DevBuf<double> d_gpow;           // g^i, i = 0 .. n_beams, g = max_n_meas / (max_n_meas + 1): its merge weights
  uint32_t gpow_beams = 0, gpow_max = 0;
  bool evals_global = true;        // k_pf_update_v3 keeps a workgroup's beam errors in global scratch, not LDS (A/B: rmclhip_pf_set_mapping bit 9 clears it)
  DevBuf<float> d_evals;           // [n_particles * n_beams], grow-only
  int mapping = 0;
  uint32_t map_ppb = 0;            // particles per workgroup of the particle-minor mapping (0: 32)
  const uint32_t* order = nullptr; // slot -> particle (device), borrowed or d_order
  uint32_t order_n = 0;
  DevBuf<uint32_t> d_order;
  bool beams_at_origin = false;  // of the beams uploaded last: all start at the sensor origin
  bool legacy = false;      // A/B: the round-2 kernel (k_pf_update_persist)
  bool big_blocks = false;  // A/B: 4096 rays per workgroup
  bool pf_tree = true;      // quantised nodes of the filter's own tree (leaves <= kPfLeafTris); false: the map's tree (A/B)
  bool full_nodes = false;  // A/B: persistent lanes on the 128-B nodes instead of their 64-B quantised twins
  int refill = 4;  // 0: rounds of one ray per lane; 1..4: persistent lanes (dynamic ray fetch), refill when 8/16/32/48
                   // lanes of a wave are idle (default 48: the refill block also evaluates the finished beams, which
                   // pays off with many lanes at once; measured best on sphere and room)
};

// GladiatorResamplerGPU analogue: owns a stream and the scratch of the {sum, max} reduction
struct rmclhip_resampler {
  rmclhip_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  DevBuf<double> d_psum;
  DevBuf<float> d_pmax, d_out;
  float* h_out = nullptr;  // pinned {sum, max}
  ChainTag tag;            // completion tag of the tournament (a kernel-only synchronous call)
  unsigned long long* h_res = nullptr;   // pinned: residual resampling's {sum, max, expect, n_draws} + the draws' total (5 words)
  // residual resampling: {double sum, double max, u64 expect, u64 n_draws} on the device, the draws' particle / count / prefix sums
  DevBuf<unsigned long long> d_res_stats, d_res_incl, d_res_btot;
  DevBuf<uint32_t> d_res_idx, d_res_cnt;
};


struct ReduceTail {
  uint32_t mode = kTailNone;
  const MicpCall* call = nullptr;
  cstats* stats_out = nullptr;
  xform Tbo = xidentity();
  MicpState* state = nullptr;
  xform* Tdelta_out = nullptr;
  unsigned long long* done = nullptr;   // host-mapped completion tag (kTailStats, one pose, unfused tail)
  uint32_t seq = 0;                     // ... and the sequence number it must carry
};

// what the tag's sum covers: `base` always; the `extra` blocks only when *code == 0 (a status block's "done", the exits that
// also wrote a state block) or when there is no code word
struct DoneCheck {
  const void* base = nullptr; size_t base_bytes = 0;
  const volatile uint32_t* code = nullptr;
  const void* extra[3] = {nullptr, nullptr, nullptr}; size_t extra_bytes[3] = {0, 0, 0};
};

// residual resampling (ResidualResamplerCPU.cpp:55-203) in three enqueue phases, each followed by ONE wait of the caller:
//   A prepare: statistics + how many copies a draw inserts on average  -> h_res[0..3]
//   B draws  : a block of draws that fills the cloud with a margin     -> h_res[4] = copies these draws insert (repeat doubled if short)
//   C fill   : the slots [first, first + count) of the new cloud
struct ResidualJob {
  rmclhip_resampler* r = nullptr;
  hipStream_t st = nullptr;
  const rmclhip_transform* poses = nullptr; const rmclhip_particle_attributes* attrs = nullptr;
  rmclhip_transform* poses_new = nullptr; rmclhip_particle_attributes* attrs_new = nullptr;
  uint32_t n_particles = 0, n_new = 0, first = 0, count = 0;
  const rmclhip_gladiator_config* cfg = nullptr;
  uint64_t seed = 0; uint32_t step = 0;
  double want = 0.0;
  uint32_t n_draws = 0;
  bool filled = false;     // phase B's draws cover the new cloud
  bool active = false;     // count != 0 && n_new != 0
};

// ---- defined in one file, called from another --------------------------------------------------------
RMCL_INTERNAL rmclhip_status rebuild_tile_planes(rmclhip_rcc* r, bool keep_tuning = false);
RMCL_INTERNAL int find_variant(const rmclhip_rcc* r, uint32_t nposes);
RMCL_INTERNAL void fill_find_params(rmclhip_rcc* r, FindParams& p, uint32_t nposes, int kind = -1);
RMCL_INTERNAL rmclhip_status ensure_model_buffers(rmclhip_rcc* r, size_t n_total);
RMCL_INTERNAL rmclhip_status find_enqueue(rmclhip_rcc* r, const xform& Tbm, bool* speculate = nullptr);
RMCL_INTERNAL rmclhip_status reduce_enqueue(rmclhip_rcc* r, const xform& Tpre, const xform* Tpre_dev, float max_dist, uint32_t nposes, const ReduceTail& tail);
RMCL_INTERNAL rmclhip_status find_batch_enqueue(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes);
RMCL_INTERNAL rmclhip_status batch_order_enqueue(rmclhip_rcc* r, FindParams& p, int variant);   // pose batches in world order (capi_rcc_tune.cpp)
RMCL_INTERNAL rmclhip_status ensure_near_grid(rmclhip_map* m, hipStream_t stream, bool full, const NearGrid** out);
RMCL_INTERNAL rmclhip_status map_upload(rmclhip_ctx* ctx, const BvhHost& bvh, rmclhip_map** out);
RMCL_INTERNAL rmclhip_status gladiator_enqueue(rmclhip_resampler* r, const rmclhip_transform* poses_dev, const rmclhip_particle_attributes* attrs_dev, uint32_t n_particles, rmclhip_transform* poses_new_dev, rmclhip_particle_attributes* attrs_new_dev, uint32_t first, uint32_t count, const rmclhip_gladiator_config* cfg, uint64_t seed, uint32_t step, hipStream_t st);
RMCL_INTERNAL rmclhip_status residual_prepare_enqueue(ResidualJob& j);
RMCL_INTERNAL rmclhip_status residual_fill_enqueue(ResidualJob& j, bool want_n_draws);
RMCL_INTERNAL rmclhip_status residual_draws_enqueue(ResidualJob& j, bool first_try);
RMCL_INTERNAL void residual_draws_done(ResidualJob& j);
RMCL_INTERNAL rmclhip_status residual_check(ResidualJob& j);
extern RMCL_INTERNAL std::atomic<unsigned long long> g_tag_sum_retries;   // capi_rcc.cpp: polls that met their sequence number before the checksum matched
//@@DECLS@@
