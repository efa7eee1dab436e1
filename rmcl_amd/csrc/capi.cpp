// capi.cpp -- implementation of the C ABI declared in include/rmclhip.h.
// Host-side orchestration only: device memory, streams, launches.  There is no CPU compute
// path here: without a HIP device every compute entry point fails with RMCLHIP_ERR_NO_DEVICE.
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

namespace {

thread_local std::string g_err;

rmclhip_status fail(rmclhip_status st, const std::string& msg) {
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

}  // namespace

// Upload on the handle's OWN stream, then wait for it.  A plain hipMemcpy runs on the null stream, which the handles'
// hipStreamNonBlocking streams do not synchronise with: a copy from pageable memory may return once the data is staged,
// and a kernel enqueued on the handle's stream right afterwards is then not ordered behind the DMA.  (Observed as a
// 1-in-200 deviation of 5e-7 rad in a correction issued immediately after set_dataset, tools/flaky_g5.py.)
static inline hipError_t upload_on(hipStream_t s, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
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
};

namespace {
inline void ctx_retain(rmclhip_ctx* c) { c->refs.fetch_add(1); }
inline void ctx_release(rmclhip_ctx* c) {
  if (c && c->refs.fetch_sub(1) == 1) delete c;
}
}  // namespace

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

extern "C" {

static rmclhip_status map_upload(rmclhip_ctx* ctx, const BvhHost& bvh, rmclhip_map** out);

const char* rmclhip_last_error(void) { return g_err.c_str(); }
const char* rmclhip_version(void) { return "rmclhip 0.1 (gfx950)"; }

// ---- context ---------------------------------------------------------------------------------
rmclhip_status rmclhip_ctx_create(int device, rmclhip_ctx** out) {
  ApiGuard guard_("rmclhip_ctx_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "ctx_create: out is null");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(RMCLHIP_ERR_NO_DEVICE, "no HIP device available (librmclhip has no CPU fallback)");
  if (device < 0 || device >= count) return fail(RMCLHIP_ERR_INVALID, "ctx_create: device index out of range");
  HIPCHK(hipSetDevice(device));
  rmclhip_ctx* c = new rmclhip_ctx();
  c->device = device;
  e = hipGetDeviceProperties(&c->props, device);
  if (e != hipSuccess) {
    delete c;
    return fail(RMCLHIP_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  }
  *out = c;
  return RMCLHIP_OK;
}

void rmclhip_ctx_destroy(rmclhip_ctx* ctx) { ctx_release(ctx); }

rmclhip_status rmclhip_ctx_set_wait_mode(rmclhip_ctx* ctx, int mode) {
  ApiGuard guard_("rmclhip_ctx_set_wait_mode");
  if (!ctx || (mode != RMCLHIP_WAIT_SPIN && mode != RMCLHIP_WAIT_BLOCK)) return fail(RMCLHIP_ERR_INVALID, "ctx_set_wait_mode: bad arguments");
  ctx->wait_block.store(mode == RMCLHIP_WAIT_BLOCK ? 1 : 0);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_ctx_device_name(rmclhip_ctx* ctx, char* buf, size_t n) {
  ApiGuard guard_("rmclhip_ctx_device_name");
  if (!ctx || !buf || n == 0) return fail(RMCLHIP_ERR_INVALID, "ctx_device_name: bad arguments");
  std::snprintf(buf, n, "%s (%s, %d CUs)", ctx->props.name, ctx->props.gcnArchName, ctx->props.multiProcessorCount);
  return RMCLHIP_OK;
}

// ---- map -------------------------------------------------------------------------------------
static void fill_info(const BvhInfo& bi, uint64_t bytes, rmclhip_map_info* out) {
  std::memset(out, 0, sizeof(*out));
  out->n_faces = bi.n_faces;
  out->n_vertices = bi.n_vertices;
  out->n_nodes = bi.n_nodes;
  out->n_tri_records = bi.n_faces;
  out->max_depth = bi.max_depth;
  out->stack_need = bi.stack_need;
  out->device_bytes = bytes;
  for (int k = 0; k < 3; ++k) { out->bbox_min[k] = bi.bbox_min[k]; out->bbox_max[k] = bi.bbox_max[k]; }
  out->height_fallbacks = bi.height_fallbacks;
  out->guarded_nodes = bi.guarded_nodes;
}

rmclhip_status rmclhip_bvh_build_host(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                                      rmclhip_map_info* info, uint32_t* nodes_out, size_t nodes_cap,
                                      uint32_t* tris_out, size_t tris_cap) {
  ApiGuard guard_("rmclhip_bvh_build_host");
  BvhHost bvh;
  const std::string err = build_bvh(v, nv, f, nf, bvh);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host: " + err);
  const size_t nd = bvh.nodes.size() * kNodeDwords, td = bvh.tris.size() * kTriDwords;
  if (info) fill_info(bvh.info, (nd + td) * 4, info);
  if (nodes_out) {
    if (nodes_cap < nd) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host: nodes buffer too small");
    std::memcpy(nodes_out, bvh.nodes.data(), nd * 4);
  }
  if (tris_out) {
    if (tris_cap < td) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host: tris buffer too small");
    std::memcpy(tris_out, bvh.tris.data(), td * 4);
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_bvh_build_host_pf(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                                         rmclhip_map_info* info, uint32_t* nodes_out, size_t nodes_cap,
                                         uint32_t* qnodes_out, size_t qnodes_cap) {
  ApiGuard guard_("rmclhip_bvh_build_host_pf");
  BvhHost bvh;
  const std::string err = build_bvh(v, nv, f, nf, bvh);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host_pf: " + err);
  const size_t nd = bvh.nodes_pf.size() * kNodeDwords, qd = bvh.qnodes_pf.size() * (sizeof(Node4Q) / 4);
  if (info) {
    fill_info(bvh.info, (nd + bvh.tris.size() * kTriDwords) * 4, info);
    info->n_nodes = bvh.info.n_nodes_pf;
    info->max_depth = bvh.info.max_depth_pf;
    info->stack_need = bvh.info.stack_need_pf;
  }
  if (nodes_out) {
    if (nodes_cap < nd) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host_pf: nodes buffer too small");
    std::memcpy(nodes_out, bvh.nodes_pf.data(), nd * 4);
  }
  if (qnodes_out) {
    if (qnodes_cap < qd) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host_pf: qnodes buffer too small");
    std::memcpy(qnodes_out, bvh.qnodes_pf.data(), qd * 4);
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_bvh_build_host_quantised(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                                                uint32_t* qnodes_out, size_t qnodes_cap) {
  ApiGuard guard_("rmclhip_bvh_build_host_quantised");
  if (!qnodes_out) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host_quantised: null");
  BvhHost bvh;
  const std::string err = build_bvh(v, nv, f, nf, bvh);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host_quantised: " + err);
  const size_t qd = bvh.qnodes.size() * (sizeof(Node4Q) / 4);
  if (qnodes_cap < qd) return fail(RMCLHIP_ERR_INVALID, "bvh_build_host_quantised: buffer too small");
  std::memcpy(qnodes_out, bvh.qnodes.data(), qd * 4);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_map_create(rmclhip_ctx* ctx, const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                                  rmclhip_map** out) {
  ApiGuard guard_("rmclhip_map_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "map_create: out is null");
  *out = nullptr;
  if (!ctx) return fail(RMCLHIP_ERR_INVALID, "map_create: ctx is null");
  BvhHost bvh;
  const std::string err = build_bvh(v, nv, f, nf, bvh);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "map_create: " + err);
  return map_upload(ctx, bvh, out);
}

// device copy of a built BVH (one build can serve several devices: rmclhip_pf_sharded_create)
static rmclhip_status map_upload(rmclhip_ctx* ctx, const BvhHost& bvh, rmclhip_map** out) {
  // an assertion since round 5: build_bvh bounds the height of the binary tree and collapses tallest-first where needed, so no mesh
  // can produce a deeper stack (bvh_build.cpp: kMaxHeight2)
  if (bvh.info.stack_need > 64 || bvh.info.stack_need_pf > 64)
    return fail(RMCLHIP_ERR_INVALID, "map_create: internal error, the BVH builder exceeded its own 64-entry stack bound");
  if (static_cast<uint64_t>(bvh.nodes.size()) * sizeof(Node4) >= (1ull << 32))
    return fail(RMCLHIP_ERR_UNSUPPORTED, "map_create: node array exceeds 4 GB (the kernels address nodes with 32-bit byte offsets)");
  HIPCHK(hipSetDevice(ctx->device));
  rmclhip_map* m = new rmclhip_map();
  m->ctx = ctx;
  m->info = bvh.info;
  const size_t nb = bvh.nodes.size() * sizeof(Node4), tb = bvh.tris.size() * sizeof(TriRec);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_nodes), nb);
  // + 3 zeroed records: the packet kernel always requests a full 4-record leaf
  const size_t tb_pad = (kMaxLeafTris - 1) * sizeof(TriRec);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_tris), tb + tb_pad);
  if (e == hipSuccess) e = hipMemset(reinterpret_cast<char*>(m->d_tris) + tb, 0, tb_pad);
  if (e == hipSuccess) e = hipMemcpy(m->d_nodes, bvh.nodes.data(), nb, hipMemcpyHostToDevice);
  const size_t qb = bvh.qnodes.size() * sizeof(Node4Q);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_qnodes), qb);
  if (e == hipSuccess) e = hipMemcpy(m->d_qnodes, bvh.qnodes.data(), qb, hipMemcpyHostToDevice);
  const size_t qpb = bvh.qnodes_pf.size() * sizeof(Node4Q);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_qnodes_pf), qpb);
  if (e == hipSuccess) e = hipMemcpy(m->d_qnodes_pf, bvh.qnodes_pf.data(), qpb, hipMemcpyHostToDevice);
  const size_t fb = bvh.frontier.size() * sizeof(Node4C::Child);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_frontier), std::max<size_t>(fb, 32));
  if (e == hipSuccess && fb) e = hipMemcpy(m->d_frontier, bvh.frontier.data(), fb, hipMemcpyHostToDevice);
  m->n_frontier = static_cast<uint32_t>(bvh.frontier.size());
  const size_t fpb = bvh.frontier_pf.size() * sizeof(Node4C::Child);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_frontier_pf), std::max<size_t>(fpb, 32));
  if (e == hipSuccess && fpb) e = hipMemcpy(m->d_frontier_pf, bvh.frontier_pf.data(), fpb, hipMemcpyHostToDevice);
  m->n_frontier_pf = static_cast<uint32_t>(bvh.frontier_pf.size());
  const size_t cb = bvh.cnodes.size() * sizeof(Node4C);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_cnodes), cb);
  if (e == hipSuccess) e = hipMemcpy(m->d_cnodes, bvh.cnodes.data(), cb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_tris, bvh.tris.data(), tb, hipMemcpyHostToDevice);
  // the map is read by kernels on the handles' non-blocking streams, which do not synchronise with the null stream these
  // copies ran on: make sure every byte has landed before the handle is handed out
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (m->d_nodes) (void)hipFree(m->d_nodes);
    if (m->d_qnodes) (void)hipFree(m->d_qnodes);
    if (m->d_qnodes_pf) (void)hipFree(m->d_qnodes_pf);
    if (m->d_frontier) (void)hipFree(m->d_frontier);
    if (m->d_frontier_pf) (void)hipFree(m->d_frontier_pf);
    if (m->d_cnodes) (void)hipFree(m->d_cnodes);
    if (m->d_tris) (void)hipFree(m->d_tris);
    delete m;
    return fail(e == hipErrorOutOfMemory ? RMCLHIP_ERR_NOMEM : RMCLHIP_ERR_HIP,
                std::string("map_create upload: ") + hipGetErrorString(e));
  }
  m->bytes = nb + qb + qpb + cb + tb + fb + fpb;
  ctx_retain(ctx);
  *out = m;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_map_retain(rmclhip_map* map) {
  ApiGuard guard_("rmclhip_map_retain");
  if (!map) return fail(RMCLHIP_ERR_INVALID, "map_retain: null");
  map->refs.fetch_add(1);
  return RMCLHIP_OK;
}

void rmclhip_map_release(rmclhip_map* map) {
  ApiGuard guard_("rmclhip_map_release");
  if (!map) return;
  if (map->refs.fetch_sub(1) == 1) {
    (void)hipSetDevice(map->ctx->device);
    if (map->d_nodes) (void)hipFree(map->d_nodes);
    if (map->d_qnodes) (void)hipFree(map->d_qnodes);
    if (map->d_qnodes_pf) (void)hipFree(map->d_qnodes_pf);
    if (map->d_frontier) (void)hipFree(map->d_frontier);
    if (map->d_frontier_pf) (void)hipFree(map->d_frontier_pf);
    if (map->d_cnodes) (void)hipFree(map->d_cnodes);
    if (map->d_tris) (void)hipFree(map->d_tris);
    for (auto& gs : map->grid_slot)
      if (gs.g.cells) (void)hipFree(const_cast<uint32_t*>(gs.g.cells));
    ctx_release(map->ctx);
    delete map;
  }
}

// ---- scenes: several meshes, placed (and possibly repeated) by affine transforms -----------------
// The reference hands a whole assimp scene to rm::import_embree_map / import_optix_map (micp_localization.cpp:187-195), which
// instance every mesh under its node's transform.  The hot path only ever sees world-space triangles, so the scene is flattened
// once on the host -- vertices transformed in float like Embree's / OptiX's instance transforms would at build time, faces
// renumbered -- and ONE tree is built over it (no two-level traversal: an instance costs its triangles again, which for the
// maps RMCL localises in -- a few static meshes -- is the cheaper side of the trade).
extern "C++" {
namespace {
std::string scene_flatten(const rmclhip_mesh* meshes, uint32_t n_meshes, const rmclhip_instance* inst, uint32_t n_inst,
                          std::vector<float>& v, std::vector<uint32_t>& f, std::vector<uint32_t>& first_face) {
#pragma clang fp contract(off)
  if (!meshes || n_meshes == 0) return "no meshes";
  for (uint32_t m = 0; m < n_meshes; ++m) {
    if ((meshes[m].n_vertices && !meshes[m].vertices_xyz) || (meshes[m].n_faces && !meshes[m].faces_ijk))
      return "mesh " + std::to_string(m) + ": null array";
    for (uint32_t k = 0; k < 3u * meshes[m].n_faces; ++k)
      if (meshes[m].faces_ijk[k] >= meshes[m].n_vertices)
        return "mesh " + std::to_string(m) + ": face " + std::to_string(k / 3u) + " references vertex " +
               std::to_string(meshes[m].faces_ijk[k]) + " of " + std::to_string(meshes[m].n_vertices);
  }
  const uint32_t n = inst ? n_inst : n_meshes;
  uint64_t nv = 0, nf = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t m = inst ? inst[i].mesh : i;
    if (m >= n_meshes) return "instance " + std::to_string(i) + ": mesh index " + std::to_string(m) + " of " + std::to_string(n_meshes);
    nv += meshes[m].n_vertices;
    nf += meshes[m].n_faces;
  }
  if (nv >= (1ull << 32) || nf >= (1ull << 32)) return "scene exceeds 2^32 vertices or faces";
  v.resize(3 * nv);
  f.resize(3 * nf);
  first_face.assign(n + 1u, 0u);
  size_t vo = 0, fo = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const rmclhip_mesh& M = meshes[inst ? inst[i].mesh : i];
    const float* A = inst ? inst[i].transform : nullptr;
    first_face[i] = static_cast<uint32_t>(fo);
    for (uint32_t k = 0; k < M.n_vertices; ++k) {
      const float x = M.vertices_xyz[3 * k], y = M.vertices_xyz[3 * k + 1], z = M.vertices_xyz[3 * k + 2];
      float* o = &v[3 * (vo + k)];
      if (A) {
        // row by row, left to right, no contraction: the flattening is part of the parity surface (tests restate it in numpy)
        for (int r = 0; r < 3; ++r) {
          float acc = A[4 * r] * x;
          acc = acc + A[4 * r + 1] * y;
          acc = acc + A[4 * r + 2] * z;
          o[r] = acc + A[4 * r + 3];
        }
      } else {
        o[0] = x; o[1] = y; o[2] = z;
      }
    }
    for (uint32_t k = 0; k < 3u * M.n_faces; ++k) f[3 * fo + k] = M.faces_ijk[k] + static_cast<uint32_t>(vo);
    vo += M.n_vertices;
    fo += M.n_faces;
  }
  first_face[n] = static_cast<uint32_t>(fo);
  return std::string();
}
}  // namespace
}  // extern "C++"

rmclhip_status rmclhip_scene_flatten_host(const rmclhip_mesh* meshes, uint32_t n_meshes, const rmclhip_instance* instances,
                                          uint32_t n_instances, float* vertices_out, size_t vertices_cap_floats,
                                          uint32_t* faces_out, size_t faces_cap_dwords, uint32_t* first_face_out,
                                          size_t first_face_cap, uint32_t* n_vertices, uint32_t* n_faces) {
  ApiGuard guard_("rmclhip_scene_flatten_host");
  std::vector<float> v;
  std::vector<uint32_t> f, ff;
  const std::string err = scene_flatten(meshes, n_meshes, instances, n_instances, v, f, ff);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "scene_flatten_host: " + err);
  if (n_vertices) *n_vertices = static_cast<uint32_t>(v.size() / 3);
  if (n_faces) *n_faces = static_cast<uint32_t>(f.size() / 3);
  if ((vertices_out && vertices_cap_floats < v.size()) || (faces_out && faces_cap_dwords < f.size()) ||
      (first_face_out && first_face_cap < ff.size()))
    return fail(RMCLHIP_ERR_INVALID, "scene_flatten_host: output buffer too small");
  if (vertices_out) std::memcpy(vertices_out, v.data(), v.size() * sizeof(float));
  if (faces_out) std::memcpy(faces_out, f.data(), f.size() * sizeof(uint32_t));
  if (first_face_out) std::memcpy(first_face_out, ff.data(), ff.size() * sizeof(uint32_t));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_map_create_scene(rmclhip_ctx* ctx, const rmclhip_mesh* meshes, uint32_t n_meshes,
                                        const rmclhip_instance* instances, uint32_t n_instances, rmclhip_map** out) {
  ApiGuard guard_("rmclhip_map_create_scene");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "map_create_scene: out is null");
  *out = nullptr;
  if (!ctx) return fail(RMCLHIP_ERR_INVALID, "map_create_scene: ctx is null");
  std::vector<float> v;
  std::vector<uint32_t> f, ff;
  std::string err = scene_flatten(meshes, n_meshes, instances, n_instances, v, f, ff);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "map_create_scene: " + err);
  BvhHost bvh;
  err = build_bvh(v.data(), static_cast<uint32_t>(v.size() / 3), f.data(), static_cast<uint32_t>(f.size() / 3), bvh);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "map_create_scene: " + err);
  const rmclhip_status st = map_upload(ctx, bvh, out);
  if (st == RMCLHIP_OK) (*out)->scene_first_face = std::move(ff);
  return st;
}

rmclhip_status rmclhip_map_scene_instances(const rmclhip_map* map, uint32_t* first_face_out, size_t cap, uint32_t* n_instances) {
  ApiGuard guard_("rmclhip_map_scene_instances");
  if (!map) return fail(RMCLHIP_ERR_INVALID, "map_scene_instances: map is null");
  // a map made by rmclhip_map_create is one instance of one mesh
  const std::vector<uint32_t> single{0u, map->info.n_faces};
  const std::vector<uint32_t>& t = map->scene_first_face.empty() ? single : map->scene_first_face;
  if (n_instances) *n_instances = static_cast<uint32_t>(t.size() - 1);
  if (first_face_out) {
    if (cap < t.size()) return fail(RMCLHIP_ERR_INVALID, "map_scene_instances: first_face_out needs n_instances + 1 entries");
    std::memcpy(first_face_out, t.data(), t.size() * sizeof(uint32_t));
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_map_scene_locate(const rmclhip_map* map, uint32_t face_id, uint32_t* instance, uint32_t* local_face) {
  ApiGuard guard_("rmclhip_map_scene_locate");
  if (!map) return fail(RMCLHIP_ERR_INVALID, "map_scene_locate: map is null");
  if (face_id >= map->info.n_faces) return fail(RMCLHIP_ERR_INVALID, "map_scene_locate: face id out of range");
  uint32_t i = 0, first = 0;
  if (!map->scene_first_face.empty()) {
    const auto it = std::upper_bound(map->scene_first_face.begin(), map->scene_first_face.end(), face_id);
    i = static_cast<uint32_t>(it - map->scene_first_face.begin()) - 1u;
    first = map->scene_first_face[i];
  }
  if (instance) *instance = i;
  if (local_face) *local_face = face_id - first;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_map_get_info(const rmclhip_map* map, rmclhip_map_info* out) {
  ApiGuard guard_("rmclhip_map_get_info");
  if (!map || !out) return fail(RMCLHIP_ERR_INVALID, "map_get_info: null");
  fill_info(map->info, map->bytes, out);
  return RMCLHIP_OK;
}

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

static rmclhip_status rebuild_tile_planes(rmclhip_rcc* r, bool keep_tuning = false);

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
  r->tuned_kind = r->tuned_batch_kind = 0; r->tuned_frontier = r->tuned_batch_frontier = true; r->tuned_tile = 0;
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

static rmclhip_status ensure_model_buffers(rmclhip_rcc* r, size_t n_total) {
  drop_moment_set(r);   // every find form comes through here first
  HIPCHK(r->d_hits.reserve(n_total));
  HIPCHK(r->d_ranges.reserve(n_total));
  HIPCHK(r->d_points.reserve(3 * n_total));
  HIPCHK(r->d_normals.reserve(3 * n_total));
  HIPCHK(r->d_face_ids.reserve(n_total));
  return RMCLHIP_OK;
}

// traversal kind of a launch of `nposes` scans (tools/latency_explore.py, tools/perf_explore.py)
static int find_variant(const rmclhip_rcc* r, uint32_t nposes) {
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
static void fill_find_params(rmclhip_rcc* r, FindParams& p, uint32_t nposes, int kind = -1) {
  const int v = kind >= 0 ? kind : find_variant(r, nposes);
  std::memset(&p, 0, sizeof(p));
  p.nodes = r->map->d_nodes;
  p.qnodes = r->map->d_qnodes;
  p.cnodes = r->map->d_cnodes;
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
  p.hits = r->d_hits.p; p.ranges = r->d_ranges.p; p.points = r->d_points.p; p.normals = r->d_normals.p;
  p.face_ids = r->d_face_ids.p;
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
    p.frontier_max_preload = (need < 64u) ? 64u - need : 0u;
    if (p.frontier_max_preload < 2u) p.tile_planes = nullptr;
  }
}

// The frontier start's plane table belongs to (model, tiling): rebuilt -- one small launch on the handle's stream -- by whatever
// changes either (the model setters, set_variant's tile shape), never inside a find (finds are captured into graphs).
static rmclhip_status rebuild_tile_planes(rmclhip_rcc* r, bool keep_tuning) {
  r->tile_planes_ok = false;
  if (!keep_tuning) {
    r->tuned_kind = r->tuned_batch_kind = 0;   // a measurement belongs to the model it was taken with
    r->tuned_frontier = r->tuned_batch_frontier = true;
    r->tuned_tile = 0;
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
static rmclhip_status find_enqueue(rmclhip_rcc* r, const xform& Tbm, bool* speculate = nullptr) {
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
    if (r->ccs_loop && r->fast_mode == 1 && !r->fused_tail && r->n_dataset != 0u && (fv == 23 || fv == 2) && r->ccs_last_maxd == r->ccs_last_maxd) {
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
static rmclhip_status ensure_near_grid(rmclhip_map* m, hipStream_t stream, bool full, const NearGrid** out) {
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
                        nullptr, false, stream, nullptr, d_coarse, m->info.n_faces, 3.0e38f, nullptr, &c);
  c.cells = d_coarse;
  if (e == hipSuccess)
    e = launch_cpc_find(m->d_nodes, m->d_tris, nullptr, static_cast<uint32_t>(total), 0.f, xidentity(), xidentity(), nullptr, nullptr, nullptr, nullptr,
                        nullptr, false, stream, nullptr, d_cells, m->info.n_faces, 3.0e38f, &c, &g, skip_d2);
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
                         r->max_dist, Tsm, xinv(Tsm), r->d_hits.p, r->d_ranges.p, r->d_points.p, r->d_normals.p,
                         r->d_face_ids.p, quad, r->stream, seed, r->cpc_tracking ? r->d_cpc_rec.p : nullptr, r->map->info.n_faces,
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
  HIPCHK(stream_wait(r->ctx, r->stream));
  return RMCLHIP_OK;
}

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

static rmclhip_status reduce_enqueue(rmclhip_rcc* r, const xform& Tpre, const xform* Tpre_dev, float max_dist,
                                     uint32_t nposes, const ReduceTail& tail) {
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
// A flag alone is NOT a sound hand-off here: round 3 measured (tools/determinism2.py, 1 in ~10^4 calls) the host seeing the
// flag of the current call while the result block -- written before the kernel's __threadfence_system(), but to another host
// allocation -- still held the previous call's values.  So the tag carries {sequence number of the call, xor of every
// result word}: the result is accepted only when the sequence number is this call's AND the words the host reads add up to
// the tag's sum; otherwise polling continues.  20 ms without an acceptable tag, or wait mode "block"
// (rmclhip_ctx_set_wait_mode), falls back to the stream.
static inline uint32_t xor_host(const void* p, size_t bytes) {
  const volatile uint32_t* w = static_cast<const volatile uint32_t*>(p);
  uint32_t x = 0;
  for (size_t i = 0; i < bytes / 4; ++i) x ^= w[i];
  return x;
}
// how often a poller saw ITS sequence number in the tag while the result words did not (yet) add up to the tag's sum: the event the
// checksum exists for (rmclhip_debug_tag_retries; profiles/r05_tag_handoff.txt)
static std::atomic<unsigned long long> g_tag_sum_retries{0};

static inline uint32_t next_seq(rmclhip_rcc* r) {
  if (++r->done_seq == 0u) r->done_seq = 1u;
  return r->done_seq;
}
// what the tag's sum covers: `base` always; the `extra` blocks only when *code == 0 (a status block's "done", the exits that
// also wrote a state block) or when there is no code word
struct DoneCheck {
  const void* base = nullptr; size_t base_bytes = 0;
  const volatile uint32_t* code = nullptr;
  const void* extra[3] = {nullptr, nullptr, nullptr}; size_t extra_bytes[3] = {0, 0, 0};
};
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
  r->last_moment_find_tiled = epilogue_allowed && (fv == 23 || fv == 2);
  if (epilogue_allowed && (fv == 23 || fv == 2)) {
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
  if (hits) *hits = r->d_hits.p;
  if (ranges) *ranges = r->d_ranges.p;
  if (points) *points = r->d_points.p;
  if (normals) *normals = r->d_normals.p;
  if (face_ids) *face_ids = r->d_face_ids.p;
  if (n) *n = r->n_model * (r->nposes_last ? r->nposes_last : 1);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_correct_once(rmclhip_rcc* r, const rmclhip_transform* Tom_, const rmclhip_transform* Tbo_,
                                        uint32_t n_iter, double convergence_progress, int refind_each_iteration,
                                        rmclhip_transform* T_out, rmclhip_cross_statistics* stats_out) {
  ApiGuard guard_("rmclhip_rcc_correct_once");
  if (!r || !Tom_ || !Tbo_ || !T_out) return fail(RMCLHIP_ERR_INVALID, "correct_once: null");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "correct_once: no sensor model");
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
        const bool tiled = r->fast_mode != 3 && (fv == 23 || fv == 2);   // the find forms the moments in its epilogue
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

static rmclhip_status find_batch_enqueue(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes);

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

// ---- pose batches sharded over devices (north_star: pose-corrections/s at 1/2/4/8 GPUs; SURVEY 8(e): "MICP pose batches: shard
// poses, no exchange at all").  One process, one operator replica per device over ONE host BVH build (as rmclhip_pf_sharded_create
// does for the filter); the poses of a batch are block-partitioned with shard_bounds, every replica's chain (pose upload, find over
// its block, reduction, per-pose solve, results to pinned host memory) is ENQUEUED before any is waited for, so the devices run
// concurrently; no collective, hence no RCCL.  Replaces the loop of lidar_corrector_optix_benchmark.cpp:86-133 (1000 poses per
// correct()) when one GPU is not enough.
}  // extern "C"

extern "C" {
static void shard_bounds(uint32_t n, uint32_t rank, uint32_t world, uint32_t* lo, uint32_t* hi);
}

struct RccRank {
  rmclhip_ctx* ctx = nullptr;
  rmclhip_map* map = nullptr;
  rmclhip_rcc* rcc = nullptr;
  xform* h_Tdelta = nullptr;   // pinned staging of this replica's block
  cstats* h_stats = nullptr;
  uint32_t cap = 0;
};
struct rmclhip_rcc_sharded {
  std::vector<RccRank> ranks;
};

extern "C" {

void rmclhip_rcc_sharded_destroy(rmclhip_rcc_sharded* h) {
  if (!h) return;
  for (RccRank& R : h->ranks) {
    if (R.ctx) (void)hipSetDevice(R.ctx->device);
    if (R.h_Tdelta) (void)hipHostFree(R.h_Tdelta);
    if (R.h_stats) (void)hipHostFree(R.h_stats);
    if (R.rcc) rmclhip_rcc_destroy(R.rcc);
    if (R.map) rmclhip_map_release(R.map);
    if (R.ctx) rmclhip_ctx_destroy(R.ctx);
  }
  delete h;
}

rmclhip_status rmclhip_rcc_sharded_create(const int* devices, uint32_t ndev, const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                                          rmclhip_rcc_sharded** out) {
  ApiGuard guard_("rmclhip_rcc_sharded_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_create: out is null");
  *out = nullptr;
  if (ndev == 0 || ndev > 64) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_create: ndev must be 1..64");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(RMCLHIP_ERR_NO_DEVICE, "no HIP device available (librmclhip has no CPU fallback)");
  for (uint32_t i = 0; i < ndev; ++i) {
    const int d = devices ? devices[i] : static_cast<int>(i);
    if (d < 0 || d >= count) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_create: device index out of range");
  }
  BvhHost bvh;   // built ONCE, uploaded to every device
  const std::string err = build_bvh(v, nv, f, nf, bvh);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_create: " + err);
  rmclhip_rcc_sharded* h = new rmclhip_rcc_sharded();
  h->ranks.resize(ndev);
  for (uint32_t r = 0; r < ndev; ++r) {
    RccRank& R = h->ranks[r];
    rmclhip_status st = rmclhip_ctx_create(devices ? devices[r] : static_cast<int>(r), &R.ctx);
    if (st == RMCLHIP_OK) st = map_upload(R.ctx, bvh, &R.map);
    if (st == RMCLHIP_OK) st = rmclhip_rcc_create(R.ctx, R.map, &R.rcc);
    if (st != RMCLHIP_OK) {
      const std::string msg = g_err;
      rmclhip_rcc_sharded_destroy(h);
      return fail(st, "rcc_sharded_create: " + msg);
    }
  }
  *out = h;
  return RMCLHIP_OK;
}

uint32_t rmclhip_rcc_sharded_size(const rmclhip_rcc_sharded* h) { return h ? static_cast<uint32_t>(h->ranks.size()) : 0u; }

rmclhip_status rmclhip_rcc_sharded_replica(rmclhip_rcc_sharded* h, uint32_t rank, rmclhip_rcc** out) {
  if (!h || !out || rank >= h->ranks.size()) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_replica: bad arguments");
  *out = h->ranks[rank].rcc;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_rcc_sharded_correct_batch(rmclhip_rcc_sharded* h, const rmclhip_transform* Tbm, uint32_t nposes,
                                                 rmclhip_transform* Tdelta_out, rmclhip_cross_statistics* stats_out) {
  ApiGuard guard_("rmclhip_rcc_sharded_correct_batch");
  if (!h || !Tbm || !Tdelta_out) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_correct_batch: null");
  if (nposes == 0) return RMCLHIP_OK;
  const uint32_t world = static_cast<uint32_t>(h->ranks.size());
  // phase 1: every replica's chain is enqueued (nothing here waits for a device)
  for (uint32_t rk = 0; rk < world; ++rk) {
    RccRank& R = h->ranks[rk];
    rmclhip_rcc* r = R.rcc;
    uint32_t lo, hi;
    shard_bounds(nposes, rk, world, &lo, &hi);
    const uint32_t cnt = hi - lo;
    if (cnt == 0) continue;
    if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_correct_batch: a replica has no sensor model");
    if (cnt > 32768) return fail(RMCLHIP_ERR_UNSUPPORTED, "rcc_sharded_correct_batch: at most 32768 poses per device and call");
    if (r->n_dataset != static_cast<size_t>(r->W) * r->H) return fail(RMCLHIP_ERR_INVALID, "rcc_sharded_correct_batch: dataset size != model size");
    HIPCHK(hipSetDevice(R.ctx->device));
    if (cnt > R.cap) {
      if (R.h_Tdelta) { (void)hipHostFree(R.h_Tdelta); R.h_Tdelta = nullptr; }
      if (R.h_stats) { (void)hipHostFree(R.h_stats); R.h_stats = nullptr; }
      R.cap = 0;
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&R.h_Tdelta), sizeof(xform) * cnt, hipHostMallocDefault));
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&R.h_stats), sizeof(cstats) * cnt, hipHostMallocDefault));
      R.cap = cnt;
    }
    HIPCHK(r->d_Tdelta.reserve(cnt)); HIPCHK(r->d_bstats.reserve(cnt));
    if (rmclhip_status st = find_batch_enqueue(r, Tbm + lo, cnt)) return st;
    ReduceTail tail;
    tail.mode = kTailBatchSolve;
    tail.Tdelta_out = r->d_Tdelta.p;
    tail.stats_out = r->d_bstats.p;
    if (rmclhip_status st = reduce_enqueue(r, xidentity(), nullptr, r->max_dist, cnt, tail)) return st;
    HIPCHK(hipMemcpyAsync(R.h_Tdelta, r->d_Tdelta.p, sizeof(xform) * cnt, hipMemcpyDeviceToHost, r->stream));
    if (stats_out) HIPCHK(hipMemcpyAsync(R.h_stats, r->d_bstats.p, sizeof(cstats) * cnt, hipMemcpyDeviceToHost, r->stream));
  }
  // phase 2: wait for each and hand its block over
  for (uint32_t rk = 0; rk < world; ++rk) {
    RccRank& R = h->ranks[rk];
    uint32_t lo, hi;
    shard_bounds(nposes, rk, world, &lo, &hi);
    if (hi == lo) continue;
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(hipStreamSynchronize(R.rcc->stream));
    std::memcpy(Tdelta_out + lo, R.h_Tdelta, sizeof(xform) * (hi - lo));
    if (stats_out) std::memcpy(stats_out + lo, R.h_stats, sizeof(cstats) * (hi - lo));
  }
  return RMCLHIP_OK;
}

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
struct TuneCand { int kind; bool frontier; int reported; };
static const TuneCand kTuneSingle[5] = {{2, true, 2}, {23, true, 23}, {23, false, 19}, {24, true, 24}, {24, false, 22}};
static const TuneCand kTuneBatch[4] = {{23, true, 23}, {23, false, 19}, {24, true, 24}, {24, false, 22}};

rmclhip_status rmclhip_rcc_autotune(rmclhip_rcc* r, const rmclhip_transform* Tbm_est, int* chosen_kind, float* kernel_ms) {
  ApiGuard guard_("rmclhip_rcc_autotune");
  if (!r || !Tbm_est) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune: bad arguments");
  if (r->kind == kModelNone || r->W == 0 || r->H == 0) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune: no sensor model");
  if (r->variant != 15) return fail(RMCLHIP_ERR_INVALID, "rcc_autotune: a traversal kind is forced (set_variant); nothing to choose");
  HIPCHK(hipSetDevice(r->ctx->device));
  // each candidate timed on THIS map, model and pose: median of 5 batches of 8 back-to-back launches
  const int saved_kind = r->tuned_kind;
  const bool saved_frontier = r->tuned_frontier;
  const TuneCand* best = nullptr;
  float best_ms = 0.f;
  for (const TuneCand& c : kTuneSingle) {
    r->tuned_kind = c.kind; r->tuned_frontier = c.frontier;
    float t[5];
    for (float& x : t) {
      if (rmclhip_status st = rmclhip_rcc_time_find(r, Tbm_est, 8, &x)) { r->tuned_kind = saved_kind; r->tuned_frontier = saved_frontier; return st; }
    }
    std::sort(t, t + 5);
    if (!best || t[2] < best_ms) { best = &c; best_ms = t[2]; }
  }
  r->tuned_kind = best->kind; r->tuned_frontier = best->frontier;
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
    float t[3];
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
  // bit 13 adds 16 to the traversal kind (kinds 16..31)
  const int kind = (variant & 0xF) | (((variant >> 13) & 1) << 4), tile = (variant >> 4) & 0xF;
  if (kind == 3 || kind == 18 || kind > 30 || tile > 7 || (variant >> 14) != 0) return fail(RMCLHIP_ERR_INVALID, "rcc_set_variant: unknown variant");
  // (kind 1 also selects the one-lane-per-point form of the closest-point query, which the product owns)
  if (kind != 15 && kind != 1 && !find_kind_in_product(kind) && lab_hooks() == nullptr)
    return fail(RMCLHIP_ERR_UNSUPPORTED, "rcc_set_variant: this traversal kind is an experiment -- it lives in librmclhip_lab.so, "
                                         "which is not loaded (the product builds kinds 0, 2, 23, 24 and the automatic rule 15)");
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
  const size_t dwords = static_cast<size_t>(nblocks) * 4u * 8u;
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

static rmclhip_status find_batch_enqueue(rmclhip_rcc* r, const rmclhip_transform* Tbm, uint32_t nposes) {
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
  for (uint32_t i = 0; i < iters; ++i) HIPCHK(launch_find(p, r->kind, bvariant, r->stream));
  HIPCHK(hipEventRecord(r->ev1, r->stream));
  HIPCHK(hipStreamSynchronize(r->stream));
  float total = 0.f;
  HIPCHK(hipEventElapsedTime(&total, r->ev0, r->ev1));
  *ms = total / static_cast<float>(iters);
  return RMCLHIP_OK;
}

// ---- host-side algebra ---------------------------------------------------------------------------
rmclhip_status rmclhip_umeyama_transform(const rmclhip_cross_statistics* s, rmclhip_transform* out) {
  ApiGuard guard_("rmclhip_umeyama_transform");
  if (!s || !out) return fail(RMCLHIP_ERR_INVALID, "umeyama_transform: null");
  from_x(umeyama(to_cs(s)), out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_cross_statistics_merge(const rmclhip_cross_statistics* a, const rmclhip_cross_statistics* b,
                                              rmclhip_cross_statistics* out) {
  ApiGuard guard_("rmclhip_cross_statistics_merge");
  if (!a || !b || !out) return fail(RMCLHIP_ERR_INVALID, "cross_statistics_merge: null");
  from_cs(cs_merge(to_cs(a), to_cs(b)), out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_cross_statistics_transform(const rmclhip_transform* T, const rmclhip_cross_statistics* s,
                                                  rmclhip_cross_statistics* out) {
  ApiGuard guard_("rmclhip_cross_statistics_transform");
  if (!T || !s || !out) return fail(RMCLHIP_ERR_INVALID, "cross_statistics_transform: null");
  from_cs(cs_transform(to_x(T), to_cs(s)), out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_transform_mult(const rmclhip_transform* a, const rmclhip_transform* b, rmclhip_transform* out) {
  ApiGuard guard_("rmclhip_transform_mult");
  if (!a || !b || !out) return fail(RMCLHIP_ERR_INVALID, "transform_mult: null");
  from_x(xmul(to_x(a), to_x(b)), out);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_transform_inv(const rmclhip_transform* a, rmclhip_transform* out) {
  ApiGuard guard_("rmclhip_transform_inv");
  if (!a || !out) return fail(RMCLHIP_ERR_INVALID, "transform_inv: null");
  from_x(xinv(to_x(a)), out);
  return RMCLHIP_OK;
}

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
  if (f->stream) (void)hipStreamDestroy(f->stream);
  rmclhip_map_release(f->map);
  ctx_release(f->ctx);
  delete f;
}

rmclhip_status rmclhip_pf_set_params(rmclhip_pf* f, const rmclhip_pf_params* p) {
  ApiGuard guard_("rmclhip_pf_set_params");
  if (!f || !p) return fail(RMCLHIP_ERR_INVALID, "pf_set_params: null");
  if (!(p->dist_sigma > 0.f)) return fail(RMCLHIP_ERR_INVALID, "pf_set_params: dist_sigma must be > 0");
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
  if (f->h_beams_cap < nf) {
    HIPCHK(hipStreamSynchronize(f->stream));
    if (f->h_beams) (void)hipHostFree(f->h_beams);
    f->h_beams = nullptr;
    f->h_beams_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&f->h_beams), nf * sizeof(float), hipHostMallocDefault));
    f->h_beams_cap = nf;
  } else {
    HIPCHK(hipStreamSynchronize(f->stream));  // the staging buffer may still be in flight
  }
  std::memcpy(f->h_beams, beams, nf * sizeof(float));
  f->beams_at_origin = true;
  for (uint32_t b = 0; b < n_beams && f->beams_at_origin; ++b)
    f->beams_at_origin = beams[b].orig.x == 0.0f && beams[b].orig.y == 0.0f && beams[b].orig.z == 0.0f;
  HIPCHK(hipMemcpyAsync(f->d_beams.p, f->h_beams, nf * sizeof(float), hipMemcpyHostToDevice, f->stream));
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
  // small clouds: fewer particles per workgroup until the launch has ~4 workgroups per CU (1000 particles x 256 beams in 125 workgroups
  // left half of the chip idle), but never less than one ray per lane
  // (round 4: 1000 x 256: 0.104 -> 0.046 ms, 10 000 x 64: 0.127 -> 0.076 ms; clouds of >= 10 000 x 256 are unchanged)
  while (pb > 1u && n / pb < 1024u && static_cast<uint64_t>(pb >> 1) * n_beams >= 256u) pb >>= 1;
  p.particle_minor = 0u;
  p.order = nullptr;
  p.near_grid = nullptr;
  p.n_tris = f->map->info.n_faces;
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

// The resamplers as ENQUEUE + WAIT (round 4): the sharded entry points enqueue every device's part before they wait for any
// (pf_sharded_resample_impl); the public single-device calls are enqueue + one wait.  `st`: the stream the work goes to (the
// resampler's own, or the communicator's stream of that device behind the all-gather of the cloud).
static rmclhip_status gladiator_enqueue(rmclhip_resampler* r, const rmclhip_transform* poses_dev, const rmclhip_particle_attributes* attrs_dev,
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
static rmclhip_status residual_check(ResidualJob& j) {
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
static rmclhip_status residual_prepare_enqueue(ResidualJob& j) {
  if (!j.active) return RMCLHIP_OK;
  rmclhip_resampler* r = j.r;
  HIPCHK(hipSetDevice(r->ctx->device));
  HIPCHK(r->d_res_stats.reserve(4));
  // ResidualResamplerCPU.cpp:72-85
  HIPCHK(launch_residual_prepare(j.attrs, j.n_particles, j.n_new, r->d_psum.p, r->d_pmax.p, r->d_res_stats.p, j.st));
  HIPCHK(hipMemcpyAsync(r->h_res, r->d_res_stats.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, j.st));
  return RMCLHIP_OK;
}
static rmclhip_status residual_draws_enqueue(ResidualJob& j, bool first_try) {
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
static void residual_draws_done(ResidualJob& j) {   // after the wait that follows residual_draws_enqueue
  if (j.active && !j.filled) j.filled = j.r->h_res[4] >= j.n_new;
}
static rmclhip_status residual_fill_enqueue(ResidualJob& j, bool want_n_draws) {
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

// ---- multi-GPU: one process drives ndev devices (the reference's localisation node is one process,
// rmcl_localization.cpp:482-552); RCCL communicators of ncclCommInitAll, resolved with dlopen so that single-GPU users never
// load the library ---------------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static bool rccl_load(std::string& err) {
  static std::mutex mtx;   // two threads may create their first communicator at the same time
  std::lock_guard<std::mutex> lock(mtx);
  if (g_rccl.lib) return true;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) { err = std::string("dlopen(librccl.so.1): ") + dlerror(); return false; }
#define RCCL_SYM(field, name)                                                       \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));        \
  if (!g_rccl.field) { err = std::string("librccl lacks ") + name; dlclose(lib); return false; }
  RCCL_SYM(CommInitAll, "ncclCommInitAll") RCCL_SYM(CommDestroy, "ncclCommDestroy") RCCL_SYM(AllGather, "ncclAllGather")
  RCCL_SYM(AllReduce, "ncclAllReduce") RCCL_SYM(GroupStart, "ncclGroupStart") RCCL_SYM(GroupEnd, "ncclGroupEnd")
  RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
  g_rccl.lib = lib;
  return true;
}

#define NCCLCHK(expr)                                                                                          \
  do {                                                                                                         \
    const ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return fail(RMCLHIP_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

}  // extern "C" (the structs below are C++)

struct rmclhip_comm {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;   // one collective stream per device
  // rmclhip_comm_create_loopback: an in-process stand-in for RCCL (every "rank" is a stream of this process, ranks may share a
  // device): collectives are device-to-device copies / one small kernel, ordered with events.  It exists so that the ndev > 1 code
  // paths of the sharded entry points run -- and are checked against the unsharded results -- on a box with ONE GPU.
  bool loopback = false;
  std::vector<hipEvent_t> ev_in, ev_out;
};

// debug trace of the sharded entry points (rmclhip_debug_trace): "E<r>" = rank r's work of a phase enqueued, "W<r>" = the host waited
// for rank r.  A phase that scales reads E0 E1 ... W0 W1 ...; E0 W0 E1 W1 serialises the devices.
static std::atomic<bool> g_trace_on{false};
static std::mutex g_trace_mtx;
static std::string g_trace;
static inline void trace(char what, uint32_t rank) {
  if (!g_trace_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lock(g_trace_mtx);
  g_trace += what;
  g_trace += std::to_string(rank);
  g_trace += ' ';
}
static inline void trace_mark(const char* label) {
  if (!g_trace_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lock(g_trace_mtx);
  g_trace += label;
  g_trace += ' ';
}

// every rank's stream waits until all ranks' streams have reached this point (loopback collectives only)
static hipError_t loopback_barrier(rmclhip_comm* c, std::vector<hipEvent_t>& evs) {
  const size_t world = c->devices.size();
  for (size_t r = 0; r < world; ++r) {
    hipError_t e = hipSetDevice(c->devices[r]);
    if (e == hipSuccess) e = hipEventRecord(evs[r], c->streams[r]);
    if (e != hipSuccess) return e;
  }
  for (size_t r = 0; r < world; ++r) {
    hipError_t e = hipSetDevice(c->devices[r]);
    for (size_t s = 0; s < world && e == hipSuccess; ++s)
      if (s != r) e = hipStreamWaitEvent(c->streams[r], evs[s], 0);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

struct PfRank {
  rmclhip_ctx* ctx = nullptr;
  rmclhip_map* map = nullptr;
  rmclhip_pf* pf = nullptr;
  rmclhip_resampler* rs = nullptr;
  uint32_t lo = 0, hi = 0;
  xform* d_poses = nullptr; void* d_attrs = nullptr;          // this rank's shard (cap particles)
  xform* d_poses_new = nullptr; void* d_attrs_new = nullptr;  // tournament output
  xform* d_poses_all = nullptr; void* d_attrs_all = nullptr;  // gathered cloud (world * cap), distributed tournament
  void* d_poses_pad = nullptr; void* d_attrs_pad = nullptr;   // ragged partitions only: the padded all-gather lands here, *_all is its dense form
  uint32_t pad_cap = 0;                                        // capacity (records per rank) the pad buffers were sized for
  float* d_w_send = nullptr;   // cap
  float* d_w_pad = nullptr;    // world * cap (all-gather layout)
  float* d_w_all = nullptr;    // n_total, dense
  double* d_mom_part = nullptr;  // 256 * 32
  double* d_mom = nullptr;       // 32 (+ 32 reduced)
  double* h_mom = nullptr;       // pinned 32
};

struct rmclhip_pf_sharded {
  rmclhip_comm* comm = nullptr;
  uint32_t n_total = 0, cap = 0;
  std::vector<PfRank> ranks;
  rmclhip_pf_params params{2.0f, 100.0f, 100.0f, 0.0f, {0.05f, 80.0f}, 10000u, 0u};
};

extern "C" {

static void shard_bounds(uint32_t n, uint32_t rank, uint32_t world, uint32_t* lo, uint32_t* hi) {
  const uint32_t base = n / world, rem = n % world;
  *lo = rank * base + std::min(rank, rem);
  *hi = *lo + base + (rank < rem ? 1u : 0u);
}

void rmclhip_shard_bounds(uint32_t n, uint32_t rank, uint32_t world, uint32_t* lo, uint32_t* hi) {
  if (world == 0 || !lo || !hi) return;
  shard_bounds(n, rank, world, lo, hi);
}

rmclhip_status rmclhip_comm_create(const int* devices, uint32_t ndev, rmclhip_comm** out) {
  ApiGuard guard_("rmclhip_comm_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "comm_create: out is null");
  *out = nullptr;
  if (ndev == 0 || ndev > 64) return fail(RMCLHIP_ERR_INVALID, "comm_create: ndev must be 1..64");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(RMCLHIP_ERR_NO_DEVICE, "no HIP device available (librmclhip has no CPU fallback)");
  std::vector<int> devs(ndev);
  for (uint32_t i = 0; i < ndev; ++i) {
    devs[i] = devices ? devices[i] : static_cast<int>(i);
    if (devs[i] < 0 || devs[i] >= count) return fail(RMCLHIP_ERR_INVALID, "comm_create: device index out of range");
    for (uint32_t j = 0; j < i; ++j)
      if (devs[j] == devs[i]) return fail(RMCLHIP_ERR_INVALID, "comm_create: duplicate device");
  }
  std::string err;
  if (!rccl_load(err)) return fail(RMCLHIP_ERR_UNSUPPORTED, "comm_create: " + err);
  rmclhip_comm* c = new rmclhip_comm();
  c->devices = devs;
  c->comms.resize(ndev);
  const ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), static_cast<int>(ndev), devs.data());
  if (r != ncclSuccess) {
    delete c;
    return fail(RMCLHIP_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
  }
  c->streams.resize(ndev);
  for (uint32_t i = 0; i < ndev; ++i) {
    (void)hipSetDevice(devs[i]);
    if (hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking) != hipSuccess) {
      rmclhip_comm_destroy(c);
      return fail(RMCLHIP_ERR_HIP, "comm_create: hipStreamCreate failed");
    }
  }
  *out = c;
  return RMCLHIP_OK;
}

void rmclhip_comm_destroy(rmclhip_comm* c) {
  if (!c) return;
  for (size_t i = 0; i < c->devices.size(); ++i) {
    (void)hipSetDevice(c->devices[i]);
    if (i < c->streams.size() && c->streams[i]) { (void)hipStreamSynchronize(c->streams[i]); (void)hipStreamDestroy(c->streams[i]); }
    if (i < c->comms.size() && c->comms[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comms[i]);
    if (i < c->ev_in.size() && c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
    if (i < c->ev_out.size() && c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
  }
  delete c;
}

uint32_t rmclhip_comm_size(const rmclhip_comm* c) { return c ? static_cast<uint32_t>(c->devices.size()) : 0u; }

// The loopback communicator (see struct rmclhip_comm): same interface, no RCCL, ranks may share a device.
rmclhip_status rmclhip_comm_create_loopback(const int* devices, uint32_t ndev, rmclhip_comm** out) {
  ApiGuard guard_("rmclhip_comm_create_loopback");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "comm_create_loopback: out is null");
  *out = nullptr;
  if (ndev == 0 || ndev > 64) return fail(RMCLHIP_ERR_INVALID, "comm_create_loopback: ndev must be 1..64");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(RMCLHIP_ERR_NO_DEVICE, "no HIP device available (librmclhip has no CPU fallback)");
  rmclhip_comm* c = new rmclhip_comm();
  c->loopback = true;
  c->devices.resize(ndev);
  c->streams.assign(ndev, nullptr);
  c->ev_in.assign(ndev, nullptr);
  c->ev_out.assign(ndev, nullptr);
  for (uint32_t i = 0; i < ndev; ++i) {
    c->devices[i] = devices ? devices[i] : 0;
    if (c->devices[i] < 0 || c->devices[i] >= count) { rmclhip_comm_destroy(c); return fail(RMCLHIP_ERR_INVALID, "comm_create_loopback: device index out of range"); }
    hipError_t e = hipSetDevice(c->devices[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming);
    if (e != hipSuccess) { rmclhip_comm_destroy(c); return fail(RMCLHIP_ERR_HIP, std::string("comm_create_loopback: ") + hipGetErrorString(e)); }
  }
  *out = c;
  return RMCLHIP_OK;
}

// ---- the collectives the sharded entry points use, on every rank's collective stream (nothing here waits on the host) ----
// all-gather of `bytes` per rank: recv[r][s * bytes ..] = send[s][0 .. bytes) for every rank r and s
static rmclhip_status comm_allgather(rmclhip_comm* c, const void* const* send, void* const* recv, size_t bytes) {
  const uint32_t world = static_cast<uint32_t>(c->devices.size());
  if (bytes == 0) return RMCLHIP_OK;
  if (!c->loopback) {
    NCCLCHK(g_rccl.GroupStart());
    for (uint32_t r = 0; r < world; ++r) {
      (void)hipSetDevice(c->devices[r]);
      const ncclResult_t nr = g_rccl.AllGather(send[r], recv[r], bytes, ncclChar, c->comms[r], c->streams[r]);
      if (nr != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(RMCLHIP_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(nr)); }
    }
    NCCLCHK(g_rccl.GroupEnd());
    return RMCLHIP_OK;
  }
  HIPCHK(loopback_barrier(c, c->ev_in));      // every rank's send buffer is complete
  for (uint32_t r = 0; r < world; ++r) {
    HIPCHK(hipSetDevice(c->devices[r]));
    for (uint32_t sr = 0; sr < world; ++sr)
      HIPCHK(hipMemcpyAsync(static_cast<char*>(recv[r]) + static_cast<size_t>(sr) * bytes, send[sr], bytes, hipMemcpyDefault, c->streams[r]));
  }
  HIPCHK(loopback_barrier(c, c->ev_out));     // ... and nobody overwrites it before every rank has read it
  return RMCLHIP_OK;
}
// all-reduce of `count` doubles (sum or max): recv[r] = op over s of send[s], identical on every rank
static rmclhip_status comm_allreduce_f64(rmclhip_comm* c, const double* const* send, double* const* recv, uint32_t count, bool is_max) {
  const uint32_t world = static_cast<uint32_t>(c->devices.size());
  if (count == 0) return RMCLHIP_OK;
  if (!c->loopback) {
    NCCLCHK(g_rccl.GroupStart());
    for (uint32_t r = 0; r < world; ++r) {
      (void)hipSetDevice(c->devices[r]);
      const ncclResult_t nr = g_rccl.AllReduce(send[r], recv[r], count, ncclDouble, is_max ? ncclMax : ncclSum, c->comms[r], c->streams[r]);
      if (nr != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(RMCLHIP_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(nr)); }
    }
    NCCLCHK(g_rccl.GroupEnd());
    return RMCLHIP_OK;
  }
  HIPCHK(loopback_barrier(c, c->ev_in));
  for (uint32_t r = 0; r < world; ++r) {
    HIPCHK(hipSetDevice(c->devices[r]));
    HIPCHK(launch_loopback_allreduce(send, world, recv[r], count, is_max, c->streams[r]));
  }
  HIPCHK(loopback_barrier(c, c->ev_out));
  return RMCLHIP_OK;
}
// the host waits for every rank's collective stream (ONE wait per rank, after everything of a phase has been enqueued)
static rmclhip_status comm_wait_all(rmclhip_comm* c) {
  for (size_t r = 0; r < c->devices.size(); ++r) {
    HIPCHK(hipSetDevice(c->devices[r]));
    HIPCHK(hipStreamSynchronize(c->streams[r]));
    trace('W', static_cast<uint32_t>(r));
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_debug_tag_retries(unsigned long long* retries_out) {
  if (!retries_out) return fail(RMCLHIP_ERR_INVALID, "debug_tag_retries: null");
  *retries_out = g_tag_sum_retries.load();
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_debug_trace(int on, char* buf, size_t cap) {
  // on = 1: start (clears), on = 0: stop; buf (nullable) receives what was recorded so far
  std::lock_guard<std::mutex> lock(g_trace_mtx);
  if (buf && cap) {
    const size_t n = std::min(cap - 1, g_trace.size());
    std::memcpy(buf, g_trace.data(), n);
    buf[n] = 0;
  }
  if (on) g_trace.clear();
  g_trace_on = on != 0;
  return RMCLHIP_OK;
}

void rmclhip_pf_sharded_destroy(rmclhip_pf_sharded* h) {
  if (!h) return;
  for (PfRank& R : h->ranks) {
    if (R.ctx) (void)hipSetDevice(R.ctx->device);
    void* bufs[] = {R.d_poses, R.d_attrs, R.d_poses_new, R.d_attrs_new, R.d_poses_all, R.d_attrs_all, R.d_w_send, R.d_w_pad,
                    R.d_w_all, R.d_mom_part, R.d_mom, R.d_poses_pad, R.d_attrs_pad};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (R.h_mom) (void)hipHostFree(R.h_mom);
    if (R.rs) rmclhip_resampler_destroy(R.rs);
    if (R.pf) rmclhip_pf_destroy(R.pf);
    if (R.map) rmclhip_map_release(R.map);
    if (R.ctx) rmclhip_ctx_destroy(R.ctx);
  }
  delete h;
}

rmclhip_status rmclhip_pf_sharded_create(rmclhip_comm* comm, const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                                         rmclhip_pf_sharded** out) {
  ApiGuard guard_("rmclhip_pf_sharded_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_create: out is null");
  *out = nullptr;
  if (!comm) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_create: null communicator");
  BvhHost bvh;   // built ONCE, uploaded to every device (mesh + BVH replicated)
  const std::string err = build_bvh(v, nv, f, nf, bvh);
  if (!err.empty()) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_create: " + err);
  rmclhip_pf_sharded* h = new rmclhip_pf_sharded();
  h->comm = comm;
  h->ranks.resize(comm->devices.size());
  for (size_t r = 0; r < h->ranks.size(); ++r) {
    PfRank& R = h->ranks[r];
    rmclhip_status st = rmclhip_ctx_create(comm->devices[r], &R.ctx);
    if (st == RMCLHIP_OK) st = map_upload(R.ctx, bvh, &R.map);
    if (st == RMCLHIP_OK) st = rmclhip_pf_create(R.ctx, R.map, &R.pf);
    if (st == RMCLHIP_OK) st = rmclhip_resampler_create(R.ctx, &R.rs);
    hipError_t e = hipSuccess;
    if (st == RMCLHIP_OK) e = hipMalloc(reinterpret_cast<void**>(&R.d_mom_part), 256 * 32 * sizeof(double));
    if (st == RMCLHIP_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&R.d_mom), 64 * sizeof(double));
    if (st == RMCLHIP_OK && e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&R.h_mom), 64 * sizeof(double), hipHostMallocDefault);
    if (st != RMCLHIP_OK || e != hipSuccess) {
      const std::string msg = (st != RMCLHIP_OK) ? g_err : std::string(hipGetErrorString(e));
      rmclhip_pf_sharded_destroy(h);
      return fail(st != RMCLHIP_OK ? st : RMCLHIP_ERR_HIP, "pf_sharded_create: " + msg);
    }
  }
  *out = h;
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_sharded_set_params(rmclhip_pf_sharded* h, const rmclhip_pf_params* p) {
  if (!h || !p) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_set_params: null");
  for (PfRank& R : h->ranks)
    if (rmclhip_status st = rmclhip_pf_set_params(R.pf, p)) return st;
  h->params = *p;
  return RMCLHIP_OK;
}

// contiguous block partition of the particle range; shards are padded to `cap` so that the collectives
// run on equal counts
rmclhip_status rmclhip_pf_sharded_set_particles(rmclhip_pf_sharded* h, const rmclhip_transform* poses,
                                                const rmclhip_particle_attributes* attrs, uint32_t n_total) {
  ApiGuard guard_("rmclhip_pf_sharded_set_particles");
  if (!h || (n_total && (!poses || !attrs))) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_set_particles: null");
  const uint32_t world = static_cast<uint32_t>(h->ranks.size());
  const uint32_t cap = (n_total + world - 1u) / world;
  for (uint32_t r = 0; r < world; ++r) {
    PfRank& R = h->ranks[r];
    HIPCHK(hipSetDevice(R.ctx->device));
    if (cap > h->cap || !R.d_poses) {
      void** bufs[] = {reinterpret_cast<void**>(&R.d_poses), &R.d_attrs, reinterpret_cast<void**>(&R.d_poses_new), &R.d_attrs_new,
                       reinterpret_cast<void**>(&R.d_poses_all), &R.d_attrs_all, reinterpret_cast<void**>(&R.d_w_send),
                       reinterpret_cast<void**>(&R.d_w_pad), reinterpret_cast<void**>(&R.d_w_all)};
      for (void** b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
      const size_t c = std::max<uint32_t>(cap, 1u);
      // d_w_all holds the dense [n_total] weights: n_total <= cap * world for every cloud this capacity admits (sizing it
      // by the n_total of the call that allocated let a later, larger cloud of the same capacity write past its end)
      const size_t sizes[] = {c * 32, c * 36, c * 32, c * 36, c * world * 32, c * world * 36, c * 4, c * world * 4, c * world * 4};
      hipError_t ae = hipSuccess;
      for (size_t k = 0; k < sizeof(sizes) / sizeof(sizes[0]) && ae == hipSuccess; ++k) ae = hipMalloc(bufs[k], sizes[k]);
      if (ae != hipSuccess) {
        // leave no half-allocated rank behind: the next call must see "no buffers" and start over
        for (void** b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
        h->cap = 0;
        return fail(ae == hipErrorOutOfMemory ? RMCLHIP_ERR_NOMEM : RMCLHIP_ERR_HIP, std::string("pf_sharded_set_particles: ") + hipGetErrorString(ae));
      }
    }
    shard_bounds(n_total, r, world, &R.lo, &R.hi);
    HIPCHK(hipMemset(R.d_poses, 0, static_cast<size_t>(std::max(cap, 1u)) * 32));
    HIPCHK(hipMemset(R.d_attrs, 0, static_cast<size_t>(std::max(cap, 1u)) * 36));
    HIPCHK(hipMemset(R.d_w_send, 0, static_cast<size_t>(std::max(cap, 1u)) * 4));
    if (R.hi > R.lo) {
      HIPCHK(hipMemcpy(R.d_poses, poses + R.lo, static_cast<size_t>(R.hi - R.lo) * 32, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(R.d_attrs, attrs + R.lo, static_cast<size_t>(R.hi - R.lo) * 36, hipMemcpyHostToDevice));
      HIPCHK(hipDeviceSynchronize());   // consumers run on non-blocking streams (see upload_on)
    }
  }
  h->n_total = n_total;
  h->cap = std::max(cap, h->cap);
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_sharded_download(rmclhip_pf_sharded* h, rmclhip_transform* poses, rmclhip_particle_attributes* attrs) {
  ApiGuard guard_("rmclhip_pf_sharded_download");
  if (!h) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_download: null");
  for (PfRank& R : h->ranks) {
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(hipDeviceSynchronize());
    if (R.hi > R.lo) {
      if (poses) HIPCHK(hipMemcpy(poses + R.lo, R.d_poses, static_cast<size_t>(R.hi - R.lo) * 32, hipMemcpyDeviceToHost));
      if (attrs) HIPCHK(hipMemcpy(attrs + R.lo, R.d_attrs, static_cast<size_t>(R.hi - R.lo) * 36, hipMemcpyDeviceToHost));
    }
  }
  return RMCLHIP_OK;
}

// all-gather of likelihood.mean (4 B x N; C5: 4 MB, one collective, never bucketed): afterwards EVERY device holds the dense
// weight vector of the whole cloud (consumer: the tournament / {sum, max}, resampling.cu:108-199)
rmclhip_status rmclhip_pf_allgather_weights(rmclhip_pf_sharded* h) {
  ApiGuard guard_("rmclhip_pf_allgather_weights");
  if (!h) return fail(RMCLHIP_ERR_INVALID, "pf_allgather_weights: null");
  if (h->n_total == 0) return RMCLHIP_OK;
  const uint32_t world = static_cast<uint32_t>(h->ranks.size()), cap = (h->n_total + world - 1u) / world;
  // extract, gather, compact: three steps per rank on its collective stream, all enqueued before the ONE wait per rank
  std::vector<const void*> send(world);
  std::vector<void*> recv(world);
  for (uint32_t r = 0; r < world; ++r) {
    PfRank& R = h->ranks[r];
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(launch_pf_extract_weights(R.d_attrs, R.hi - R.lo, R.d_w_send, h->comm->streams[r]));
    send[r] = R.d_w_send; recv[r] = R.d_w_pad;
  }
  if (rmclhip_status st = comm_allgather(h->comm, send.data(), recv.data(), static_cast<size_t>(cap) * sizeof(float))) return st;
  for (uint32_t r = 0; r < world; ++r) {
    PfRank& R = h->ranks[r];
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(launch_compact_shards(R.d_w_pad, R.d_w_all, h->n_total, world, cap, h->comm->streams[r]));
    trace('E', r);
  }
  return comm_wait_all(h->comm);
}

// PCDSensorUpdater*::update on every device's block of the particles (concurrently: one stream per device), then the
// weight all-gather
rmclhip_status rmclhip_pf_update_sharded(rmclhip_pf_sharded* h, const rmclhip_range_measurement* beams, uint32_t n_beams,
                                         const rmclhip_transform* Tsb) {
  ApiGuard guard_("rmclhip_pf_update_sharded");
  if (!h || !Tsb || (n_beams && !beams)) return fail(RMCLHIP_ERR_INVALID, "pf_update_sharded: null");
  trace_mark("update:");
  for (size_t r = 0; r < h->ranks.size(); ++r) {
    PfRank& R = h->ranks[r];
    if (R.hi == R.lo) continue;
    if (rmclhip_status st = rmclhip_pf_update_async(R.pf, reinterpret_cast<const rmclhip_transform*>(R.d_poses),
                                                    static_cast<rmclhip_particle_attributes*>(R.d_attrs), R.hi - R.lo, beams, n_beams, Tsb))
      return st;
    // the gather runs on the communicator's stream of this device: it waits for the update on the DEVICE (an event), not on the host
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(hipEventRecord(R.pf->ev1, R.pf->stream));
    HIPCHK(hipStreamWaitEvent(h->comm->streams[r], R.pf->ev1, 0));
    trace('E', static_cast<uint32_t>(r));
  }
  trace_mark("gather:");
  return rmclhip_pf_allgather_weights(h);
}

static rmclhip_status pf_sharded_resample_impl(rmclhip_pf_sharded* h, const rmclhip_gladiator_config* cfg, uint64_t seed, uint32_t step,
                                              bool residual);
rmclhip_status rmclhip_pf_allreduce_stats(rmclhip_pf_sharded* h, rmclhip_likelihood_stats* out);
// MotionUpdater<MemT>::update on every device's block (particle_motion.cu:11-46 + the collision ray of TFMotionUpdaterCPU.cpp:17-50,
// 207-221): one k_pf_motion per rank on that rank's update stream.  `wait`: false leaves the launches in flight -- the sensor update of
// the same cycle is enqueued behind them on the same streams (rmclhip_pf_sharded_step).
static rmclhip_status pf_sharded_motion_enqueue(rmclhip_pf_sharded* h, const rmclhip_transform* T_bnew_bold, double forget_rate, int check_collision) {
  for (size_t r = 0; r < h->ranks.size(); ++r) {
    PfRank& R = h->ranks[r];
    if (R.hi == R.lo) continue;
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(launch_pf_motion(R.map->d_qnodes, R.map->d_tris, R.d_poses, static_cast<rmclhip_particle_attributes*>(R.d_attrs), R.hi - R.lo,
                            to_x(T_bnew_bold), forget_rate, h->params.max_n_meas, check_collision != 0, R.pf->stream));
    trace('E', static_cast<uint32_t>(r));
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_sharded_motion_update(rmclhip_pf_sharded* h, const rmclhip_transform* T_bnew_bold, double forget_rate,
                                                int check_collision) {
  ApiGuard guard_("rmclhip_pf_sharded_motion_update");
  if (!h || !T_bnew_bold) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_motion_update: null");
  if (h->n_total == 0) return RMCLHIP_OK;
  trace_mark("motion:");
  if (rmclhip_status st = pf_sharded_motion_enqueue(h, T_bnew_bold, forget_rate, check_collision)) return st;
  for (size_t r = 0; r < h->ranks.size(); ++r) {   // every rank's launch is in flight before the host waits for any
    PfRank& R = h->ranks[r];
    if (R.hi == R.lo) continue;
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(R.pf->tag.wait_chain_end(R.ctx, R.pf->stream));
    trace('W', static_cast<uint32_t>(r));
  }
  return RMCLHIP_OK;
}

// One cycle of the filter node (rmcl_localization.cpp:84, 432-552: motionUpdate, sensorUpdate, resampling with its {sum, max}) on the
// sharded cloud: motion (nullable T_bnew_bold: skipped) -> sensor update -> weight all-gather -> {sum, max} all-reduce -> resampling
// (resample: 0 none, 1 gladiator tournament, 2 residual).  Motion and sensor update of a rank share its stream, so the host waits
// once for the gather, once for the statistics and once inside the resampler -- never between motion and update.
rmclhip_status rmclhip_pf_sharded_step(rmclhip_pf_sharded* h, const rmclhip_transform* T_bnew_bold, double forget_rate, int check_collision,
                                       const rmclhip_range_measurement* beams, uint32_t n_beams, const rmclhip_transform* Tsb,
                                       int resample, const rmclhip_gladiator_config* cfg, uint64_t seed, uint32_t step,
                                       rmclhip_likelihood_stats* stats_out) {
  ApiGuard guard_("rmclhip_pf_sharded_step");
  if (!h || !Tsb || (n_beams && !beams)) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_step: null");
  if (resample < 0 || resample > 2 || (resample != 0 && !cfg)) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_step: bad resampling arguments");
  if (h->n_total == 0) { if (stats_out) { stats_out->sum = 0.f; stats_out->max = 0.f; } return RMCLHIP_OK; }
  if (T_bnew_bold) {
    trace_mark("motion:");
    if (rmclhip_status st = pf_sharded_motion_enqueue(h, T_bnew_bold, forget_rate, check_collision)) return st;
  }
  if (rmclhip_status st = rmclhip_pf_update_sharded(h, beams, n_beams, Tsb)) return st;
  rmclhip_likelihood_stats st_local;
  trace_mark("stats:");
  if (rmclhip_status st = rmclhip_pf_allreduce_stats(h, stats_out ? stats_out : &st_local)) return st;
  if (resample != 0) {
    if (rmclhip_status st = pf_sharded_resample_impl(h, cfg, seed, step, resample == 2)) return st;
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_pf_sharded_get_weights(rmclhip_pf_sharded* h, uint32_t rank, float* weights_host) {
  ApiGuard guard_("rmclhip_pf_sharded_get_weights");
  if (!h || !weights_host || rank >= h->ranks.size()) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_get_weights: bad arguments");
  PfRank& R = h->ranks[rank];
  HIPCHK(hipSetDevice(R.ctx->device));
  if (h->n_total) HIPCHK(hipMemcpy(weights_host, R.d_w_all, static_cast<size_t>(h->n_total) * 4, hipMemcpyDeviceToHost));
  return RMCLHIP_OK;
}

// moments of one pass on every rank's overlap with [0, n_use), all-reduced (sums with ncclSum, maxima with ncclMax);
// result (identical on every rank) in out32
static rmclhip_status sharded_moments(rmclhip_pf_sharded* h, uint32_t n_use, int pass, double L_sum, const xform& Tbm, double* out32) {
  const uint32_t world = static_cast<uint32_t>(h->ranks.size());
  for (uint32_t r = 0; r < world; ++r) {
    PfRank& R = h->ranks[r];
    HIPCHK(hipSetDevice(R.ctx->device));
    const uint32_t hi = std::min(R.hi, n_use);
    const uint32_t n = (hi > R.lo) ? (hi - R.lo) : 0u;
    HIPCHK(launch_pose_moments(R.d_poses, R.d_attrs, n, pass, L_sum, Tbm, R.d_mom_part, R.d_mom, h->comm->streams[r]));
  }
  std::vector<const double*> s_sum(world), s_max(world);
  std::vector<double*> r_sum(world), r_max(world);
  for (uint32_t r = 0; r < world; ++r) {
    PfRank& R = h->ranks[r];
    s_sum[r] = R.d_mom; r_sum[r] = R.d_mom + 32; s_max[r] = R.d_mom + 24; r_max[r] = R.d_mom + 32 + 24;
  }
  if (rmclhip_status st = comm_allreduce_f64(h->comm, s_sum.data(), r_sum.data(), 24, false)) return st;
  if (rmclhip_status st = comm_allreduce_f64(h->comm, s_max.data(), r_max.data(), 8, true)) return st;
  PfRank& R0 = h->ranks[0];
  HIPCHK(hipSetDevice(R0.ctx->device));
  HIPCHK(hipMemcpyAsync(R0.h_mom, R0.d_mom + 32, 32 * sizeof(double), hipMemcpyDeviceToHost, h->comm->streams[0]));
  if (rmclhip_status st = comm_wait_all(h->comm)) return st;
  std::memcpy(out32, R0.h_mom, 32 * sizeof(double));
  return RMCLHIP_OK;
}

// global {sum, max} of the likelihoods: the distributed form of simple_stats_kernel (resampling.cu:41-92)
rmclhip_status rmclhip_pf_allreduce_stats(rmclhip_pf_sharded* h, rmclhip_likelihood_stats* out) {
  ApiGuard guard_("rmclhip_pf_allreduce_stats");
  if (!h || !out) return fail(RMCLHIP_ERR_INVALID, "pf_allreduce_stats: null");
  double m[32];
  if (rmclhip_status st = sharded_moments(h, h->n_total, 0, 1.0, xidentity(), m)) return st;
  out->sum = static_cast<float>(m[0]);
  out->max = static_cast<float>(std::max(m[24], 0.0));   // seeded with 0 like the reference's shared-memory init
  return RMCLHIP_OK;
}

// largest eigenvector of a symmetric 4x4 matrix (cyclic Jacobi, double)
static void sym4_largest_eigenvector(const double* M10, double* q) {
  double A[4][4], V[4][4];
  int k = 0;
  for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b) { A[a][b] = A[b][a] = M10[k++]; }
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) V[a][b] = (a == b) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0;
    for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) off += A[a][b] * A[a][b];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int qq = p + 1; qq < 4; ++qq) {
        if (A[p][qq] == 0.0) continue;
        const double theta = (A[qq][qq] - A[p][p]) / (2.0 * A[p][qq]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int i = 0; i < 4; ++i) { const double ip = A[i][p], iq = A[i][qq]; A[i][p] = c * ip - sn * iq; A[i][qq] = sn * ip + c * iq; }
        for (int i = 0; i < 4; ++i) { const double pi_ = A[p][i], qi = A[qq][i]; A[p][i] = c * pi_ - sn * qi; A[qq][i] = sn * pi_ + c * qi; }
        for (int i = 0; i < 4; ++i) { const double ip = V[i][p], iq = V[i][qq]; V[i][p] = c * ip - sn * iq; V[i][qq] = sn * ip + c * iq; }
      }
  }
  int best = 0;
  for (int a = 1; a < 4; ++a) if (A[a][a] > A[best][best]) best = a;
  double n = 0.0;
  for (int a = 0; a < 4; ++a) n += V[a][best] * V[a][best];
  n = std::sqrt(n);
  const double sgn = (V[3][best] < 0.0) ? -1.0 : 1.0;   // canonical sign: w >= 0
  for (int a = 0; a < 4; ++a) q[a] = sgn * V[a][best] / n;
}

// RmclNode::estimateStats (rmcl_localization.cpp:642-731) over the first n_induction particles of the sharded cloud
rmclhip_status rmclhip_pf_allreduce_pose_estimate(rmclhip_pf_sharded* h, uint32_t n_induction, rmclhip_pose_estimate* out) {
  ApiGuard guard_("rmclhip_pf_allreduce_pose_estimate");
  if (!h || !out) return fail(RMCLHIP_ERR_INVALID, "pf_allreduce_pose_estimate: null");
  std::memset(out, 0, sizeof(*out));
  const uint32_t n_use = std::min(n_induction, h->n_total);
  if (n_use == 0) return fail(RMCLHIP_ERR_INVALID, "pf_allreduce_pose_estimate: no particles");
  double m[32];
  if (rmclhip_status st = sharded_moments(h, n_use, 0, 1.0, xidentity(), m)) return st;
  const double L_sum = m[0], L_n = m[2];
  const double L_mean = L_sum / L_n;
  out->n_particles = n_use;
  out->likelihood_mean = L_mean;
  out->likelihood_sigma = std::sqrt(std::max(m[1] / L_n - L_mean * L_mean, 0.0));
  out->likelihood_max = std::max(m[24], 0.0);   // L_max starts at 0.0 in the reference (:665)
  out->likelihood_min = -m[25];
  for (int k = 0; k < 3; ++k) { out->trans_bb_max[k] = static_cast<float>(m[26 + k]); out->trans_bb_min[k] = static_cast<float>(-m[29 + k]); }
  // first pass: mean (rm::markley_mean with weights L_i / L_sum)
  if (rmclhip_status st = sharded_moments(h, n_use, 1, L_sum, xidentity(), m)) return st;
  double q[4];
  sym4_largest_eigenvector(m, q);
  xform Tbm = xidentity();
  Tbm.R.x = static_cast<float>(q[0]); Tbm.R.y = static_cast<float>(q[1]); Tbm.R.z = static_cast<float>(q[2]); Tbm.R.w = static_cast<float>(q[3]);
  Tbm.t = mk3(static_cast<float>(m[10]), static_cast<float>(m[11]), static_cast<float>(m[12]));
  from_x(Tbm, &out->pose);
  // second pass: covariance around the mean
  if (rmclhip_status st = sharded_moments(h, n_use, 2, L_sum, Tbm, m)) return st;
  int k = 0;
  for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { out->covariance[6 * a + b] = out->covariance[6 * b + a] = m[k++]; }
  return RMCLHIP_OK;
}

// distributed gladiator tournament: the enemy of a champion may live on any rank, so the cloud (68 B per
// particle) is all-gathered once, then every rank resamples its own champions against the gathered copy; the Philox
// stream is a function of the GLOBAL champion index, so the result equals the single-GPU tournament

rmclhip_status rmclhip_pf_sharded_resample(rmclhip_pf_sharded* h, const rmclhip_gladiator_config* cfg, uint64_t seed, uint32_t step) {
  ApiGuard guard_("rmclhip_pf_sharded_resample");
  return pf_sharded_resample_impl(h, cfg, seed, step, false);
}

rmclhip_status rmclhip_pf_sharded_resample_residual(rmclhip_pf_sharded* h, const rmclhip_gladiator_config* cfg, uint64_t seed, uint32_t step) {
  ApiGuard guard_("rmclhip_pf_sharded_resample_residual");
  return pf_sharded_resample_impl(h, cfg, seed, step, true);
}

static rmclhip_status pf_sharded_resample_impl(rmclhip_pf_sharded* h, const rmclhip_gladiator_config* cfg, uint64_t seed, uint32_t step,
                                              bool residual) {
  if (!h || !cfg) return fail(RMCLHIP_ERR_INVALID, "pf_sharded_resample: null");
  if (h->n_total == 0) return RMCLHIP_OK;
  const uint32_t world = static_cast<uint32_t>(h->ranks.size()), cap = (h->n_total + world - 1u) / world;
  // a ragged partition (n_total not a multiple of the number of devices): the all-gather needs equal counts, so the padded shards
  // land in a second buffer and one kernel per record type squeezes the padding out (68 B x N read + written once more per rank)
  const bool ragged = (h->n_total % world) != 0u && world > 1u;
  if (ragged) {
    for (PfRank& R : h->ranks) {
      if (R.d_poses_pad && R.pad_cap >= cap) continue;
      HIPCHK(hipSetDevice(R.ctx->device));
      if (R.d_poses_pad) { (void)hipFree(R.d_poses_pad); R.d_poses_pad = nullptr; }
      if (R.d_attrs_pad) { (void)hipFree(R.d_attrs_pad); R.d_attrs_pad = nullptr; }
      R.pad_cap = 0;
      const size_t c = static_cast<size_t>(std::max(cap, h->cap)) * world;
      hipError_t ae = hipMalloc(&R.d_poses_pad, c * 32);
      if (ae == hipSuccess) ae = hipMalloc(&R.d_attrs_pad, c * 36);
      if (ae != hipSuccess) {
        if (R.d_poses_pad) { (void)hipFree(R.d_poses_pad); R.d_poses_pad = nullptr; }
        return fail(ae == hipErrorOutOfMemory ? RMCLHIP_ERR_NOMEM : RMCLHIP_ERR_HIP, std::string("pf_sharded_resample: ") + hipGetErrorString(ae));
      }
      R.pad_cap = std::max(cap, h->cap);
    }
  }
  // (1) the cloud (68 B per particle) is all-gathered on every rank's collective stream; (2) BEHIND it, on the same stream, every
  // rank's tournament / slot fill over its own champions -- enqueued for ALL ranks before the host waits for any (round 3 ran the N
  // tournaments one after the other, each with its own launch + wait).  The tournament reads one enemy per champion; gathering the
  // whole cloud instead of an indexed exchange of the winners costs 68 B x N x (world - 1) / world per rank: C5 = 59.5 MB per rank,
  // ~0.2 ms at the ~300 GB/s an 8-GPU RCCL all-gather reaches over xGMI (an estimate: no node to measure on) against a 2.9 ms
  // sensor update per resampling step -- accepted, stated, and the first thing to replace if a profile says otherwise.
  trace_mark("resample:");
  std::vector<const void*> sp(world), sa(world);
  std::vector<void*> rp(world), ra_(world);
  for (uint32_t r = 0; r < world; ++r) {
    PfRank& R = h->ranks[r];
    sp[r] = R.d_poses; rp[r] = ragged ? R.d_poses_pad : static_cast<void*>(R.d_poses_all);
    sa[r] = R.d_attrs; ra_[r] = ragged ? R.d_attrs_pad : R.d_attrs_all;
  }
  if (rmclhip_status st = comm_allgather(h->comm, sp.data(), rp.data(), static_cast<size_t>(cap) * 32)) return st;
  if (rmclhip_status st = comm_allgather(h->comm, sa.data(), ra_.data(), static_cast<size_t>(cap) * 36)) return st;
  if (ragged) {
    for (uint32_t r = 0; r < world; ++r) {
      PfRank& R = h->ranks[r];
      HIPCHK(hipSetDevice(R.ctx->device));
      HIPCHK(launch_compact_records(R.d_poses_pad, R.d_poses_all, h->n_total, world, cap, 32u, h->comm->streams[r]));
      HIPCHK(launch_compact_records(R.d_attrs_pad, R.d_attrs_all, h->n_total, world, cap, 36u, h->comm->streams[r]));
    }
  }
  if (!residual) {
    for (uint32_t r = 0; r < world; ++r) {
      PfRank& R = h->ranks[r];
      if (R.hi == R.lo) continue;
      // every device resamples ITS champions against the whole gathered cloud: the random stream is a function of the global index
      if (rmclhip_status st = gladiator_enqueue(R.rs, reinterpret_cast<const rmclhip_transform*>(R.d_poses_all),
                                                static_cast<const rmclhip_particle_attributes*>(R.d_attrs_all), h->n_total,
                                                reinterpret_cast<rmclhip_transform*>(R.d_poses_new),
                                                static_cast<rmclhip_particle_attributes*>(R.d_attrs_new), R.lo, R.hi - R.lo, cfg, seed, step,
                                                h->comm->streams[r]))
        return st;
      trace('E', r);
    }
    if (rmclhip_status st = comm_wait_all(h->comm)) return st;
  } else {
    std::vector<ResidualJob> jobs(world);
    for (uint32_t r = 0; r < world; ++r) {
      PfRank& R = h->ranks[r];
      ResidualJob& j = jobs[r];
      j.r = R.rs; j.st = h->comm->streams[r];
      j.poses = reinterpret_cast<const rmclhip_transform*>(R.d_poses_all); j.attrs = static_cast<const rmclhip_particle_attributes*>(R.d_attrs_all);
      j.poses_new = reinterpret_cast<rmclhip_transform*>(R.d_poses_new); j.attrs_new = static_cast<rmclhip_particle_attributes*>(R.d_attrs_new);
      j.n_particles = h->n_total; j.n_new = h->n_total; j.first = R.lo; j.count = R.hi - R.lo; j.cfg = cfg; j.seed = seed; j.step = step;
      if (rmclhip_status st = residual_check(j)) return st;
    }
    // three phases, each enqueued on every rank before the one wait per rank (round 3: three waits per rank, rank after rank)
    for (uint32_t r = 0; r < world; ++r) { if (rmclhip_status st = residual_prepare_enqueue(jobs[r])) return st; trace('E', r); }
    if (rmclhip_status st = comm_wait_all(h->comm)) return st;
    for (bool first_try = true;; first_try = false) {
      bool any = false;
      for (uint32_t r = 0; r < world; ++r) {
        if (!jobs[r].active || jobs[r].filled) continue;
        any = true;
        if (rmclhip_status st = residual_draws_enqueue(jobs[r], first_try)) return st;
        trace('E', r);
      }
      if (!any) break;
      if (rmclhip_status st = comm_wait_all(h->comm)) return st;
      for (uint32_t r = 0; r < world; ++r) residual_draws_done(jobs[r]);
    }
    for (uint32_t r = 0; r < world; ++r) { if (rmclhip_status st = residual_fill_enqueue(jobs[r], false)) return st; trace('E', r); }
    if (rmclhip_status st = comm_wait_all(h->comm)) return st;
  }
  for (PfRank& R : h->ranks) {
    if (R.hi == R.lo) continue;
    std::swap(R.d_poses, R.d_poses_new);
    std::swap(R.d_attrs, R.d_attrs_new);
  }
  return RMCLHIP_OK;
}

// ---- device memory helpers ---------------------------------------------------------------------------
rmclhip_status rmclhip_malloc(rmclhip_ctx* ctx, size_t bytes, void** out) {
  ApiGuard guard_("rmclhip_malloc");
  if (!ctx || !out) return fail(RMCLHIP_ERR_INVALID, "malloc: null");
  HIPCHK(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(out, bytes ? bytes : 1);
  if (e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? RMCLHIP_ERR_NOMEM : RMCLHIP_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_free(rmclhip_ctx* ctx, void* p) {
  ApiGuard guard_("rmclhip_free");
  if (!ctx) return fail(RMCLHIP_ERR_INVALID, "free: null ctx");
  if (!p) return RMCLHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipFree(p));
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_memcpy_h2d(rmclhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  ApiGuard guard_("rmclhip_memcpy_h2d");
  if (!ctx || (bytes && (!dst || !src))) return fail(RMCLHIP_ERR_INVALID, "memcpy_h2d: null");
  HIPCHK(hipSetDevice(ctx->device));
  if (bytes) {
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipDeviceSynchronize());   // consumers run on non-blocking streams (see upload_on)
  }
  return RMCLHIP_OK;
}

rmclhip_status rmclhip_memcpy_d2h(rmclhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  ApiGuard guard_("rmclhip_memcpy_d2h");
  if (!ctx || (bytes && (!dst || !src))) return fail(RMCLHIP_ERR_INVALID, "memcpy_d2h: null");
  HIPCHK(hipSetDevice(ctx->device));
  if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return RMCLHIP_OK;
}

}  // extern "C"
