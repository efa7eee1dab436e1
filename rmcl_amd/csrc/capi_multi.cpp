// capi_multi.cpp -- see capi_internal.h
#include "capi_internal.h"

// ---- pose batches sharded over devices (north_star: pose-corrections/s at 1/2/4/8 GPUs; SURVEY 8(e): "MICP pose batches: shard
// poses, no exchange at all").  One process, one operator replica per device over ONE host BVH build (as rmclhip_pf_sharded_create
// does for the filter); the poses of a batch are block-partitioned with shard_bounds, every replica's chain (pose upload, find over
// its block, reduction, per-pose solve, results to pinned host memory) is ENQUEUED before any is waited for, so the devices run
// concurrently; no collective, hence no RCCL.  Replaces the loop of lidar_corrector_optix_benchmark.cpp:86-133 (1000 poses per
// correct()) when one GPU is not enough.

static void shard_bounds(uint32_t n, uint32_t rank, uint32_t world, uint32_t* lo, uint32_t* hi);

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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
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
  RCCL_SYM(GetErrorString, "ncclGetErrorString") RCCL_SYM(CommCount, "ncclCommCount")
#undef RCCL_SYM
  g_rccl.lib = lib;
  return true;
}

#define NCCLCHK(expr)                                                                                          \
  do {                                                                                                         \
    const ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return fail(RMCLHIP_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)


struct rmclhip_comm {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;   // one collective stream per device
  // rmclhip_comm_create_loopback: an in-process stand-in for RCCL (every "rank" is a stream of this process, ranks may share a
  // device): collectives are device-to-device copies / one small kernel, ordered with events.  It exists so that the ndev > 1 code
  // paths of the sharded entry points run -- and are checked against the unsharded results -- on a box with ONE GPU.
  bool loopback = false;
  std::vector<hipEvent_t> ev_in, ev_out;
  uint32_t reduce_rotation = 0;   // loopback only (rmclhip_comm_loopback_set_reduce_rotation): the all-reduce adds the ranks starting at this one
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
  bool weights_fresh = false;   // every rank's d_w_all holds the likelihoods of the cloud as it is (set by the all-gather, cleared by whatever rewrites attributes)
};


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
  // (RCCL chooses its own order of summation; the stand-in can be told to choose another one -- a result that must not depend on the
  // library's order is tested by rotating it)
  std::vector<const double*> rot(world);
  for (uint32_t k = 0; k < world; ++k) rot[k] = send[(k + c->reduce_rotation) % world];
  for (uint32_t r = 0; r < world; ++r) {
    HIPCHK(hipSetDevice(c->devices[r]));
    HIPCHK(launch_loopback_allreduce(rot.data(), world, recv[r], count, is_max, c->streams[r]));
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

rmclhip_status rmclhip_comm_loopback_set_reduce_rotation(rmclhip_comm* c, uint32_t first_rank) {
  if (!c || !c->loopback) return fail(RMCLHIP_ERR_INVALID, "comm_loopback_set_reduce_rotation: a loopback communicator is needed");
  c->reduce_rotation = first_rank % static_cast<uint32_t>(c->devices.size());
  return RMCLHIP_OK;
}

/* how many ranks the communicator's collectives span: ncclCommCount of the first rank's communicator for RCCL, the rank list's size for
 * the loopback stand-in (bench.py records it beside its sharded figures: a scaling run shows that RCCL saw N ranks) */
rmclhip_status rmclhip_comm_collective_ranks(rmclhip_comm* c, uint32_t* n_ranks, int* is_rccl) {
  if (!c || !n_ranks) return fail(RMCLHIP_ERR_INVALID, "comm_collective_ranks: null");
  if (is_rccl) *is_rccl = c->loopback ? 0 : 1;
  if (c->loopback) { *n_ranks = static_cast<uint32_t>(c->devices.size()); return RMCLHIP_OK; }
  int cnt = 0;
  if (g_rccl.CommCount == nullptr) return fail(RMCLHIP_ERR_UNSUPPORTED, "comm_collective_ranks: ncclCommCount not resolved");
  NCCLCHK(g_rccl.CommCount(c->comms[0], &cnt));
  *n_ranks = static_cast<uint32_t>(cnt);
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
  h->weights_fresh = false;
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
  if (rmclhip_status st = comm_wait_all(h->comm)) return st;
  h->weights_fresh = true;
  return RMCLHIP_OK;
}

// PCDSensorUpdater*::update on every device's block of the particles (concurrently: one stream per device), then the
// weight all-gather
rmclhip_status rmclhip_pf_update_sharded(rmclhip_pf_sharded* h, const rmclhip_range_measurement* beams, uint32_t n_beams,
                                         const rmclhip_transform* Tsb) {
  ApiGuard guard_("rmclhip_pf_update_sharded");
  if (!h || !Tsb || (n_beams && !beams)) return fail(RMCLHIP_ERR_INVALID, "pf_update_sharded: null");
  h->weights_fresh = false;
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
  h->weights_fresh = false;   // (a particle that crosses a wall gets likelihood {0, 0, MAX})
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

// global {sum, max} of the likelihoods (simple_stats_kernel, resampling.cu:41-92; consumer rmcl_localization.cpp:664-689 and the residual
// resampler's size_t(L / sum * N)).  Round 6: NO collective -- after the weight all-gather every rank holds all N likelihoods, so every
// rank reduces its own copy with the single-device kernel (same blocks, same order: the value is the single-device value bit for bit
// whatever library moved the weights, and does not depend on a reduction order RCCL is free to choose).  The name is kept for the ABI.
// A cloud whose attributes changed since the last gather (set_particles, motion update, resampling) is gathered first.
rmclhip_status rmclhip_pf_allreduce_stats(rmclhip_pf_sharded* h, rmclhip_likelihood_stats* out) {
  ApiGuard guard_("rmclhip_pf_allreduce_stats");
  if (!h || !out) return fail(RMCLHIP_ERR_INVALID, "pf_allreduce_stats: null");
  if (h->n_total == 0) { out->sum = 0.f; out->max = 0.f; return RMCLHIP_OK; }
  if (!h->weights_fresh)
    if (rmclhip_status st = rmclhip_pf_allgather_weights(h)) return st;
  const uint32_t world = static_cast<uint32_t>(h->ranks.size());
  for (uint32_t r = 0; r < world; ++r) {
    PfRank& R = h->ranks[r];
    HIPCHK(hipSetDevice(R.ctx->device));
    HIPCHK(launch_likelihood_stats_dense(R.d_w_all, h->n_total, R.rs->d_psum.p, R.rs->d_pmax.p, R.rs->d_out.p, h->comm->streams[r]));
    trace('E', r);
  }
  // the host needs ONE copy: rank 0's (the other ranks keep theirs on the device, ordered on their streams)
  PfRank& R0 = h->ranks[0];
  HIPCHK(hipSetDevice(R0.ctx->device));
  HIPCHK(hipMemcpyAsync(R0.rs->h_out, R0.rs->d_out.p, 2 * sizeof(float), hipMemcpyDeviceToHost, h->comm->streams[0]));
  HIPCHK(hipStreamSynchronize(h->comm->streams[0]));
  trace('W', 0u);
  out->sum = R0.rs->h_out[0];
  out->max = R0.rs->h_out[1];
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
  h->weights_fresh = false;   // the cloud is about to be replaced
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

