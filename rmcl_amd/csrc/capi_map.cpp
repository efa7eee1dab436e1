// capi_map.cpp -- see capi_internal.h
#include "capi_internal.h"

#include <cstdlib>



RMCL_INTERNAL thread_local std::string g_err;

const char* rmclhip_last_error(void) { return g_err.c_str(); }
const char* rmclhip_version(void) { return "rmclhip 0.1 (gfx950)"; }

// ---- context ---------------------------------------------------------------------------------
rmclhip_status rmclhip_ctx_create(int device, rmclhip_ctx** out) {
  ApiGuard guard_("rmclhip_ctx_create");
  if (!out) return fail(RMCLHIP_ERR_INVALID, "ctx_create: out is null");
  *out = nullptr;
  // Every operator (and the free statistics_p2l) runs on a stream of its own, and the sensors of one MICP correction -- or two operators
  // a node keeps in flight -- count on those streams running CONCURRENTLY.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
  // queues (default 4) in turn: with a fifth stream in the process two of them share a queue and their kernels run one after the other
  // (round 6, measured: a two-sensor correction 46 -> 63 us, two scans in flight 11.4 -> 7.8 G rays/s, once the bench held five
  // streams).  The runtime reads the variable when it initialises, i.e. at the process's first HIP call: if that is this one and the
  // caller has not chosen a value, ask for eight queues.  A process that has initialised HIP before (another library) sets
  // GPU_MAX_HW_QUEUES itself -- INTEGRATION.md.
  if (std::getenv("GPU_MAX_HW_QUEUES") == nullptr) (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
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

// the scratch of rmclhip_statistics_p2l (capi_rcc.cpp) belongs to the context and goes with its last holder
rmclhip_ctx::~rmclhip_ctx() {
  if (p2l_stream == nullptr) return;
  (void)hipSetDevice(device);
  (void)hipStreamSynchronize(p2l_stream);
  if (p2l_partials) (void)hipFree(p2l_partials);
  if (p2l_tickets) (void)hipFree(p2l_tickets);
  if (p2l_h_stats) (void)hipHostFree(p2l_h_stats);
  if (p2l_h_done) (void)hipHostFree(p2l_h_done);
  (void)hipStreamDestroy(p2l_stream);
}

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
  out->n_tri_records = bi.n_records;
  out->spatial_splits = bi.spatial_splits;
  out->reserved = 0;
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
RMCL_INTERNAL rmclhip_status map_upload(rmclhip_ctx* ctx, const BvhHost& bvh, rmclhip_map** out) {
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
  // the 16-wide twins for the cooperative descent of find kind 32 (512 B per node: 8 MB for a 100 k-triangle map), derived on the device;
  // maps beyond kMaxNodes16 nodes (~18 M triangles) go without -- the descent then walks the four-wide nodes
  constexpr size_t kMaxNodes16 = 6000000;   // (3 GB of twins for the largest: a 10 M-triangle map carries 1.7 GB, of 288)
  size_t c16b = 0;
  if (e == hipSuccess && bvh.cnodes.size() <= kMaxNodes16) {
    c16b = bvh.cnodes.size() * sizeof(Node16C);
    e = hipMalloc(reinterpret_cast<void**>(&m->d_cnodes16), c16b);
    if (e == hipSuccess) e = launch_build_cnodes16(m->d_cnodes, static_cast<uint32_t>(bvh.cnodes.size()), m->d_cnodes16, nullptr);
  }
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
    if (m->d_cnodes16) (void)hipFree(m->d_cnodes16);
    if (m->d_tris) (void)hipFree(m->d_tris);
    delete m;
    return fail(e == hipErrorOutOfMemory ? RMCLHIP_ERR_NOMEM : RMCLHIP_ERR_HIP,
                std::string("map_create upload: ") + hipGetErrorString(e));
  }
  m->bytes = nb + qb + qpb + cb + c16b + tb + fb + fpb;
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
    if (map->d_cnodes16) (void)hipFree(map->d_cnodes16);
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


