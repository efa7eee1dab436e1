/*
 * rmclhip.h -- C ABI of librmclhip.so: the MI355X (gfx950) implementation of
 * RMCL / MICP-L's ray-casting-correspondence + pose-correction hot path.
 *
 * This is the drop-in boundary.  Every entry point names the
 * reference interface it replaces (paths relative to the uos/rmcl tree).  The
 * C++ adapters in include/rmcl_hip/ present it with the reference's class
 * shapes (Correspondences_<MemT>, SensorUpdater<MemT>); INTEGRATION.md shows the
 * reference-side glue.
 *
 * Conventions
 *  - plain pointers + sizes, no C++ / torch types; all structs are PODs with the
 *    reference's in-memory layouts (rmagine::Transform = {Quaternion{x,y,z,w},
 *    Vector{x,y,z}, uint32 stamp} = 32 B, pinned by
 *    rmcl_ros/src/nodes/rmcl_localization.cpp:245-249).
 *  - every function returns rmclhip_status (0 = OK) and never throws; the text
 *    of the last error on the calling thread is rmclhip_last_error().
 *    (The reference throws std::runtime_error -- micp_localization.cpp:613,
 *    PCDSensorUpdaterOptix.cpp:179-192 -- the C++ adapters rethrow.)
 *  - there is NO CPU fallback: without a HIP device every compute call fails
 *    with RMCLHIP_ERR_NO_DEVICE.
 *  - a handle is thread-compatible (no concurrent calls on one handle); each
 *    rcc / pf handle owns one HIP stream; calls are synchronous on return unless
 *    suffixed _async.
 *  - pointers suffixed _dev are device pointers (hipMalloc'd, e.g. a torch
 *    tensor's data_ptr()); all other pointers are host memory.
 */
#ifndef RMCLHIP_H
#define RMCLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMCLHIP_VERSION_MAJOR 0
#define RMCLHIP_VERSION_MINOR 1

typedef int rmclhip_status;
enum {
  RMCLHIP_OK = 0,
  RMCLHIP_ERR_INVALID = 1,    /* bad argument / state (reference: std::runtime_error) */
  RMCLHIP_ERR_NO_DEVICE = 2,  /* no HIP device: the product has no CPU path */
  RMCLHIP_ERR_HIP = 3,        /* a HIP runtime call failed (reference: RM_CUDA_CHECK throw) */
  RMCLHIP_ERR_NOMEM = 4,
  RMCLHIP_ERR_UNSUPPORTED = 5
};

/* ---- PODs ------------------------------------------------------------------ */
typedef struct { float x, y, z; } rmclhip_vec3;
typedef struct { float x, y, z, w; } rmclhip_quat;
/* rmagine::Transform (rmcl_localization.cpp:245-249) */
typedef struct { rmclhip_quat R; rmclhip_vec3 t; uint32_t stamp; } rmclhip_transform;
typedef struct { float min, inc; uint32_t size; } rmclhip_discrete_interval;
typedef struct { float min, max; } rmclhip_interval;
/* rmagine::SphericalModel; field meaning pinned by rmcl_ros/src/util/conversions.cpp:22-34
 * (phi = vertical/rows/height, theta = horizontal/cols/width) */
typedef struct {
  rmclhip_discrete_interval phi;
  rmclhip_discrete_interval theta;
  rmclhip_interval range;
} rmclhip_spherical_model;
/* rmagine::CrossStatistics (members pinned by micp_localization.cpp:87-105,934,1010-1011).
 * covariance is row-major, C(r,c) = 1/n sum (m_i - model_mean)_r (d_i - dataset_mean)_c */
typedef struct {
  rmclhip_vec3 dataset_mean;
  rmclhip_vec3 model_mean;
  float covariance[9];
  uint32_t n_meas;
} rmclhip_cross_statistics;
/* rmagine::Gaussian1D + rmcl::ParticleAttributes (ParticleAttributes.hpp:18-34), 36 B */
typedef struct { float mean, sigma; uint32_t n_meas; } rmclhip_gaussian1d;
typedef struct { rmclhip_gaussian1d likelihood; float state_sigma[6]; } rmclhip_particle_attributes;
/* rmcl::RangeMeasurement (RangeMeasurement.hpp:10-21), 64 B; cov row-major */
typedef struct { rmclhip_vec3 orig, dir; float range; float cov[9]; } rmclhip_range_measurement;
/* sensor_update.* parameters, defaults PCDSensorUpdaterEmbree.cpp:122-134
 * (== optix/EvaluationDataOptix.hpp:62-81 minus pointers) */
typedef struct {
  float dist_sigma;                 /* 2.0  */
  float real_hit_sim_miss_error;    /* 100  */
  float real_miss_sim_hit_error;    /* 100  */
  float real_miss_sim_miss_error;   /* 0    */
  rmclhip_interval sensor_range;    /* [0.05, 80] */
  uint32_t max_n_meas;              /* MAX_N_MEAS = 10000, ParticleAttributes.hpp:34 */
  uint32_t correspondence_type;     /* 0 = RCC, the hybrid SURVEY.md App. B.1 recommends: UNIT face normals (as the OptiX
                                     *     program and rmagine's simulators) with the Embree updater's ray rules
                                     *     (tfar = inf, sim hit needs t > sensor_range.min);
                                     * 1 = CPC (evaluate_cpc, :88-95);
                                     * 2 = RCC exactly as PCDSensorUpdaterEmbree.cpp:18-86: like 0 but the error is taken
                                     *     against Embree's un-normalised rayhit.hit.Ng = cross(v2-v0, v0-v1), i.e. it scales
                                     *     with twice the triangle's area;
                                     * 3 = RCC exactly as optix/BeamEvaluateProgram.cu:15-130: unit normals, optixTrace
                                     *     tmax = 1e4, every hit is a sim hit (no sensor_range.min test) */
} rmclhip_pf_params;

/* sensor_msgs/PointCloud2 layout of the fields this path reads (datatype: PointField FLOAT32 = 7, FLOAT64 = 8) and
 * the row / column sub-sampling of rmcl::FilterOptions2D (scan_operations.cpp:41-52) */
typedef struct {
  uint32_t width, height, point_step, row_step;
  uint32_t offset_x, offset_y, offset_z;
  uint32_t datatype;
} rmclhip_pointcloud2_layout;
typedef struct { uint32_t skip_begin, skip_end, increment; } rmclhip_filter1d;

/* rmcl::GladiatorResamplerConfig (GladiatorResamplerConfig.hpp:7-20), defaults GladiatorResamplerGPU.cpp:34-44 */
typedef struct {
  float min_noise_tx, min_noise_ty, min_noise_tz;        /* 0.03 0.03 0 */
  float min_noise_roll, min_noise_pitch, min_noise_yaw;  /* 0 0 0.01 */
  float likelihood_forget_per_meter;                     /* 0.3 */
  float likelihood_forget_per_radian;                    /* 0.2 */
  uint32_t trans_dist_metric; /* 0 = |t| (resampling.cu:179), 1 = |t|^2 (GladiatorResamplerCPU.cpp:156) */
} rmclhip_gladiator_config;
/* rmcl::SimpleLikelihoodStats (resampling.cuh:26-30) */
typedef struct { float sum, max; } rmclhip_likelihood_stats;
/* rmcl_msgs/ParticleStats as RmclNode::estimateStats fills it (rmcl_localization.cpp:642-731) */
typedef struct {
  rmclhip_transform pose;           /* rm::markley_mean of the poses, weights likelihood.mean / sum (:703-705) */
  double covariance[36];            /* rm::covariance around that mean (:716-718), row-major 6x6: x y z roll pitch yaw */
  double likelihood_mean, likelihood_sigma, likelihood_min, likelihood_max;   /* :664-689 */
  float trans_bb_min[3], trans_bb_max[3];
  uint32_t n_particles;             /* min(n_particles, max_induction_particles) */
  uint32_t reserved;
} rmclhip_pose_estimate;

typedef struct {
  uint32_t n_faces, n_vertices;
  uint32_t n_nodes;        /* BVH4 nodes (128 B each) */
  uint32_t n_tri_records;  /* 64 B each, leaf order; >= n_faces: a face that the builder's spatial splits reference from k leaves has k
                            * identical records (round 6; == n_faces for maps no spatial split touched) */
  uint32_t max_depth;      /* BVH4 depth */
  uint32_t stack_need;     /* worst-case traversal stack entries */
  uint64_t device_bytes;
  float bbox_min[3], bbox_max[3];
  /* round 5: stack_need <= 64 holds for EVERY mesh (no mesh is refused for its shape; rm::import_embree_map takes any Assimp mesh,
     micp_localization.cpp:187-195).  These two count how often the bound had to be enforced -- 0 on ordinary meshes: object-median
     splits forced by the height budget of the binary tree, and BVH4 nodes expanded tallest-child-first instead of largest-area-first */
  uint32_t height_fallbacks, guarded_nodes;
  /* round 6: spatial splits (SBVH) the builder took: nodes cut by a plane, a straddling triangle referenced from both sides with the box
   * of its part there.  0 on regular meshes (the object split is kept unless the spatial one is >= 10 % cheaper); CAD-like mixes of
   * huge and tiny triangles and slivers get them, within a budget of extra records (n_tri_records - n_faces). */
  uint32_t spatial_splits, reserved;
} rmclhip_map_info;

typedef struct rmclhip_ctx rmclhip_ctx;
typedef struct rmclhip_map rmclhip_map;
typedef struct rmclhip_rcc rmclhip_rcc;
typedef struct rmclhip_pf rmclhip_pf;
typedef struct rmclhip_resampler rmclhip_resampler;
typedef struct rmclhip_comm rmclhip_comm;              /* RCCL communicators of one process driving several devices */
typedef struct rmclhip_pf_sharded rmclhip_pf_sharded;  /* a particle cloud block-partitioned over those devices */
typedef struct rmclhip_rcc_sharded rmclhip_rcc_sharded; /* one correspondence operator per device: pose batches block-partitioned */

/* ---- library ------------------------------------------------------------------ */
const char* rmclhip_last_error(void);
const char* rmclhip_version(void);

/* context = one device (+ default resources). device index as seen by HIP.  The context is reference counted:
 * every map / rcc / pf / resampler handle created from it holds a reference, rmclhip_ctx_destroy drops the
 * creator's, and the context is freed with its last holder -- destroy order does not matter. */
rmclhip_status rmclhip_ctx_create(int device, rmclhip_ctx** out);
void rmclhip_ctx_destroy(rmclhip_ctx* ctx);

/* How the SYNCHRONOUS calls that return a result through host-mapped memory (computeCrossStatistics, correct_once,
 * micp_correct_once) wait for it.  RMCLHIP_WAIT_SPIN (default): the calling thread polls a completion tag the last kernel
 * of the call writes (sequence number + checksum of the result, verified by the host) -- ~9 us less latency per call, one
 * busy host core while a call is in flight.  RMCLHIP_WAIT_BLOCK: hipStreamSynchronize -- the thread sleeps in the runtime.
 * The reference's node has ONE correction thread per node (micp_localization.cpp:300-302); integrators who cannot spare a
 * spinning core choose BLOCK.  Applies to every handle of the context, may be changed at any time.
 * Since round 4 the synchronous calls that return NOTHING through the host (find, find_cpc, pf_update, pf_motion_update,
 * pf_extract_weights, resampler_gladiator) wait the same way in SPIN mode: a one-thread launch behind the call's kernels stores
 * the tag (a 128x1024 find returns after 24.5 instead of 31.6 us); the call's results are complete when it returns either way. */
#define RMCLHIP_WAIT_SPIN 0
#define RMCLHIP_WAIT_BLOCK 1
rmclhip_status rmclhip_ctx_set_wait_mode(rmclhip_ctx* ctx, int mode);
rmclhip_status rmclhip_ctx_device_name(rmclhip_ctx* ctx, char* buf, size_t n);

/* ---- map ------------------------------------------------------------------------
 * replaces rm::import_embree_map / import_optix_map + scene commit
 * (micp_localization.cpp:187-195; PCDSensorUpdaterEmbree.cpp:143-174): copies the
 * indexed triangle mesh, builds the BVH on the host, uploads; immutable afterwards.
 * Ref-counted so it can live in an rm::MapMap-like registry under "<name>.hip". */
rmclhip_status rmclhip_map_create(rmclhip_ctx* ctx, const float* vertices_xyz, uint32_t n_vertices,
                                  const uint32_t* faces_ijk, uint32_t n_faces, rmclhip_map** out);
rmclhip_status rmclhip_map_retain(rmclhip_map* map);
void rmclhip_map_release(rmclhip_map* map);
rmclhip_status rmclhip_map_get_info(const rmclhip_map* map, rmclhip_map_info* out);

/* A whole scene: n_meshes indexed meshes, placed -- and possibly repeated -- by n_instances affine transforms.  This is what
 * rm::import_embree_map / import_optix_map make of an assimp scene (micp_localization.cpp:187-195: one mesh per aiMesh, one
 * instance per aiNode that references it).  The scene is flattened on the host into ONE world-space triangle soup and one
 * tree (an instance costs its triangles again; RMCL's maps are a few static meshes).  instances == NULL: every mesh once,
 * untransformed (n_instances is ignored).  transform: the first three rows of the row-major 4x4 node matrix (aiMatrix4x4's
 * own layout, a1..c4), p_map = A p_mesh + t, applied in float row by row.  Face ids the find / download calls return are
 * GLOBAL: instance i owns ids [first_face[i], first_face[i+1]), in the mesh's own face order. */
typedef struct {
  const float* vertices_xyz;
  const uint32_t* faces_ijk;
  uint32_t n_vertices, n_faces;
} rmclhip_mesh;
typedef struct {
  float transform[12];   /* rows of [A | t] */
  uint32_t mesh;         /* index into meshes[] */
  uint32_t reserved[3];
} rmclhip_instance;
rmclhip_status rmclhip_map_create_scene(rmclhip_ctx* ctx, const rmclhip_mesh* meshes, uint32_t n_meshes,
                                        const rmclhip_instance* instances, uint32_t n_instances, rmclhip_map** out);
/* the face-id offset table of a map: first_face_out gets n_instances + 1 entries (the last = n_faces); a map made by
 * rmclhip_map_create reports one instance.  first_face_out may be NULL to query n_instances. */
rmclhip_status rmclhip_map_scene_instances(const rmclhip_map* map, uint32_t* first_face_out, size_t capacity,
                                           uint32_t* n_instances);
/* global face id -> (instance, face index within that instance's mesh) */
rmclhip_status rmclhip_map_scene_locate(const rmclhip_map* map, uint32_t face_id, uint32_t* instance, uint32_t* local_face);
/* the flattening alone, host only (no device): what rmclhip_map_create_scene hands to the BVH builder.  Output pointers may
 * be NULL to query the sizes. */
rmclhip_status rmclhip_scene_flatten_host(const rmclhip_mesh* meshes, uint32_t n_meshes, const rmclhip_instance* instances,
                                          uint32_t n_instances, float* vertices_out, size_t vertices_cap_floats,
                                          uint32_t* faces_out, size_t faces_cap_dwords, uint32_t* first_face_out,
                                          size_t first_face_cap, uint32_t* n_vertices, uint32_t* n_faces);

/* Host-only BVH build (no device needed): fills caller buffers with the exact
 * arrays map_create uploads.  nodes: n_nodes*32 dwords, tris: n_tri_records*16 dwords.
 * Pass NULL buffers to query sizes via info.  Used by the CPU tests to check the
 * builder's invariants without a GPU. */
rmclhip_status rmclhip_bvh_build_host(const float* vertices_xyz, uint32_t n_vertices,
                                      const uint32_t* faces_ijk, uint32_t n_faces,
                                      rmclhip_map_info* info, uint32_t* nodes_out, size_t nodes_cap_dwords,
                                      uint32_t* tris_out, size_t tris_cap_dwords);
/* the 64-B quantised twins of the nodes (layout.h: Node4Q, 16 dwords each, same indices and child references as the
 * Node4 array of rmclhip_bvh_build_host): what the incoherent traversals (particle filter, pose batches) read */
rmclhip_status rmclhip_bvh_build_host_quantised(const float* vertices_xyz, uint32_t n_vertices, const uint32_t* faces_ijk,
                                                uint32_t n_faces, uint32_t* qnodes_out, size_t qnodes_capacity_dwords);

/* the particle filter's own tree of the same map: the builder splits ONE BVH2 down to leaves of <= 2 triangles; the map's
 * tree (leaves <= 4, rmclhip_bvh_build_host) and this one are two cuts through it and share the leaf-ordered record
 * array.  nodes_out: Node4 (32 dwords each), qnodes_out: Node4Q (16 dwords each, what k_pf_update_* reads); either may be
 * null; info->n_nodes / max_depth / stack_need describe THIS tree.  Replaces nothing in the reference (Embree / OptiX
 * build their own acceleration structures, PCDSensorUpdaterEmbree.cpp:158, PCDSensorUpdaterOptix.cpp:123-154). */
rmclhip_status rmclhip_bvh_build_host_pf(const float* vertices_xyz, uint32_t n_vertices, const uint32_t* faces_ijk,
                                         uint32_t n_faces, rmclhip_map_info* info, uint32_t* nodes_out,
                                         size_t nodes_cap_dwords, uint32_t* qnodes_out, size_t qnodes_cap_dwords);

/* ---- ray-casting correspondences (MICP-L) ------------------------------------------
 * rmcl::RCCEmbreeSpherical / RCCEmbreeO1Dn / RCCOptixSpherical
 * (rmcl/include/rmcl/registration/RCCEmbree.hpp:18-83, RCCOptix.hpp:18-93) on top of
 * rmcl::Correspondences_<MemT> (Correspondences.hpp:16-88). */
rmclhip_status rmclhip_rcc_create(rmclhip_ctx* ctx, rmclhip_map* map, rmclhip_rcc** out);
void rmclhip_rcc_destroy(rmclhip_rcc* rcc);
/* Correspondences_::setTsb (Correspondences.hpp:31-34; RCCEmbree.cpp:15-19) */
rmclhip_status rmclhip_rcc_set_tsb(rmclhip_rcc* rcc, const rmclhip_transform* Tsb);
/* rm::ModelSetter<SphericalModel>::setModel (RCCEmbree.cpp:21-24) */
rmclhip_status rmclhip_rcc_set_model_spherical(rmclhip_rcc* rcc, const rmclhip_spherical_model* model);
/* rm::ModelSetter<O1DnModel>::setModel (RCCEmbree.cpp:84-87; fields conversions.cpp:74-94):
 * dirs_xyz is width*height*3 floats, row-major buffer id = vid*width + hid */
rmclhip_status rmclhip_rcc_set_model_o1dn(rmclhip_rcc* rcc, uint32_t width, uint32_t height,
                                          rmclhip_interval range, rmclhip_vec3 orig, const float* dirs_xyz);
/* rm::ModelSetter<PinholeModel>::setModel (RCCEmbreePinhole, RCCEmbree.cpp:39-68; f = {fx, fy}, c = {cx, cy}:
 * conversions.cpp:36-60): direction = normalize((hid-cx)/fx, (vid-cy)/fy, 1) mapped optical -> x-forward frame */
rmclhip_status rmclhip_rcc_set_model_pinhole(rmclhip_rcc* rcc, uint32_t width, uint32_t height,
                                             rmclhip_interval range, float fx, float fy, float cx, float cy);
/* rm::ModelSetter<OnDnModel>::setModel (RCCEmbreeOnDn, RCCEmbree.cpp:102-130; fields conversions.cpp:96-120):
 * one origin AND one direction per ray, width*height*3 floats each, buffer id = vid*width + hid */
rmclhip_status rmclhip_rcc_set_model_ondn(rmclhip_rcc* rcc, uint32_t width, uint32_t height,
                                          rmclhip_interval range, const float* origs_xyz, const float* dirs_xyz);
/* public members Correspondences_::params.max_dist / adaptive_max_dist_min
 * (written by the node: micp_localization.cpp:608-609) */
rmclhip_status rmclhip_rcc_set_params(rmclhip_rcc* rcc, float max_dist, float adaptive_max_dist_min);
/* Correspondences_::dataset {points, mask} (filled by MICPSphericalSensorCPU::unpackMessage,
 * MICPSphericalSensorCPU.cpp:181-233).  src_is_device != 0: pointers are device memory.
 * mask may be NULL (all valid). */
rmclhip_status rmclhip_rcc_set_dataset(rmclhip_rcc* rcc, const float* points_xyz, const uint8_t* mask,
                                       uint32_t n, int src_is_device);
/* Correspondences_::dataset kept in the CALLER's device memory (rmagine::PointCloud_<VRAM_HIP>: Correspondences.hpp:24,
 * written by `correspondences_->dataset.points = dataset_cpu_.points`, MICPSphericalSensorCUDA.cpp:231-232): the handle
 * borrows the pointers -- no copy -- until the next set_dataset* call; the memory must stay valid meanwhile.
 * mask_dev may be NULL (all valid). */
rmclhip_status rmclhip_rcc_set_dataset_view(rmclhip_rcc* rcc, const float* points_xyz_dev, const uint8_t* mask_dev, uint32_t n);
/* convenience mirror of unpackMessage: ranges -> dataset points = dir(vid,hid)*range (+orig for
 * O1Dn), mask = range in [range.min, range.max]; returns valid count */
rmclhip_status rmclhip_rcc_set_dataset_from_ranges(rmclhip_rcc* rcc, const float* ranges, uint32_t n,
                                                   uint32_t* n_valid_out);
/* RCC*::find(Tbm_est) (RCCEmbree.cpp:26-36, RCCOptix.cpp:28-43): grow-only model buffers,
 * simulate {points, normals, hits} (+ ranges, face ids) in the SENSOR frame. No-op on an empty model. */
rmclhip_status rmclhip_rcc_find(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est);
rmclhip_status rmclhip_rcc_find_async(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est);
rmclhip_status rmclhip_rcc_sync(rmclhip_rcc* rcc);
/* Wire-format input: the bytes of a sensor_msgs/PointCloud2 become the O1Dn sensor model AND the dataset in one
 * device pass -- estimateModelAndData (rmcl_ros/src/util/conversions.cpp:869-1002: range = |p|, dir = p / range,
 * non-finite -> 0) + filter (scan_operations.cpp:41-116; NULL = keep every row / column) +
 * MICPO1DnSensorCPU::unpackMessage (MICPO1DnSensorCPU.cpp:176-227: point = dir * range + orig,
 * mask = range within `range`).  Replaces set_model_o1dn + set_dataset for clouds; `data` may be a host or a
 * device pointer.  Returns the filtered image size and the number of valid measurements. */
rmclhip_status rmclhip_rcc_set_input_pointcloud2(rmclhip_rcc* rcc, const uint8_t* data, size_t nbytes,
                                                 const rmclhip_pointcloud2_layout* layout,
                                                 const rmclhip_filter1d* filter_height,
                                                 const rmclhip_filter1d* filter_width, rmclhip_interval range,
                                                 int src_is_device, uint32_t* out_width, uint32_t* out_height,
                                                 uint32_t* n_valid);
/* rmcl::CPCEmbree::find (rmcl/src/rmcl/registration/CPCEmbree.cpp:18-44), closest-point correspondences (`type: CP`):
 * for every dataset point Pm = Tsm * d_i the nearest surface point of the map; model buffers (sized like the
 * dataset) receive hits = (distance <= params.max_dist), points = Tms * p_closest, normals = Tms.R * n_face
 * (+ ranges = distance, face ids).  computeCrossStatistics then works unchanged. */
rmclhip_status rmclhip_rcc_find_cpc(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est);
/* Tracking (default on): the operator remembers which triangle every dataset point was closest to and starts the next
 * find_cpc of the SAME dataset with that triangle's distance as the search bound.  The triangle is a real candidate, so the
 * answer -- min distance, then min face id -- is the same bit for bit; what changes is the number of leaves a query visits
 * when the pose moved little between calls (an ICP loop).  A new dataset, or on = 0, starts cold. */
rmclhip_status rmclhip_rcc_set_cpc_tracking(rmclhip_rcc* rcc, int on);
/* Bounded search (default off = the reference's semantics, CPCEmbree.cpp:36-41: the global closest point of EVERY dataset point,
 * hits = (d <= params.max_dist)).  On: nothing farther than max_dist is looked for.  hits, and every output of a point that
 * hits, are unchanged bit for bit; a point with no surface within max_dist gets hits = 0 and NaN point / normal / distance,
 * face id 0xFFFFFFFF -- it is gated out of every statistic either way.  A first (cold) find then costs about what a tracked one
 * does; the two options combine. */
rmclhip_status rmclhip_rcc_set_cpc_bounded(rmclhip_rcc* rcc, int on);
/* The map's near grid (default on): ~2 M cubic cells over the map's box, each holding the triangle closest to its centre, built on the
 * first closest-point query of any operator of the map (one launch, a few ms, 4 B per cell).  A point without a tracking seed --
 * every point of a COLD query: first scan, new dataset, tracking off -- starts from the record of its cell: an actual candidate,
 * hence a bound within about a cell of the answer, so the cold query prunes like a tracked one (CPCEmbree.cpp:18-44 has no such
 * state: Embree's rtcPointQuery starts unbounded every time).  Results do not depend on it.  on = 0: A/B. */
rmclhip_status rmclhip_rcc_set_cpc_grid(rmclhip_rcc* rcc, int on);
/* Correspondences{CPU,CUDA}::computeCrossStatistics (CorrespondencesCPU.cpp:10-39):
 * max_dist' = max_dist (1-p) + adaptive_max_dist_min p; rm::statistics_p2l(T_snew_sold, ...)
 * Round 4: answered on the host from the moments of the current find's correspondences when a published set covers the call
 * (rmclhip_ccs_info below), by a streaming reduction otherwise -- same result.  CONTRACT that makes this sound: the dataset's
 * CONTENTS do not change between a find and the computeCrossStatistics calls that follow it unless one of the rmclhip_rcc_set_dataset*
 * calls announces it (they drop the set); the reference holds data_correction_mutex_ over both (micp_localization.cpp:868,968).
 * For the same reason rmclhip_rcc_find[_async] may READ the dataset (a find that was followed by such calls last time forms the
 * moments in its epilogue): a view handed over with rmclhip_rcc_set_dataset_view must stay valid until it is replaced. */
rmclhip_status rmclhip_rcc_compute_cross_statistics(rmclhip_rcc* rcc, const rmclhip_transform* T_snew_sold,
                                                    double convergence_progress,
                                                    rmclhip_cross_statistics* out);
/* modelView() read-back (Correspondences.hpp:47-55) + ranges / face ids. Any pointer may be NULL.
 * Sizes: n = model size of the last find. */
rmclhip_status rmclhip_rcc_download(rmclhip_rcc* rcc, uint8_t* hits, float* ranges, float* points_xyz,
                                    float* normals_xyz, uint32_t* face_ids);
/* borrowed device views of the model buffers (valid until the next find that grows them) */
rmclhip_status rmclhip_rcc_device_views(rmclhip_rcc* rcc, const uint8_t** hits_dev, const float** ranges_dev,
                                        const float** points_dev, const float** normals_dev,
                                        const uint32_t** face_ids_dev, uint32_t* n);
/* MICPLocalizationNode::correctOnce inner loop for ONE sensor, entirely on the device
 * (micp_localization.cpp:900-964 with MICPSensor.hpp:146-184): find(Tom*Tbo) once, then n_iter x
 * { T_bnew_bold = ~Tbo T_onew_oold Tbo; T_snew_sold = ~Tsb T_bnew_bold Tsb; reduce; Tsb*, Tbo*;
 *   umeyama; T_onew_oold *= T_inner }.  Outputs T_onew_oold and the last merged statistics (odom frame). */
rmclhip_status rmclhip_rcc_correct_once(rmclhip_rcc* rcc, const rmclhip_transform* Tom,
                                        const rmclhip_transform* Tbo, uint32_t n_iter,
                                        double convergence_progress, int refind_each_iteration,
                                        rmclhip_transform* T_onew_oold_out,
                                        rmclhip_cross_statistics* stats_o_out);
/* MICPLocalizationNode::correctOnce inner loop for N <= 8 sensors of one device (micp_localization.cpp:900-964): one find per
 * sensor at Tbm = Tom * Tbo[s], then n_iter x { per sensor statistics at T_snew_sold (MICPSensor.hpp:178) in its own frame,
 * Cs_o = Tbo * (Tsb * stats_s), Cmerged_o += Cs_o, Cmerged_weighted_o += Cs_o with n_meas *= merge_weight_multiplier[s]
 * (truncating, :934), T_onew_oold *= umeyama(Cmerged_weighted_o) } -- all on the device, one synchronisation at the end.
 * merge_weight_multiplier may be NULL (all 1).  merged_out: the UNWEIGHTED statistics of the last iteration (:1010-1011).
 * Every sensor's find (and moment pass) is enqueued on that sensor's OWN stream, so the scans run concurrently; the loop runs on
 * sensor 0's stream behind all of them (an in-kernel flag per sensor; stream events in the per-iteration fallback).  When the call
 * returns, all of it is complete and every sensor's model buffers hold its scan at Tom * Tbo[s]. */
rmclhip_status rmclhip_micp_correct_once(rmclhip_rcc* const* sensors, uint32_t n_sensors, const rmclhip_transform* Tom,
                                         const rmclhip_transform* Tbo, const double* merge_weight_multiplier, uint32_t n_iter,
                                         double convergence_progress, rmclhip_transform* T_onew_oold_out,
                                         rmclhip_cross_statistics* merged_out);
/* stale v1 SphereCorrector API (lidar_corrector_embree_benchmark.cpp:86-133): nposes hypotheses share
 * the dataset; one raycast + reduction + Umeyama per pose; Tdelta_out[i] such that
 * T_new[i] = Tbm[i] * Tdelta_out[i]. */
rmclhip_status rmclhip_rcc_correct_batch(rmclhip_rcc* rcc, const rmclhip_transform* Tbm, uint32_t nposes,
                                         rmclhip_transform* Tdelta_out, rmclhip_cross_statistics* stats_out);

/* ---- pose batches over several devices (north_star: "pose-corrections/s at 1/2/4/8 GPUs"; SURVEY 8(e): MICP pose batches shard
 * by pose with NO exchange) -------------------------------------------------------------------------------------------------
 * One process, one operator replica per entry of `devices` (NULL: devices 0 .. ndev-1) over ONE host BVH build.  No collective is
 * involved, hence no RCCL communicator; an entry may repeat a device (two replicas on one GPU: of use to tests and to fill a GPU
 * that a single stream of small batches leaves idle).  Configure every replica through its borrowed handle with the ordinary
 * setters (rmclhip_rcc_set_tsb / _set_model_* / _set_params / _set_dataset*): the corrector does not care whether they agree.
 * rmclhip_rcc_sharded_correct_batch = rmclhip_rcc_correct_batch (the v1 SphereCorrector::correct shape,
 * lidar_corrector_optix_benchmark.cpp:86-133) with poses [lo, hi) = rmclhip_shard_bounds(nposes, rank, ndev) on replica `rank`:
 * every replica's chain is enqueued before any is waited for, results arrive in pose order, bit-identical to the unsharded call
 * (a pose's result does not depend on its neighbours in the batch). */
rmclhip_status rmclhip_rcc_sharded_create(const int* devices, uint32_t ndev, const float* vertices_xyz, uint32_t n_vertices,
                                          const uint32_t* faces_ijk, uint32_t n_faces, rmclhip_rcc_sharded** out);
void rmclhip_rcc_sharded_destroy(rmclhip_rcc_sharded* h);
uint32_t rmclhip_rcc_sharded_size(const rmclhip_rcc_sharded* h);
rmclhip_status rmclhip_rcc_sharded_replica(rmclhip_rcc_sharded* h, uint32_t rank, rmclhip_rcc** rcc_borrowed);
rmclhip_status rmclhip_rcc_sharded_correct_batch(rmclhip_rcc_sharded* h, const rmclhip_transform* Tbm, uint32_t nposes,
                                                 rmclhip_transform* Tdelta_out, rmclhip_cross_statistics* stats_out);

/* kernel variant selection: bits 0..3 (+ bit 13 = 16 more, bit 14 = 32 more) traversal kind.  15 = automatic, the
 * default: four lanes per ray up to 57344 rays in flight (kind 2); above that one lane per ray STARTING AT THE MAP'S FRONTIER (the
 * wave culls the <= 256 references of BFS depth 4 against its tile's pyramid and every ray tests the few survivors: the top levels of
 * the descent cost one cooperative pass) -- kind 23 on the full-precision nodes up to 262144 rays, kind 24 on the 64-B quantised
 * nodes of the particle filter's tree (leaves <= 2 triangles) above (large scans, pose batches).  librmclhip.so builds those three plus
 * 0 = wave packet and 32 (round 6) = the wave keeps descending COOPERATIVELY below the frontier: lane l tests entry (l & 15) of the
 * 16-wide twin of surviving node (l >> 4) against the tile's pyramid (two tree levels per round trip and wave instead of one round
 * trip per node visit and ray), then every ray marks the final leaves whose box it enters in a 64-bit mask and tests just those -- no
 * ordered hand-over, no stack.  29 % faster than 23 on the open benchmark map (C2 sphere-100k 17.2 -> 12.2 us, sphere-1M 22.3 -> 19.6),
 * slower on rooms whose grazing tiles see long strips of the map (room-100k 25.4 -> 30.8 us): the rule does not select it,
 * rmclhip_rcc_autotune measures it on the caller's map.  Every other kind (31 = round 6's first form of 32 with the sorted
 * hand-over among them) is a measured-and-rejected or superseded experiment that lives in librmclhip_lab.so (include/rmclhip_lab.h lists
 * them) and is accepted here only while that library is loaded (RMCLHIP_ERR_UNSUPPORTED otherwise),
 * bits 4..7 = 1 + log2(tile width) of the wave's scan-image tile (0 = automatic: 16 wide x 4 tall, 8x8 for the wave packet),
 * bit 8 = fused last-block reduction tail (A/B), bit 9 = disable the hipGraph MICP loop (A/B), bits 10..12 = form
 * of the MICP loop (0 = one launch per iteration, the default; 1 = reduce + solve launches; 2..6 = persistent
 * kernel with 16..256 blocks and a grid barrier, A/B) */
rmclhip_status rmclhip_rcc_set_variant(rmclhip_rcc* rcc, int variant);
/* Moment form of the schedule-(R) loop of rmclhip_rcc_correct_once (refind_each_iteration = 0; micp_localization.cpp:900-964
 * iterates statistics_p2l + umeyama over FIXED correspondences): the gate |dist| < max_dist is the only part of the 16 raw
 * sums that is not a polynomial in the pre-transform, so correspondences whose gate decision cannot change while the
 * pre-transform stays within bounds learnt from the previous corrections are summarised ONCE as 82 moments, the few others
 * are re-evaluated every iteration, and all iterations run in one single-workgroup launch.  When a pre-transform leaves the
 * bounds, or too many correspondences are undecided, the library falls back to one streaming launch per iteration: the
 * result never depends on the bounds (both forms agree to f64 summation order).  mode 0 = never; 1 = automatic (default): TWO plain
 * launches -- the find (per-ray kind 23 or quad kind 2: every single scan the rule serves) forms the moments in its epilogue, a
 * 10 x 10 product of factor vectors per correspondence through f64 MFMA; a fold launch sums the per-workgroup rows and publishes
 * {82 moments, the undecided correspondences (<= 1024)} to the host -- and the ITERATIONS RUN ON THE HOST (~0.5 us each; the same
 * published set then also answers rmclhip_rcc_compute_cross_statistics without a launch, see rmclhip_ccs_info); more than 1024
 * undecided: the device loop of mode 4 on the same rows.  2 = device loop, find + moments pass + loop replayed from a hipGraph
 * behind an H2D copy node (A/B); 3 = device loop, three plain launches (A/B: the moments always in a pass of their own); 4 = device
 * loop behind a find with the moment epilogue (round 3's default, A/B: one lane of the GPU solves every iteration). */
typedef struct {
  uint32_t attempts, done, cap_exits, overflows;   /* outcomes since the operator was created */
  uint32_t last_code;                              /* 0 done, 1 pre-transform left the bounds, 2 too many undecided: > 1024 leaves the host form for the device loop, > 4096 the moment form altogether */
  uint32_t last_uncertain;                         /* undecided correspondences of the last attempt */
  float last_rho, last_tau;                        /* largest |2 sin(theta/2)| and |t| of its pre-transforms */
  float rho_cap, tau_cap;                          /* bounds the next attempt will use */
  uint32_t last_setup_clocks, last_loop_clocks;    /* diagnostics: shader clocks the single-workgroup launch spent before /
                                                    * in its iterations (last completed attempt; 0 when the host ran them) */
  uint32_t host_loops;                             /* attempts whose iterations ran on the host (mode 1) */
} rmclhip_micp_fast_info;
rmclhip_status rmclhip_rcc_set_micp_fast(rmclhip_rcc* rcc, int mode);
/* How rmclhip_rcc_compute_cross_statistics was served since the operator was created.  With the moment form on (mode != 0) a call is
 * answered ON THE HOST from the 82 moments + the undecided correspondences of the current find whenever a published set covers its
 * (pre-transform, max_dist'): no launch, no wait (micp_localization.cpp:915-964 calls it once per sensor and iteration on fixed
 * correspondences).  The set comes from the find itself when the previous find was followed by such calls (`speculative_finds`:
 * moment epilogue + publish, classified for max_dist' within +-8 % of the last one), else from ONE moment pass the first call
 * after a find pays (`passes`); everything else takes the streaming reduction (calls - from_moments). */
typedef struct {
  uint32_t calls;              /* computeCrossStatistics calls that were eligible (moment form on) */
  uint32_t from_moments;       /* ... answered on the host without a launch */
  uint32_t passes;             /* moment passes run by a call (no covering set yet) */
  uint32_t speculative_finds;  /* finds that formed the moments in their epilogue */
} rmclhip_ccs_info;
rmclhip_status rmclhip_rcc_ccs_info(const rmclhip_rcc* rcc, rmclhip_ccs_info* out);
/* The host half of the moment form on its own (no device): classifies the n correspondences (D = dataset point, I = model point,
 * N = model normal, valid nullable) for every max_dist' in [gate_lo, gate_hi] and every pre-transform within (rho_cap = |2 sin
 * theta/2|, tau_cap = |t|), accumulates the 82 moments of the certainly-gated-in ones, keeps the undecided ones (<= 1024 for the host, <= 4096 for the device loop), and
 * evaluates rm::statistics_p2l(Tpre, ..., max_dist) from them.  *covered = 0 (and Identity statistics) when (Tpre, max_dist) lies
 * outside what the set was formed for or more than 1024 correspondences are undecided: the library then uses the streaming
 * reduction.  What the device publishes per find is this set; the entry point exists so that the arithmetic can be checked
 * against the reference's per-element loop (MICPSensorCPU.cpp:70-84) without a GPU. */
rmclhip_status rmclhip_host_moment_statistics(const float* dataset_points, const float* model_points, const float* model_normals,
                                              const uint8_t* valid, uint32_t n, float gate_lo, float gate_hi, float rho_cap,
                                              float tau_cap, const rmclhip_transform* Tpre, float max_dist,
                                              rmclhip_cross_statistics* out, uint32_t* n_undecided, int* covered);
rmclhip_status rmclhip_rcc_micp_fast_info(const rmclhip_rcc* rcc, rmclhip_micp_fast_info* out);
/* the traversal (bits 0..3 above, never 15) a find of `nposes` scans of the current model would launch */
rmclhip_status rmclhip_rcc_find_variant(const rmclhip_rcc* rcc, uint32_t nposes, int* variant_out);
/* Measurement instead of brackets: the automatic rule above was tuned on two synthetic maps; which traversal is fastest for a
 * single scan depends on the map (open / occluded), the model's size and shape, and where the sensor is.  This call times the
 * product's single-scan kinds (2, 23, 24 -- the latter two with and without the frontier start -- and 32 with a short and a long list) on THIS operator's map and model at
 * the given pose (40 short launches each, HIP events), then the winner's tile shape (4, 8, 16 or 32 rays wide), then which workgroup
 * computes which tile (as the hardware deals workgroups over the XCDs -- the default --, an eighth of the image per XCD, or a CU's two
 * workgroups from the image's two halves), and makes the fastest combination the automatic choice for single scans until the model
 * or the tiling changes (~10 ms in all).  Results do not
 * depend on the kind (bit-identical); batches keep the rule.  Opt-in: never run behind the caller's back.
 * chosen_kind / kernel_ms may be NULL. */
rmclhip_status rmclhip_rcc_autotune(rmclhip_rcc* rcc, const rmclhip_transform* Tbm_est, int* chosen_kind, float* kernel_ms);
/* the same for pose batches (find_batch / correct_batch) of about nposes scans: kinds 23 / 24, each with and without the frontier
 * start (on an occluded map the per-wave culling can cost a batch more than the top levels it skips).  Both calls report the
 * choice as 2 / 23 / 24 / 32 or -- frontier start off -- as 19 / 22, round 2's numbers for the same traversals. */
rmclhip_status rmclhip_rcc_autotune_batch(rmclhip_rcc* rcc, const rmclhip_transform* Tbm, uint32_t nposes, int* chosen_kind,
                                          float* kernel_ms);
/* rm::Simulator::simulate(Memory<Transform>, Bundle&) (batch form, lidar_corrector_embree_benchmark.cpp:117):
 * one launch for nposes x H x W rays; model buffers become pose-major [pose][vid][hid]. */
rmclhip_status rmclhip_rcc_find_batch(rmclhip_rcc* rcc, const rmclhip_transform* Tbm, uint32_t nposes);

/* ---- the rmagine-level Simulator interface ---------------------------------------------------------------
 * The reference's RCC classes ARE simulators: `class RCCEmbreeSpherical : public CorrespondencesCPU, public
 * rmagine::ModelSetter<SphericalModel>, protected rmagine::SphereSimulatorEmbree` (rmcl/include/rmcl/registration/RCCEmbree.hpp:18-22),
 * find() is `simulate(Tbm_est, model_buffers_)` (RCCEmbree.cpp:35), and the simulators are also used on their own
 * (rmcl_ros/src/nodes/filter/scan_map_segmentation_embree.cpp:38-39,76-87; lidar_corrector_embree_benchmark.cpp:117).  Here the
 * rmclhip_rcc handle is that simulator: rmclhip_rcc_set_tsb / rmclhip_rcc_set_model_* are Simulator::setTsb / ::setModel, and
 * rmclhip_rcc_simulate is Simulator::simulate(const Transform&, BundleT&) / simulate(Memory<Transform>&, BundleT&).
 *
 * Bundle attribute selection.  rmagine's simulate<Bundle<Ranges, Normals>> writes only the attributes the bundle names
 * (scan_map_segmentation_embree.cpp:80-87); MICP's own bundle is {Points, Normals, Hits} = 25 B/ray (Correspondences.hpp:81-85).
 * An attribute is selected by a bit; the kernel skips the stores of the others. */
#define RMCLHIP_OUT_HITS 1u      /* rmagine::Hits     uint8  per ray */
#define RMCLHIP_OUT_RANGES 2u    /* rmagine::Ranges   float  per ray (miss: range.max + 1) */
#define RMCLHIP_OUT_POINTS 4u    /* rmagine::Points   3 floats, sensor frame (miss: NaN) */
#define RMCLHIP_OUT_NORMALS 8u   /* rmagine::Normals  3 floats, sensor frame, facing the ray (miss: NaN) */
#define RMCLHIP_OUT_FACE_IDS 16u /* rmagine::FaceIds  uint32 (miss: 0xFFFFFFFF) */
#define RMCLHIP_OUT_ALL 31u
#define RMCLHIP_OUT_MICP (RMCLHIP_OUT_HITS | RMCLHIP_OUT_POINTS | RMCLHIP_OUT_NORMALS) /* Correspondences_::model_buffers_ */
/* Which of the operator's OWN model buffers find / find_batch write (default RMCLHIP_OUT_ALL).  A deselected buffer is not written --
 * it keeps what the last find that selected it left there (nothing, if none did: download / device_views of it then fail / return
 * NULL).  computeCrossStatistics and the corrections read {hits, points, normals}: they fail with RMCLHIP_ERR_INVALID while one of
 * those is deselected.  mask == 0 or bits beyond RMCLHIP_OUT_ALL: RMCLHIP_ERR_INVALID. */
rmclhip_status rmclhip_rcc_set_outputs(rmclhip_rcc* rcc, uint32_t mask);
rmclhip_status rmclhip_rcc_get_outputs(const rmclhip_rcc* rcc, uint32_t* mask);
/* a rmagine Bundle over CALLER-owned device memory: one nullable pointer per attribute (NULL = the bundle does not carry it);
 * every non-null buffer holds nposes * width * height elements, index pose * W * H + vid * W + hid */
typedef struct {
  uint8_t* hits_dev;
  float* ranges_dev;
  float* points_xyz_dev;
  float* normals_xyz_dev;
  uint32_t* face_ids_dev;
} rmclhip_bundle_views;
/* rm::Simulator::simulate(const Transform& Tbm, BundleT& res) (nposes == 1; scan_map_segmentation_embree.cpp:87) and its batch form
 * simulate(const Memory<Transform>& Tbm, BundleT& res) (lidar_corrector_embree_benchmark.cpp:117, _optix_benchmark.cpp:119 -- there the
 * poses are device memory: Tbm_is_device != 0).  One launch for nposes x H x W rays with Tsm = Tbm[i] * Tsb, results in the sensor
 * frame into the caller's bundle.  The operator's own model buffers, its dataset and whatever computeCrossStatistics has cached of them
 * are NOT touched.  No-op on an empty model or nposes == 0.  The _async form returns once the launch is enqueued on the handle's stream
 * (rmclhip_rcc_sync waits); Tbm is copied before it returns. */
rmclhip_status rmclhip_rcc_simulate(rmclhip_rcc* rcc, const rmclhip_transform* Tbm, uint32_t nposes, int Tbm_is_device,
                                    const rmclhip_bundle_views* out);
rmclhip_status rmclhip_rcc_simulate_async(rmclhip_rcc* rcc, const rmclhip_transform* Tbm, uint32_t nposes, int Tbm_is_device,
                                          const rmclhip_bundle_views* out);
/* rm::statistics_p2l(const Transform& Tpre, const PointCloudView_<MemT>& dataset, const PointCloudView_<MemT>& model,
 * const UmeyamaReductionConstraints& params) -> CrossStatistics, as a free function on CALLER-owned device views
 * (rmcl/src/rmcl/registration/CorrespondencesCUDA.cpp:28; gate and projection pinned by rmcl_ros/src/micpl/MICPSensorCPU.cpp:70-84):
 * for i < n with dataset_mask[i] > 0 && model_mask[i] > 0: Di = Tpre * dataset_points[i], d = (Ii - Di) . Ni, kept iff |d| < max_dist,
 * Mi = Di + Ni d; means of D and M, covariance 1/n sum (M - mean_M)(D - mean_D)^T, count.  Either mask may be NULL (all valid).
 * The streaming reduction of rmclhip_rcc_compute_cross_statistics (k_reduce_partials + finalize) on the context's own stream;
 * synchronous, thread-safe per context (calls on one context serialise). */
rmclhip_status rmclhip_statistics_p2l(rmclhip_ctx* ctx, const rmclhip_transform* Tpre, const float* dataset_points_xyz_dev,
                                      const uint8_t* dataset_mask_dev, const float* model_points_xyz_dev,
                                      const float* model_normals_xyz_dev, const uint8_t* model_mask_dev, uint32_t n, float max_dist,
                                      rmclhip_cross_statistics* out);

/* ---- host-side algebra (rmagine math the callers of the hot path use) -------------- */
/* rm::umeyama_transform(CrossStatistics) (micp_localization.cpp:952-953) */
rmclhip_status rmclhip_umeyama_transform(const rmclhip_cross_statistics* stats, rmclhip_transform* out);
/* CrossStatistics::operator+= (micp_localization.cpp:936-937) */
rmclhip_status rmclhip_cross_statistics_merge(const rmclhip_cross_statistics* a,
                                              const rmclhip_cross_statistics* b,
                                              rmclhip_cross_statistics* out);
/* Transform * CrossStatistics (MICPSensor.hpp:182, micp_localization.cpp:931) */
rmclhip_status rmclhip_cross_statistics_transform(const rmclhip_transform* T,
                                                  const rmclhip_cross_statistics* s,
                                                  rmclhip_cross_statistics* out);
/* Transform::operator*, operator~ (micp_localization.cpp:926,963) */
rmclhip_status rmclhip_transform_mult(const rmclhip_transform* a, const rmclhip_transform* b,
                                      rmclhip_transform* out);
rmclhip_status rmclhip_transform_inv(const rmclhip_transform* a, rmclhip_transform* out);

/* ---- particle-filter sensor update (RMCL) --------------------------------------------
 * rmcl::PCDSensorUpdaterEmbree / PCDSensorUpdaterOptix
 * (rmcl_ros/src/rmcl/PCDSensorUpdaterEmbree.cpp:244-352, PCDSensorUpdaterOptix.cpp:174-350,
 *  kernels optix/BeamEvaluateProgram.cu:15-130) behind SensorUpdater<MemT>::update
 * (rmcl_ros/include/rmcl_ros/rmcl/SensorUpdater.hpp:18-42, ParticleUpdater.hpp:24-44). */
rmclhip_status rmclhip_pf_create(rmclhip_ctx* ctx, rmclhip_map* map, rmclhip_pf** out);
void rmclhip_pf_destroy(rmclhip_pf* pf);
rmclhip_status rmclhip_pf_set_params(rmclhip_pf* pf, const rmclhip_pf_params* params);
/* update(poses, attrs): all `n_beams` sampled measurements (sensor frame) are applied to every
 * particle IN ORDER, in one launch; attrs are updated in place on the device.
 * poses_dev / attrs_dev: device arrays of n_particles rmclhip_transform / rmclhip_particle_attributes. */
rmclhip_status rmclhip_pf_update(rmclhip_pf* pf, const rmclhip_transform* poses_dev,
                                 rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                 const rmclhip_range_measurement* beams, uint32_t n_beams,
                                 const rmclhip_transform* Tsb);
rmclhip_status rmclhip_pf_update_async(rmclhip_pf* pf, const rmclhip_transform* poses_dev,
                                       rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                       const rmclhip_range_measurement* beams, uint32_t n_beams,
                                       const rmclhip_transform* Tsb);
rmclhip_status rmclhip_pf_sync(rmclhip_pf* pf);
/* optional debug/parity output of the next update: per (particle, beam) error in metres
 * (device buffer of n_particles*n_beams floats, or NULL to disable) */
rmclhip_status rmclhip_pf_set_error_output(rmclhip_pf* pf, float* errors_dev);
/* MotionUpdater<MemT>::update inner loop: TFMotionUpdaterGPU / particle_move_and_forget_kernel
 * (rmcl_ros/src/rmcl/particle_motion.cu:11-46) and, with check_collision != 0, the wall-collision test the CPU
 * updater adds (TFMotionUpdaterCPU.cpp:17-50,207-221): pose <- pose * T_bnew_bold;
 * n_meas -= forget_rate * n_meas; a particle whose step crosses the mesh gets likelihood {0, 0, MAX_N_MEAS}.
 * forget_rate is the already combined rate (TFMotionUpdaterCPU.cpp:172-174). */
rmclhip_status rmclhip_pf_motion_update(rmclhip_pf* pf, rmclhip_transform* poses_dev,
                                        rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                        const rmclhip_transform* T_bnew_bold, double forget_rate, int check_collision);
/* gather likelihood.mean of every particle into a dense float array (the payload of the
 * multi-GPU all-gather) */
rmclhip_status rmclhip_pf_extract_weights(rmclhip_pf* pf, const rmclhip_particle_attributes* attrs_dev,
                                          uint32_t n_particles, float* weights_dev);
/* bits 0..3: traversal (0 = while-while with the hybrid LDS + scratch stack, 1 = stack entirely in LDS, 2 = single
 * loop, A/B); bits 4..6: ray scheduling (0 = rounds of one ray per lane; 1..4 = persistent lanes that fetch the
 * next ray when 8 / 16 / 32 / 48 lanes of their wave are idle); bit 7: persistent lanes on the 128-B nodes instead
 * of their 64-B quantised twins (A/B); bit 8: the round-2 kernel on the quantised nodes (A/B); bit 9: 4096 instead of
 * 2048 rays per workgroup (A/B); bit 10: traverse the map's tree (leaves <= 4 triangles) instead of the filter's own
 * (leaves <= 2, rmclhip_bvh_build_host_pf); bit 11 (A/B, round 5): a node's children in the ray's slot order instead of sorted by
 * entry distance (measured slower: more node visits than instructions saved); bit 12 (A/B): the STORED form of rounds 3 / 4 -- every
 * beam's error kept (global scratch), a dense likelihood pass and one lane per particle walking the reference's in-order
 * Gaussian1D += chain, bit-identical to the oracle's float chain -- instead of the round-5 default, the order-independent
 * accumulation (closed-form merge weights, fixed-point accumulators: no per-beam storage, the same bits for every schedule and shard,
 * mean / sigma within 2e-6 / 5e-6 relative of the sequential chain, n_meas exact).  A fresh handle uses traversal 0,
 * refill at 48, quantised nodes of the filter's tree, sorted children, the accumulation.  The round kernels (bits 4..6 = 0,
 * bits 0..1) and the round-2 kernel (bits 7 / 8) are experiments: accepted only while librmclhip_lab.so is loaded. */
rmclhip_status rmclhip_pf_set_variant(rmclhip_pf* pf, int variant);
/* schedule of the persistent-lane kernel: a wave fetches new beams once `refill_idle_lanes` of its lanes are idle (0: the
 * threshold selected by set_variant) and leaves its node phase when at most `tail_lanes` lanes still descend while
 * another holds a leaf (default 8).  The result does not depend on either (tests/test_gpu_pf.py). */
/* Ray dealing of the sensor update.  mapping 0 (default): beam-minor -- a workgroup takes ~2048 rays of a few particles, a wave's lanes
 * hold DIFFERENT beams.  mapping 1: particle-minor -- a workgroup takes `particles_per_block` (0 = 32, at most 64) consecutive SLOTS
 * and deals their rays out so that the lanes of a wave hold the same beam of consecutive slots: nearly the same ray when the cloud has
 * converged and neighbouring slots hold neighbouring particles.  order_dev (nullable, borrowed device memory, n_order = the particle
 * count it is for): slot -> particle index, e.g. the Morton order of (x, y, yaw) (rmcl_amd.synthetic.morton_order_xy_yaw is the host
 * form tests and bench.py use); null = slot i is particle i.  The rays, every beam's error and the in-order Gaussian1D merge per
 * particle are those of mapping 0: results do not depend on the mapping or the order.
 * MEASURED NEUTRAL (round 4, profiles/r04_pf_converged_mapping.txt): on a converged cloud (sigma 0.25 m / 5 deg) mapping 1 with 16
 * slots per workgroup and the Morton order is 3 % faster than mapping 0, with larger workgroups slower -- the kernel is bound by
 * VALU issue, and coherent lanes save cache lines, not instructions.  Kept as an option; nothing selects it automatically.
 * Bit 9 of `mapping` (A/B): a workgroup's beam errors wait in LDS (rounds 3) instead of the updater's global scratch (round 4: 16 KB
 * less LDS per workgroup, 3-5 % faster, identical results; profiles/r04_pf_occupancy.txt).
 * Bit 8 of `mapping` (A/B): correspondence_type 1 (closest-point errors) WITHOUT the near-grid seed its queries start from by
 * default (room-100k: 11.7 ms seeded vs 29.7 ms; identical results). */
rmclhip_status rmclhip_pf_set_mapping(rmclhip_pf* pf, int mapping, uint32_t particles_per_block, const uint32_t* order_dev,
                                      uint32_t n_order);
rmclhip_status rmclhip_pf_set_schedule(rmclhip_pf* pf, uint32_t refill_idle_lanes, uint32_t tail_lanes);
/* Beam sampling of PCDSensorUpdater{Embree,Optix}::update (PCDSensorUpdaterEmbree.cpp:276-327) on the raw
 * sensor_msgs/PointCloud2 bytes (HOST function, no device needed): `samples` uniformly random points, each with up to 100
 * retries for one without NaN, become RangeMeasurements {orig 0, dir = p / |p|, range = |p|, cov = 0.1 I}.  The reference
 * draws from an unseeded function-static engine; here: std::mt19937(seed), index = draw % (width * height).  *n_out <
 * samples iff a sample stayed invalid (the reference returns early there). */
rmclhip_status rmclhip_pf_sample_beams_pointcloud2(const uint8_t* data, size_t nbytes, const rmclhip_pointcloud2_layout* layout,
                                                   uint32_t samples, uint64_t seed, rmclhip_range_measurement* beams_out,
                                                   uint32_t* n_out);


/* ---- resampling: rmcl::GladiatorResamplerGPU (GladiatorResamplerGPU.cpp:46-81, resampling.cu:41-219) ----
 * compute_stats: {sum, max} of likelihood.mean over n particles (simple_stats_kernel), returned to the host.
 * gladiator: champions first .. first+count-1 each fight one random enemy out of ALL n_particles; the winner
 * (enemy: copied, perturbed by the min_noise_* Gaussians, n_meas *= remember_rate) lands in
 * poses_new_dev / attrs_new_dev [0 .. count).  first/count let one GPU resample its shard of an all-gathered
 * cloud.  Random numbers: Philox4x32-10, key = seed, counter = (champion index, step, draw, 0) -- reproducible
 * and independent of the sharding (the reference's cuRAND / mt19937 streams are not reproducible). */
rmclhip_status rmclhip_resampler_create(rmclhip_ctx* ctx, rmclhip_resampler** out);
void rmclhip_resampler_destroy(rmclhip_resampler* rs);
rmclhip_status rmclhip_resampler_compute_stats(rmclhip_resampler* rs, const rmclhip_particle_attributes* attrs_dev,
                                               uint32_t n, rmclhip_likelihood_stats* out);
/* the same {sum, max} of a DENSE weight vector on the device -- the all-gathered likelihood.mean of a sharded cloud (every
 * rank holds all N weights after the gather and computes the statistics locally, no second collective): the kernel and the order of
 * rmclhip_resampler_compute_stats, hence the same bits as that call on attributes holding the same values, on every rank */
rmclhip_status rmclhip_resampler_compute_stats_weights(rmclhip_resampler* rs, const float* weights_dev, uint32_t n,
                                                       rmclhip_likelihood_stats* out);
rmclhip_status rmclhip_resampler_gladiator(rmclhip_resampler* rs, const rmclhip_transform* poses_dev,
                                           const rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                           rmclhip_transform* poses_new_dev, rmclhip_particle_attributes* attrs_new_dev,
                                           uint32_t first, uint32_t count, const rmclhip_gladiator_config* config,
                                           uint64_t seed, uint32_t step);

/* rmcl::ResidualResamplerCPU::update (ResidualResamplerCPU.cpp:55-203), the PF node's second resampler plugin: until the new
 * cloud of n_new particles is full, draw a random particle and insert size_t(L / sum(L) * n_new) perturbed copies of it
 * (Gaussians of width min_noise_* / (L / max(L)); n_meas *= forget_per_meter^|dt|^2 * forget_per_radian^l2norm(dR)).  The
 * reference's loop is sequential; here the draws are a counter-based stream (draw k -> particle philox(k, step, 2)[0] % n; the
 * Gaussians of slot j from philox(j, step, 3 / 4)), so their counts, a prefix sum and the slots are three data-parallel passes
 * with the sequential loop's result.  Slots first .. first+count-1 land in poses_new_dev / attrs_new_dev [0 .. count) (a GPU
 * can fill its shard of an all-gathered cloud).  config: the gladiator's struct (trans_dist_metric is ignored).
 * n_draws_out (nullable): draws the sequential loop uses -- set when the call fills the LAST slot.  Errors: likelihoods that
 * sum to zero; shares that all truncate to zero (the reference's loop would not terminate). */
rmclhip_status rmclhip_resampler_residual(rmclhip_resampler* rs, const rmclhip_transform* poses_dev,
                                          const rmclhip_particle_attributes* attrs_dev, uint32_t n_particles,
                                          rmclhip_transform* poses_new_dev, rmclhip_particle_attributes* attrs_new_dev,
                                          uint32_t n_new, uint32_t first, uint32_t count, const rmclhip_gladiator_config* config,
                                          uint64_t seed, uint32_t step, uint64_t* n_draws_out);

/* ---- multi-GPU: ONE process drives several devices -------------------------------------------------------------
 * (the reference's localisation node is one process: rmcl_localization.cpp:482-552; particle store rmcl_localization.hpp:65-77.
 * The reference has no distributed code -- SURVEY.md 8(e) defines this part.)  Particles are block-partitioned over the
 * devices, mesh + BVH are replicated (built once), beams are identical everywhere; exchanges are RCCL collectives over xGMI
 * (librccl is loaded with dlopen by rmclhip_comm_create: single-GPU users never touch it). */
/* devices: HIP device indices (NULL = 0 .. ndev-1).  ncclCommInitAll. */
rmclhip_status rmclhip_comm_create(const int* devices, uint32_t ndev, rmclhip_comm** out);
/* TEST INFRASTRUCTURE: an in-process stand-in for the RCCL communicator -- same handle type, accepted by every rmclhip_pf_sharded_*
 * entry point; every "rank" is a stream of this process and `devices` may repeat a device; an all-gather is device-to-device
 * copies, an all-reduce one small kernel, ordered with events.  It lets the ndev > 1 code paths (grouped all-gather of weights and of
 * the cloud, the moment all-reduces, the per-rank tournament / residual fill) run, and be checked against the unsharded results,
 * on a box with ONE GPU; it is not a transport (no xGMI, no peer access set-up). */
rmclhip_status rmclhip_comm_create_loopback(const int* devices, uint32_t ndev, rmclhip_comm** out);
void rmclhip_comm_destroy(rmclhip_comm* comm);
uint32_t rmclhip_comm_size(const rmclhip_comm* comm);
/* contiguous block partition of [0, n): rank owns [lo, hi); the first n % world ranks own one extra element */
void rmclhip_shard_bounds(uint32_t n, uint32_t rank, uint32_t world, uint32_t* lo, uint32_t* hi);
/* one context + map replica + sensor updater + resampler per device of the communicator */
rmclhip_status rmclhip_pf_sharded_create(rmclhip_comm* comm, const float* vertices_xyz, uint32_t n_vertices,
                                         const uint32_t* faces_ijk, uint32_t n_faces, rmclhip_pf_sharded** out);
void rmclhip_pf_sharded_destroy(rmclhip_pf_sharded* pf);
rmclhip_status rmclhip_pf_sharded_set_params(rmclhip_pf_sharded* pf, const rmclhip_pf_params* params);
/* scatter the (host) cloud: rank r receives particles [lo_r, hi_r) */
rmclhip_status rmclhip_pf_sharded_set_particles(rmclhip_pf_sharded* pf, const rmclhip_transform* poses,
                                                const rmclhip_particle_attributes* attrs, uint32_t n_total);
rmclhip_status rmclhip_pf_sharded_download(rmclhip_pf_sharded* pf, rmclhip_transform* poses, rmclhip_particle_attributes* attrs);
/* PCDSensorUpdater*::update on every device's block (concurrently), then rmclhip_pf_allgather_weights */
rmclhip_status rmclhip_pf_update_sharded(rmclhip_pf_sharded* pf, const rmclhip_range_measurement* beams, uint32_t n_beams,
                                         const rmclhip_transform* Tsb);
/* MotionUpdater<MemT>::update on every device's block (rmcl_localization.cpp:432-480; particle_move_and_forget_kernel,
 * rmcl_ros/src/rmcl/particle_motion.cu:11-46, with check_collision != 0 also the wall-collision ray of TFMotionUpdaterCPU.cpp:17-50,
 * 207-221): the arguments of rmclhip_pf_motion_update, applied to the whole sharded cloud in place -- one launch per device, all
 * enqueued before the host waits for any.  Identical to rmclhip_pf_motion_update on the unsharded cloud (a particle's result depends on
 * that particle alone). */
rmclhip_status rmclhip_pf_sharded_motion_update(rmclhip_pf_sharded* pf, const rmclhip_transform* T_bnew_bold, double forget_rate,
                                                int check_collision);
/* One cycle of the filter node on the sharded cloud (rmcl_localization.cpp:84, 432-552): motion update (T_bnew_bold NULL: skipped) ->
 * sensor update -> weight all-gather -> {sum, max} of the gathered weights on every device (stats_out, nullable; no collective) -> resampling (resample: 0 none, 1 gladiator
 * tournament, 2 residual; config / seed / step as rmclhip_pf_sharded_resample).  A device's motion and sensor-update launches share a
 * stream: the host never waits between them.  Equal, bit for bit, to the single-device sequence rmclhip_pf_motion_update,
 * rmclhip_pf_update, rmclhip_resampler_compute_stats, rmclhip_resampler_gladiator / _residual. */
rmclhip_status rmclhip_pf_sharded_step(rmclhip_pf_sharded* pf, const rmclhip_transform* T_bnew_bold, double forget_rate, int check_collision,
                                       const rmclhip_range_measurement* beams, uint32_t n_beams, const rmclhip_transform* Tsb,
                                       int resample, const rmclhip_gladiator_config* config, uint64_t seed, uint32_t step,
                                       rmclhip_likelihood_stats* stats_out);
/* ONE ncclAllGather of likelihood.mean (4 B x N on equal padded shards): afterwards every device holds the dense weight
 * vector of the whole cloud; rmclhip_pf_sharded_get_weights copies rank's copy to the host */
rmclhip_status rmclhip_pf_allgather_weights(rmclhip_pf_sharded* pf);
rmclhip_status rmclhip_pf_sharded_get_weights(rmclhip_pf_sharded* pf, uint32_t rank, float* weights_host);
/* global {sum, max} (simple_stats_kernel, resampling.cu:41-92).  Since round 6 NOT a collective, the name notwithstanding: after the
 * weight all-gather every device holds all N likelihoods and reduces its own copy with the single-device kernel in that kernel's fixed
 * order -- the single-device value bit for bit, independent of the order in which a collective library would have added per-device
 * partials (the value feeds size_t(L / sum * N) in the residual resampler).  Gathers first if the attributes changed since the last gather. */
rmclhip_status rmclhip_pf_allreduce_stats(rmclhip_pf_sharded* pf, rmclhip_likelihood_stats* out);
/* ranks the communicator's collectives span, as the collective library itself reports them (ncclCommCount; the loopback stand-in: its
 * rank list) and whether it is RCCL (1) or the stand-in (0) */
rmclhip_status rmclhip_comm_collective_ranks(rmclhip_comm* comm, uint32_t* n_ranks, int* is_rccl);
/* RmclNode::estimateStats (rmcl_localization.cpp:642-731) over the first n_induction particles: three passes of per-device
 * moments (<= 24 doubles each: likelihood sums + bounding box, weighted quaternion outer products + translations for the
 * Markley mean, 6x6 covariance around it), each followed by one ncclAllReduce */
rmclhip_status rmclhip_pf_allreduce_pose_estimate(rmclhip_pf_sharded* pf, uint32_t n_induction, rmclhip_pose_estimate* out);
/* distributed GladiatorResamplerGPU::update: all-gather of the 68-B particle records, then every device resamples its own
 * champions against the gathered cloud (Philox counter = GLOBAL champion index => identical to one GPU).  A particle count
 * that is not a multiple of the device count gathers padded shards and squeezes them dense on every device first. */
rmclhip_status rmclhip_pf_sharded_resample(rmclhip_pf_sharded* pf, const rmclhip_gladiator_config* config, uint64_t seed, uint32_t step);
/* the same exchange with the residual resampler (rmclhip_resampler_residual): every device fills its slots of the new cloud from the
 * gathered one -- identical to one GPU (draws and Gaussians are functions of global indices) */
rmclhip_status rmclhip_pf_sharded_resample_residual(rmclhip_pf_sharded* pf, const rmclhip_gladiator_config* config, uint64_t seed, uint32_t step);

/* ---- device memory helpers for hosts without their own allocator ---------------------- */
rmclhip_status rmclhip_malloc(rmclhip_ctx* ctx, size_t bytes, void** out_dev);
rmclhip_status rmclhip_free(rmclhip_ctx* ctx, void* ptr_dev);
rmclhip_status rmclhip_memcpy_h2d(rmclhip_ctx* ctx, void* dst_dev, const void* src, size_t bytes);
rmclhip_status rmclhip_memcpy_d2h(rmclhip_ctx* ctx, void* dst, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* RMCLHIP_H */
