// rmcl_hip.hpp -- header-only C++17 adapters that present librmclhip's C ABI (include/rmclhip.h) with the
// class shapes of the reference, so MICP-L / RMCL callers keep their code:
//
//   rmcl::Correspondences_<MemT>            rmcl/include/rmcl/registration/Correspondences.hpp:16-88
//   rmcl::CorrespondencesCUDA               rmcl/include/rmcl/registration/CorrespondencesCUDA.hpp:10-17
//   rmcl::RCCOptixSpherical / RCCEmbreeO1Dn rmcl/include/rmcl/registration/RCCOptix.hpp:18-93, RCCEmbree.hpp:60-83
//   rmagine::SphereSimulator* / O1Dn / Pinhole / OnDn, Bundle<...>   as used by RCCEmbree.hpp:18-83 (protected bases of the RCC classes),
//                                           rmcl_ros/src/nodes/filter/scan_map_segmentation_embree.cpp:38-39,76-87 and
//                                           rmcl_ros/src/benchmarks/lidar_corrector_embree_benchmark.cpp:117
//   rmagine::statistics_p2l (free function) rmcl/src/rmcl/registration/CorrespondencesCUDA.cpp:28
//   rmcl::SensorUpdater<MemT>               rmcl_ros/include/rmcl_ros/rmcl/SensorUpdater.hpp:18-42
//   rmcl::ParticleUpdater<MemT>::update     rmcl_ros/include/rmcl_ros/rmcl/ParticleUpdater.hpp:24-44
//
// rmagine is not a dependency: the few PODs the hot path needs are restated here with rmagine's member
// names and memory layouts (INTEGRATION.md shows the two-line conversions to/from the rmagine types).
// Errors: the C ABI returns status codes; these adapters rethrow std::runtime_error like the reference
// (micp_localization.cpp:613, PCDSensorUpdaterOptix.cpp:179-192).
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../rmclhip.h"

namespace rmcl_hip {

struct VRAM_HIP {};  // memory-space tag, sibling of rmagine::RAM / VRAM_CUDA

using Vector = rmclhip_vec3;
using Quaternion = rmclhip_quat;
using Transform = rmclhip_transform;
using CrossStatistics = rmclhip_cross_statistics;
using ParticleAttributes = rmclhip_particle_attributes;
using RangeMeasurement = rmclhip_range_measurement;

// rmagine::Interval / DiscreteInterval / SphericalModel: the C ABI's PODs (same members, same layout) with the member functions the
// reference's callers use -- model.range.inside(r), model.getHeight() / getWidth() / getBufferId(vid, hid) / getDirection(vid, hid) /
// getOrigin(vid, hid) / size() (scan_map_segmentation_embree.cpp:108-125, MICPSphericalSensorCPU.cpp:212-219)
struct Interval {
  float min, max;
  bool inside(float v) const { return v >= min && v <= max; }
  operator rmclhip_interval() const { return rmclhip_interval{min, max}; }
};
struct DiscreteInterval {
  float min, inc;
  uint32_t size;
};
struct SphericalModel {
  DiscreteInterval phi;     // vertical: rows, height (conversions.cpp:22-34)
  DiscreteInterval theta;   // horizontal: columns, width
  Interval range;
  uint32_t getWidth() const { return theta.size; }
  uint32_t getHeight() const { return phi.size; }
  size_t size() const { return static_cast<size_t>(phi.size) * theta.size; }
  uint32_t getBufferId(uint32_t vid, uint32_t hid) const { return vid * theta.size + hid; }
  // direction convention pinned by rmcl_ros/src/util/conversions.cpp:174-188
  Vector getDirection(uint32_t vid, uint32_t hid) const {
    const float p = phi.min + static_cast<float>(vid) * phi.inc;
    const float t = theta.min + static_cast<float>(hid) * theta.inc;
    return Vector{std::cos(p) * std::cos(t), std::cos(p) * std::sin(t), std::sin(p)};
  }
  Vector getOrigin(uint32_t, uint32_t) const { return Vector{0.f, 0.f, 0.f}; }
  const rmclhip_spherical_model* c_model() const { return reinterpret_cast<const rmclhip_spherical_model*>(this); }
};
static_assert(sizeof(Interval) == sizeof(rmclhip_interval) && sizeof(DiscreteInterval) == sizeof(rmclhip_discrete_interval) &&
                  sizeof(SphericalModel) == sizeof(rmclhip_spherical_model) && std::is_standard_layout<SphericalModel>::value,
              "SphericalModel must keep the C ABI's layout");

inline void check(rmclhip_status st) {
  if (st != RMCLHIP_OK) throw std::runtime_error(std::string("rmclhip: ") + rmclhip_last_error());
}

inline Transform identity() { return Transform{{0.f, 0.f, 0.f, 1.f}, {0.f, 0.f, 0.f}, 0u}; }

}  // namespace rmcl_hip

// The PODs are the C ABI's structs, which live in the GLOBAL namespace: their operators must live there too, or argument-dependent
// lookup does not find them from a caller's namespace (`namespace rm = rmcl_hip;` as the reference writes `rm::`).
// Transform::operator* / operator~ (micp_localization.cpp:926,963)
inline rmclhip_transform operator*(const rmclhip_transform& a, const rmclhip_transform& b) {
  rmclhip_transform r;
  rmcl_hip::check(rmclhip_transform_mult(&a, &b, &r));
  return r;
}
inline rmclhip_transform operator~(const rmclhip_transform& a) {
  rmclhip_transform r;
  rmcl_hip::check(rmclhip_transform_inv(&a, &r));
  return r;
}
// Transform * CrossStatistics (MICPSensor.hpp:182)
inline rmclhip_cross_statistics operator*(const rmclhip_transform& T, const rmclhip_cross_statistics& s) {
  rmclhip_cross_statistics r;
  rmcl_hip::check(rmclhip_cross_statistics_transform(&T, &s, &r));
  return r;
}
// CrossStatistics::operator+= (micp_localization.cpp:936-937)
inline rmclhip_cross_statistics& operator+=(rmclhip_cross_statistics& a, const rmclhip_cross_statistics& b) {
  rmclhip_cross_statistics r;
  rmcl_hip::check(rmclhip_cross_statistics_merge(&a, &b, &r));
  a = r;
  return a;
}
// rmagine::Vector arithmetic the callers of simulate() use on its results (scan_map_segmentation_embree.cpp:125-135)
inline rmclhip_vec3 operator*(const rmclhip_vec3& a, float s) { return rmclhip_vec3{a.x * s, a.y * s, a.z * s}; }
inline rmclhip_vec3 operator+(const rmclhip_vec3& a, const rmclhip_vec3& b) { return rmclhip_vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline rmclhip_vec3 operator-(const rmclhip_vec3& a, const rmclhip_vec3& b) { return rmclhip_vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }

namespace rmcl_hip {

inline float dot(const Vector& a, const Vector& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float l2norm(const Vector& a) { return std::sqrt(dot(a, a)); }

inline CrossStatistics cross_statistics_identity() { return CrossStatistics{}; }
// rm::umeyama_transform (micp_localization.cpp:952-953)
inline Transform umeyama_transform(const CrossStatistics& s) {
  Transform r;
  check(rmclhip_umeyama_transform(&s, &r));
  return r;
}

struct UmeyamaReductionConstraints { float max_dist = 1.0f; };

// one HIP device
class Context {
 public:
  explicit Context(int device = 0) { check(rmclhip_ctx_create(device, &h_)); }
  ~Context() { rmclhip_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  rmclhip_ctx* handle() const { return h_; }

 private:
  rmclhip_ctx* h_ = nullptr;
};
using ContextPtr = std::shared_ptr<Context>;

// rm::EmbreeMap / rm::OptixMap analogue; shared via shared_ptr like EmbreeMapPtr (PCDSensorUpdaterEmbree.cpp:143-174)
class HipMap {
 public:
  HipMap(ContextPtr ctx, const float* vertices_xyz, uint32_t n_vertices, const uint32_t* faces_ijk, uint32_t n_faces)
      : ctx_(std::move(ctx)) {
    check(rmclhip_map_create(ctx_->handle(), vertices_xyz, n_vertices, faces_ijk, n_faces, &h_));
  }
  // a whole scene -- what rm::import_embree_map / import_optix_map make of an assimp file (micp_localization.cpp:187-195):
  // meshes placed (and possibly repeated) by instances; no instances = every mesh once, untransformed.  Face ids of find()
  // are global, sceneInstances() / locate() translate them back.
  HipMap(ContextPtr ctx, const std::vector<rmclhip_mesh>& meshes, const std::vector<rmclhip_instance>& instances = {})
      : ctx_(std::move(ctx)) {
    check(rmclhip_map_create_scene(ctx_->handle(), meshes.data(), static_cast<uint32_t>(meshes.size()),
                                   instances.empty() ? nullptr : instances.data(), static_cast<uint32_t>(instances.size()), &h_));
  }
  // rmagine places an instance with a Transform and a per-axis scale (A = R diag(scale), t): the 3x4 the C ABI takes
  static rmclhip_instance instance(uint32_t mesh, const Transform& T, const Vector& scale = Vector{1.f, 1.f, 1.f}) {
    const float x = T.R.x, y = T.R.y, z = T.R.z, w = T.R.w;
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - z * w), 2.f * (x * z + y * w)},
                           {2.f * (x * y + z * w), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - x * w)},
                           {2.f * (x * z - y * w), 2.f * (y * z + x * w), 1.f - 2.f * (x * x + y * y)}};
    const float sc[3] = {scale.x, scale.y, scale.z}, t[3] = {T.t.x, T.t.y, T.t.z};
    rmclhip_instance I{};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) I.transform[4 * r + c] = R[r][c] * sc[c];
      I.transform[4 * r + 3] = t[r];
    }
    I.mesh = mesh;
    return I;
  }
  // first global face id of every instance, + n_faces as the last entry
  std::vector<uint32_t> sceneInstances() const {
    uint32_t n = 0;
    check(rmclhip_map_scene_instances(h_, nullptr, 0, &n));
    std::vector<uint32_t> first(n + 1u);
    check(rmclhip_map_scene_instances(h_, first.data(), first.size(), &n));
    return first;
  }
  struct FaceLocation { uint32_t instance, face; };
  FaceLocation locate(uint32_t face_id) const {
    FaceLocation L{};
    check(rmclhip_map_scene_locate(h_, face_id, &L.instance, &L.face));
    return L;
  }
  ~HipMap() { rmclhip_map_release(h_); }
  HipMap(const HipMap&) = delete;
  HipMap& operator=(const HipMap&) = delete;
  rmclhip_map* handle() const { return h_; }
  const ContextPtr& context() const { return ctx_; }

 private:
  ContextPtr ctx_;
  rmclhip_map* h_ = nullptr;
};
using HipMapPtr = std::shared_ptr<HipMap>;

// non-owning device view (rmagine::MemoryView<T, VRAM_HIP>)
template <typename T>
struct DeviceView {
  T* ptr = nullptr;
  size_t n = 0;
  T* raw() const { return ptr; }
  size_t size() const { return n; }
};

// rmagine::Memory<T, MemT> for the two memory spaces the adapters need (rmcl_localization.cpp:408-421 uses resize / raw /
// size / operator[] / cross-space assignment).  RAM = host, VRAM_HIP = device memory of one context.
struct RAM {};
template <typename T, typename MemT>
class Memory;

// rmagine::MemoryView<T, RAM>: non-owning host view (`const rm::MemoryView<float, rm::RAM> ranges = res.ranges;`,
// scan_map_segmentation_embree.cpp:89-90); MemoryView<T, VRAM_HIP> is DeviceView<T>
template <typename T, typename MemT>
struct MemoryView;
template <typename T>
struct MemoryView<T, RAM> {
  T* ptr = nullptr;
  size_t n = 0;
  MemoryView() = default;
  MemoryView(T* p, size_t count) : ptr(p), n(count) {}
  template <typename U, typename = typename std::enable_if<std::is_same<typename std::remove_const<T>::type, U>::value>::type>
  MemoryView(const Memory<U, RAM>& m) : ptr(const_cast<T*>(m.raw())), n(m.size()) {}   // (T is const-qualified for a const Memory)
  T* raw() const { return ptr; }
  size_t size() const { return n; }
  T& operator[](size_t i) const { return ptr[i]; }
};
template <typename T>
struct MemoryView<T, VRAM_HIP> : DeviceView<T> {};

template <typename T>
class Memory<T, RAM> {
 public:
  Memory() = default;
  explicit Memory(size_t n) : v_(n) {}
  void resize(size_t n) { v_.resize(n); }
  size_t size() const { return v_.size(); }
  T* raw() { return v_.data(); }
  const T* raw() const { return v_.data(); }
  T& operator[](size_t i) { return v_[i]; }
  const T& operator[](size_t i) const { return v_[i]; }

 private:
  std::vector<T> v_;
};

template <typename T>
class Memory<T, VRAM_HIP> {
 public:
  Memory() = default;
  explicit Memory(ContextPtr ctx) : ctx_(std::move(ctx)) {}
  ~Memory() { release(); }
  Memory(const Memory&) = delete;
  Memory& operator=(const Memory&) = delete;
  // movable: `ResultT res = sim->simulate<ResultT>(T)` returns a bundle of these by value (scan_map_segmentation_embree.cpp:87)
  Memory(Memory&& o) noexcept : ctx_(std::move(o.ctx_)), ptr_(o.ptr_), cap_(o.cap_), n_(o.n_), version_(o.version_ + 1u) {
    o.ptr_ = nullptr;
    o.cap_ = o.n_ = 0;
  }
  Memory& operator=(Memory&& o) noexcept {
    if (this != &o) {
      release();
      ctx_ = std::move(o.ctx_);
      ptr_ = o.ptr_; cap_ = o.cap_; n_ = o.n_;
      o.ptr_ = nullptr;
      o.cap_ = o.n_ = 0;
      ++version_;
    }
    return *this;
  }
  void setContext(ContextPtr ctx) { ctx_ = std::move(ctx); }
  const ContextPtr& context() const { return ctx_; }
  // grow-only like rm::Memory::resize on the reference's device paths (contents are not preserved)
  void resize(size_t n) {
    if (n > cap_) {
      release();
      if (!ctx_) throw std::runtime_error("Memory<VRAM_HIP>: no context");
      void* p = nullptr;
      check(rmclhip_malloc(ctx_->handle(), n * sizeof(T), &p));
      ptr_ = static_cast<T*>(p);
      cap_ = n;
      ++version_;   // the buffer moved: an operator that borrowed the old pointer rebinds before its next use
    }
    n_ = n;
  }
  size_t size() const { return n_; }
  T* raw() { return ptr_; }
  const T* raw() const { return ptr_; }
  // cross-space assignment: `correspondences_->dataset.points = dataset_cpu_.points` (MICPSphericalSensorCUDA.cpp:231)
  Memory& operator=(const Memory<T, RAM>& host) {
    resize(host.size());
    if (host.size()) check(rmclhip_memcpy_h2d(ctx_->handle(), ptr_, host.raw(), host.size() * sizeof(T)));
    ++version_;
    return *this;
  }
  void download(Memory<T, RAM>& host) const {
    host.resize(n_);
    if (n_) check(rmclhip_memcpy_d2h(ctx_->handle(), host.raw(), ptr_, n_ * sizeof(T)));
  }
  uint64_t version() const { return version_; }   // bumped by every upload (the operator rebinds its dataset view)
  void touch() { ++version_; }                    // after writing through raw() with one's own kernel

 private:
  void release() {
    if (ptr_ && ctx_) (void)rmclhip_free(ctx_->handle(), ptr_);
    ptr_ = nullptr;
    cap_ = n_ = 0;
  }
  ContextPtr ctx_;
  T* ptr_ = nullptr;
  size_t cap_ = 0, n_ = 0;
  uint64_t version_ = 0;
};

// device view -> host memory (rmagine: `Memory<T, RAM> host = view;`), e.g. the members of modelView()
template <typename T>
inline void download(const Context& ctx, const DeviceView<const T>& view, Memory<T, RAM>& host) {
  host.resize(view.size());
  if (view.size()) check(rmclhip_memcpy_d2h(ctx.handle(), host.raw(), view.raw(), view.size() * sizeof(T)));
}

// rmagine::PointCloud_<MemT> / PointCloudView_<MemT> (Correspondences.hpp:24,47-62): {points, mask} (+ normals in views)
template <typename MemT>
struct PointCloud_ {
  Memory<Vector, MemT> points;
  Memory<uint8_t, MemT> mask;
};
template <typename MemT>
struct PointCloudView_ {
  DeviceView<const Vector> points;
  DeviceView<const uint8_t> mask;
  DeviceView<const Vector> normals;
  rmclhip_ctx* ctx = nullptr;   // the device the views live on (filled by watch() / modelView() / datasetView()): statistics_p2l runs there
};
// rm::watch(dataset) (CorrespondencesCUDA.cpp:13): a view of an owning point cloud
inline PointCloudView_<VRAM_HIP> watch(const PointCloud_<VRAM_HIP>& cloud) {
  PointCloudView_<VRAM_HIP> v;
  v.points = {cloud.points.raw(), cloud.points.size()};
  v.mask = {cloud.mask.raw(), cloud.mask.size()};
  v.ctx = cloud.points.context() ? cloud.points.context()->handle() : nullptr;
  return v;
}
// rm::statistics_p2l(Tpre, dataset, model, params) -> CrossStatistics on device views (CorrespondencesCUDA.cpp:28; gate and projection
// as MICPSensorCPU.cpp:70-84): what CorrespondencesCUDA::computeCrossStatistics calls after its max_dist interpolation.  A view
// without a mask counts every element as valid; the views must live on one device.
inline CrossStatistics statistics_p2l(const Transform& Tpre, const PointCloudView_<VRAM_HIP>& dataset, const PointCloudView_<VRAM_HIP>& model,
                                      const UmeyamaReductionConstraints& params) {
  rmclhip_ctx* ctx = dataset.ctx ? dataset.ctx : model.ctx;
  if (!ctx) throw std::runtime_error("statistics_p2l: views without a context (use watch() / modelView())");
  if (dataset.ctx && model.ctx && dataset.ctx != model.ctx) throw std::runtime_error("statistics_p2l: dataset and model live on different contexts");
  if (model.normals.size() < model.points.size()) throw std::runtime_error("statistics_p2l: the model view needs normals");
  if ((dataset.mask.size() && dataset.mask.size() < dataset.points.size()) || (model.mask.size() && model.mask.size() < model.points.size()))
    throw std::runtime_error("statistics_p2l: mask shorter than its points");
  const size_t n = dataset.points.size() < model.points.size() ? dataset.points.size() : model.points.size();
  CrossStatistics out;
  check(rmclhip_statistics_p2l(ctx, &Tpre, reinterpret_cast<const float*>(dataset.points.raw()), dataset.mask.size() ? dataset.mask.raw() : nullptr,
                               reinterpret_cast<const float*>(model.points.raw()), reinterpret_cast<const float*>(model.normals.raw()),
                               model.mask.size() ? model.mask.raw() : nullptr, static_cast<uint32_t>(n), params.max_dist, &out));
  return out;
}

// ---- rmagine::Bundle and its attributes (rmagine/simulation/SimulationResults.hpp as used by Correspondences.hpp:81-85,
// scan_map_segmentation_embree.cpp:82-85, lidar_corrector_embree_benchmark.cpp:95-100) -------------------------------------------
// An attribute is a struct with ONE Memory member of rmagine's name; a Bundle derives from the attributes it carries; simulate()
// writes exactly those.  MemT = VRAM_HIP: the kernel writes the caller's device memory; MemT = RAM: results are staged on the device
// and copied to the host (the Embree callers' code, unchanged).
template <typename MemT> struct Hits { Memory<uint8_t, MemT> hits; Memory<uint8_t, MemT>& attr_memory_() { return hits; } };
template <typename MemT> struct Ranges { Memory<float, MemT> ranges; Memory<float, MemT>& attr_memory_() { return ranges; } };
template <typename MemT> struct Points { Memory<Vector, MemT> points; Memory<Vector, MemT>& attr_memory_() { return points; } };
template <typename MemT> struct Normals { Memory<Vector, MemT> normals; Memory<Vector, MemT>& attr_memory_() { return normals; } };
template <typename MemT> struct FaceIds { Memory<uint32_t, MemT> face_ids; Memory<uint32_t, MemT>& attr_memory_() { return face_ids; } };
template <typename... Attrs>
struct Bundle : public Attrs... {};

namespace detail {
template <template <typename> class Attr, typename BundleT>
constexpr bool has_attr() { return std::is_base_of<Attr<RAM>, BundleT>::value || std::is_base_of<Attr<VRAM_HIP>, BundleT>::value; }
// resize one attribute of a bundle if it carries it in MemT
template <typename MemT, template <typename> class Attr, typename BundleT>
inline void resize_attr(BundleT& b, size_t n, const ContextPtr& ctx) {
  if constexpr (std::is_base_of<Attr<MemT>, BundleT>::value) {
    auto& m = static_cast<Attr<MemT>&>(b).attr_memory_();
    if constexpr (std::is_same<MemT, VRAM_HIP>::value) { if (!m.context()) m.setContext(ctx); }
    m.resize(n);
  }
}
}  // namespace detail
// rm::resize_memory_bundle<MemT>(bundle, H, W, N) (RCCEmbree.cpp:32): every attribute the bundle carries in MemT gets H * W * N elements
// (device attributes without a context yet take `ctx`)
template <typename MemT, typename BundleT>
inline void resize_memory_bundle(BundleT& b, size_t H, size_t W, size_t N, const ContextPtr& ctx = nullptr) {
  const size_t n = H * W * N;
  detail::resize_attr<MemT, Hits>(b, n, ctx);
  detail::resize_attr<MemT, Ranges>(b, n, ctx);
  detail::resize_attr<MemT, Points>(b, n, ctx);
  detail::resize_attr<MemT, Normals>(b, n, ctx);
  detail::resize_attr<MemT, FaceIds>(b, n, ctx);
}

template <typename MemT>
class Correspondences_;

// rmcl::Correspondences_<VRAM_HIP> + CorrespondencesCUDA::computeCrossStatistics
template <>
class Correspondences_<VRAM_HIP> {
 public:
  // public attributes that have to be filled (Correspondences.hpp:19-29)
  UmeyamaReductionConstraints params;
  float adaptive_max_dist_min = 1.0f;
  // Correspondences.hpp:24: sensors write `dataset.points = host_points; dataset.mask = host_mask;` (device memory owned
  // here, borrowed by the library: rmclhip_rcc_set_dataset_view) -- reference sensor code compiles against this unchanged
  PointCloud_<VRAM_HIP> dataset;
  bool outdated = true;

  explicit Correspondences_(HipMapPtr map) : map_(std::move(map)) {
    if (!map_) throw std::runtime_error("NO MAP");
    check(rmclhip_rcc_create(map_->context()->handle(), map_->handle(), &h_));
    dataset.points.setContext(map_->context());
    dataset.mask.setContext(map_->context());
  }
  virtual ~Correspondences_() { rmclhip_rcc_destroy(h_); }
  Correspondences_(const Correspondences_&) = delete;
  Correspondences_& operator=(const Correspondences_&) = delete;

  virtual void setTsb(const Transform& Tsb) {
    Tsb_ = Tsb;
    check(rmclhip_rcc_set_tsb(h_, &Tsb));
  }
  // finds and fills the model buffers
  // (binds `dataset` and pushes `params` first: since round 4 a find that is followed by computeCrossStatistics calls forms the
  // moments of ITS correspondences in its epilogue -- it reads the dataset, so a re-assigned `dataset` member must be bound before)
  virtual void find(const Transform& Tbm_est) {
    bindDataset();
    check(rmclhip_rcc_set_params(h_, params.max_dist, adaptive_max_dist_min));
    check(rmclhip_rcc_find(h_, &Tbm_est));
  }
  virtual CrossStatistics computeCrossStatistics(const Transform& T_snew_sold, double convergence_progress = 0.0) const {
    bindDataset();
    check(rmclhip_rcc_set_params(h_, params.max_dist, adaptive_max_dist_min));
    CrossStatistics out;
    check(rmclhip_rcc_compute_cross_statistics(h_, &T_snew_sold, convergence_progress, &out));
    return out;
  }
  // dataset {points, mask}: host or device source (the CUDA sensors upload once per scan,
  // MICPSphericalSensorCUDA.cpp:231-232)
  void setDataset(const float* points_xyz, const uint8_t* mask, uint32_t n, bool src_is_device = false) {
    check(rmclhip_rcc_set_dataset(h_, points_xyz, mask, n, src_is_device ? 1 : 0));
    bound_points_ = bound_mask_ = ~0ull;   // the library's own copy is current until `dataset` is written again
    outdated = true;
  }
  uint32_t setDatasetFromRanges(const float* ranges, uint32_t n) {
    uint32_t valid = 0;
    check(rmclhip_rcc_set_dataset_from_ranges(h_, ranges, n, &valid));
    bound_points_ = bound_mask_ = ~0ull;
    outdated = true;
    return valid;
  }
  // datasetView() (Correspondences.hpp:55-62)
  PointCloudView_<VRAM_HIP> datasetView() const {
    PointCloudView_<VRAM_HIP> v;
    v.points = {dataset.points.raw(), dataset.points.size()};
    v.mask = {dataset.mask.raw(), dataset.mask.size()};
    v.ctx = map_->context()->handle();
    return v;
  }
  // Bundle attribute selection of find() (rmclhip.h: rmclhip_rcc_set_outputs).  The C ABI's default is all five; the reference's
  // model_buffers_ is Bundle<Points, Normals, Hits> (Correspondences.hpp:81-85) = RMCLHIP_OUT_MICP, 25 instead of 33 B per ray.
  void setOutputs(uint32_t mask) { check(rmclhip_rcc_set_outputs(h_, mask)); }
  uint32_t outputs() const {
    uint32_t m = 0;
    check(rmclhip_rcc_get_outputs(h_, &m));
    return m;
  }
  // measurement-driven choice of the single-scan traversal for this map and model (rmclhip.h: rmclhip_rcc_autotune); returns the kind
  int autotune(const Transform& Tbm_est) {
    int kind = 0;
    check(rmclhip_rcc_autotune(h_, &Tbm_est, &kind, nullptr));
    return kind;
  }
  int autotuneBatch(const std::vector<Transform>& Tbm) {
    int kind = 0;
    check(rmclhip_rcc_autotune_batch(h_, Tbm.data(), static_cast<uint32_t>(Tbm.size()), &kind, nullptr));
    return kind;
  }
  // modelView() (Correspondences.hpp:47-53): PointCloudView_ {points, mask = hits, normals} over the model buffers of the last
  // find -- the shape MICPSensorCUDA.cpp:66-84 reads
  PointCloudView_<VRAM_HIP> modelView() const {
    const ModelBuffers b = modelBuffers();
    PointCloudView_<VRAM_HIP> v;
    v.points = {reinterpret_cast<const Vector*>(b.points), b.points ? b.n : 0u};
    v.mask = {b.mask, b.mask ? b.n : 0u};
    v.normals = {reinterpret_cast<const Vector*>(b.normals), b.normals ? b.n : 0u};
    v.ctx = map_->context()->handle();
    return v;
  }
  // everything find() writes, as raw borrowed device pointers (+ ranges, face ids, which the reference's bundle does not carry)
  struct ModelBuffers {
    const float* points;
    const uint8_t* mask;
    const float* normals;
    const float* ranges;
    const uint32_t* face_ids;
    uint32_t n;
  };
  ModelBuffers modelBuffers() const {
    ModelBuffers v{};
    check(rmclhip_rcc_device_views(h_, &v.mask, &v.ranges, &v.points, &v.normals, &v.face_ids, &v.n));
    return v;
  }
  void download(uint8_t* hits, float* ranges, float* points, float* normals, uint32_t* face_ids) const {
    check(rmclhip_rcc_download(h_, hits, ranges, points, normals, face_ids));
  }
  // device-resident MICP-L inner loop for this sensor (micp_localization.cpp:900-964)
  Transform correctOnce(const Transform& Tom, const Transform& Tbo, uint32_t iterations, double convergence_progress,
                        bool refind_each_iteration, CrossStatistics* stats_o = nullptr) {
    bindDataset();
    check(rmclhip_rcc_set_params(h_, params.max_dist, adaptive_max_dist_min));
    Transform T;
    check(rmclhip_rcc_correct_once(h_, &Tom, &Tbo, iterations, convergence_progress, refind_each_iteration ? 1 : 0, &T,
                                   stats_o));
    return T;
  }
  // v1 SphereCorrector::correct (lidar_corrector_embree_benchmark.cpp:127-135)
  std::vector<Transform> correctBatch(const std::vector<Transform>& Tbm, std::vector<CrossStatistics>* stats = nullptr) {
    bindDataset();
    check(rmclhip_rcc_set_params(h_, params.max_dist, adaptive_max_dist_min));
    std::vector<Transform> out(Tbm.size());
    if (stats) stats->resize(Tbm.size());
    check(rmclhip_rcc_correct_batch(h_, Tbm.data(), static_cast<uint32_t>(Tbm.size()), out.data(),
                                    stats ? stats->data() : nullptr));
    return out;
  }
  rmclhip_rcc* handle() const { return h_; }
  // moment form of correctOnce's fixed-correspondence loop (rmclhip.h: rmclhip_rcc_set_micp_fast): 0 never, 1 automatic
  void setMicpFast(int mode) { check(rmclhip_rcc_set_micp_fast(h_, mode)); }
  rmclhip_micp_fast_info micpFastInfo() const {
    rmclhip_micp_fast_info info{};
    check(rmclhip_rcc_micp_fast_info(h_, &info));
    return info;
  }
  // how computeCrossStatistics was served: from the find's published moments on the host (no launch) or by a streaming reduction
  rmclhip_ccs_info ccsInfo() const {
    rmclhip_ccs_info info{};
    check(rmclhip_rcc_ccs_info(h_, &info));
    return info;
  }
  // bind `dataset` and push `params` before a device-resident loop reads them (correctOnce for several sensors)
  void prepareForDeviceLoop() {
    bindDataset();
    check(rmclhip_rcc_set_params(h_, params.max_dist, adaptive_max_dist_min));
    outdated = false;
  }

 protected:
  // hand the current `dataset` memory to the library when it was (re)written since the last call
  void bindDataset() const {
    if (dataset.points.size() == 0) return;   // dataset handed over with setDataset*() instead
    if (dataset.points.version() == bound_points_ && dataset.mask.version() == bound_mask_) return;
    if (dataset.mask.size() != 0 && dataset.mask.size() != dataset.points.size())
      throw std::runtime_error("Correspondences: dataset.mask.size() != dataset.points.size()");
    check(rmclhip_rcc_set_dataset_view(h_, reinterpret_cast<const float*>(dataset.points.raw()),
                                       dataset.mask.size() ? dataset.mask.raw() : nullptr,
                                       static_cast<uint32_t>(dataset.points.size())));
    bound_points_ = dataset.points.version();
    bound_mask_ = dataset.mask.version();
  }
  HipMapPtr map_;
  rmclhip_rcc* h_ = nullptr;
  Transform Tsb_ = identity();
  mutable uint64_t bound_points_ = ~0ull, bound_mask_ = ~0ull;
};
using CorrespondencesHIP = Correspondences_<VRAM_HIP>;

// MICPLocalizationNode::correctOnce inner loop (micp_localization.cpp:900-964) for all sensors of the node, resident on the
// device: `sensors[i]` is the correspondence operator of sensor i, Tbo[i] its odom frame at its stamp, weights[i] its
// merge_weight_multiplier.  Returns T_onew_oold; merged (optional) receives the unweighted Cmerged_o of the last iteration.
inline Transform correctOnce(const std::vector<CorrespondencesHIP*>& sensors, const Transform& Tom, const std::vector<Transform>& Tbo,
                             const std::vector<double>& weights, uint32_t iterations, double convergence_progress,
                             CrossStatistics* merged = nullptr) {
  if (sensors.empty() || Tbo.size() != sensors.size() || (!weights.empty() && weights.size() != sensors.size()))
    throw std::runtime_error("correctOnce: one Tbo (and weight) per sensor");
  std::vector<rmclhip_rcc*> h(sensors.size());
  for (size_t i = 0; i < sensors.size(); ++i) {
    sensors[i]->prepareForDeviceLoop();
    h[i] = sensors[i]->handle();
  }
  Transform T;
  check(rmclhip_micp_correct_once(h.data(), static_cast<uint32_t>(h.size()), &Tom, Tbo.data(), weights.empty() ? nullptr : weights.data(),
                                  iterations, convergence_progress, &T, merged));
  return T;
}

// rmagine::ModelSetter<ModelT>
template <typename ModelT>
struct ModelSetter {
  virtual ~ModelSetter() = default;
  virtual void setModel(const ModelT&) = 0;
};

// rmagine::O1DnModel (fields: rmcl_ros/src/util/conversions.cpp:74-94; buffer id = vid * width + hid: conversions.cpp:974)
struct O1DnModel {
  uint32_t width = 0, height = 0;
  Interval range{0.f, 0.f};
  Vector orig{0.f, 0.f, 0.f};
  std::vector<Vector> dirs;
  uint32_t getWidth() const { return width; }
  uint32_t getHeight() const { return height; }
  size_t size() const { return static_cast<size_t>(width) * height; }
  uint32_t getBufferId(uint32_t vid, uint32_t hid) const { return vid * width + hid; }
  Vector getDirection(uint32_t vid, uint32_t hid) const { return dirs[getBufferId(vid, hid)]; }
  Vector getOrigin(uint32_t, uint32_t) const { return orig; }
};

// rmagine::SphericalModel::getDirection / getBufferId (convention pinned by rmcl_ros/src/util/conversions.cpp:174-188):
// what the reference's unpackMessage evaluates per measurement when it fills `dataset` (MICPSphericalSensorCPU.cpp:212-219)
inline Vector getDirection(const SphericalModel& m, uint32_t vid, uint32_t hid) { return m.getDirection(vid, hid); }
inline uint32_t getBufferId(const SphericalModel& m, uint32_t vid, uint32_t hid) { return m.getBufferId(vid, hid); }

// rmagine::PinholeModel (fields: rmcl_ros/src/util/conversions.cpp:36-60)
struct PinholeModel {
  uint32_t width = 0, height = 0;
  Interval range{0.f, 0.f};
  float f[2] = {1.f, 1.f};
  float c[2] = {0.f, 0.f};
  uint32_t getWidth() const { return width; }
  uint32_t getHeight() const { return height; }
  size_t size() const { return static_cast<size_t>(width) * height; }
  uint32_t getBufferId(uint32_t vid, uint32_t hid) const { return vid * width + hid; }
};

// rmagine::OnDnModel (fields: rmcl_ros/src/util/conversions.cpp:96-120)
struct OnDnModel {
  uint32_t width = 0, height = 0;
  Interval range{0.f, 0.f};
  std::vector<Vector> origs, dirs;
  uint32_t getWidth() const { return width; }
  uint32_t getHeight() const { return height; }
  size_t size() const { return static_cast<size_t>(width) * height; }
  uint32_t getBufferId(uint32_t vid, uint32_t hid) const { return vid * width + hid; }
  Vector getDirection(uint32_t vid, uint32_t hid) const { return dirs[getBufferId(vid, hid)]; }
  Vector getOrigin(uint32_t vid, uint32_t hid) const { return origs[getBufferId(vid, hid)]; }
};

namespace detail {
// Simulator::setModel per model type -> the C ABI's setter
inline void set_model(rmclhip_rcc* h, const SphericalModel& m) { check(rmclhip_rcc_set_model_spherical(h, m.c_model())); }
inline void set_model(rmclhip_rcc* h, const O1DnModel& m) {
  if (m.dirs.size() != m.size()) throw std::runtime_error("O1DnModel: dirs.size() != width*height");
  check(rmclhip_rcc_set_model_o1dn(h, m.width, m.height, m.range, m.orig, reinterpret_cast<const float*>(m.dirs.data())));
}
inline void set_model(rmclhip_rcc* h, const PinholeModel& m) {
  check(rmclhip_rcc_set_model_pinhole(h, m.width, m.height, m.range, m.f[0], m.f[1], m.c[0], m.c[1]));
}
inline void set_model(rmclhip_rcc* h, const OnDnModel& m) {
  if (m.dirs.size() != m.size() || m.origs.size() != m.size()) throw std::runtime_error("OnDnModel: origs/dirs size != width*height");
  check(rmclhip_rcc_set_model_ondn(h, m.width, m.height, m.range, reinterpret_cast<const float*>(m.origs.data()),
                                   reinterpret_cast<const float*>(m.dirs.data())));
}
}  // namespace detail

// rmagine::SphereSimulatorEmbree / O1DnSimulatorEmbree / PinholeSimulatorEmbree / OnDnSimulatorEmbree (and their Optix twins) on gfx950:
//   setTsb, setModel, simulate(const Transform&, BundleT&), simulate<BundleT>(Transform), simulate(Memory<Transform>&, BundleT&)
// (scan_map_segmentation_embree.cpp:38-39,78,87; lidar_corrector_embree_benchmark.cpp:117; lidar_corrector_optix_benchmark.cpp:119 with
// the poses in device memory).  Results are in the SENSOR frame; a bundle receives exactly the attributes it carries
// (rmclhip_rcc_simulate): the kernel skips the stores of the others.  Standalone (`SphereSimulatorHip sim(map)`) it owns its engine
// handle; as the protected base of an RCCHip* class it borrows the operator's (RCCEmbree.hpp:18-22).
template <typename ModelT>
class SimulatorHip {
 public:
  explicit SimulatorHip(HipMapPtr map) : sim_map_(std::move(map)), sim_owns_(true) {
    if (!sim_map_) throw std::runtime_error("NO MAP");
    check(rmclhip_rcc_create(sim_map_->context()->handle(), sim_map_->handle(), &sim_));
  }
  virtual ~SimulatorHip() {
    if (sim_owns_) rmclhip_rcc_destroy(sim_);
  }
  SimulatorHip(const SimulatorHip&) = delete;
  SimulatorHip& operator=(const SimulatorHip&) = delete;

  void setTsb(const Transform& Tsb) { check(rmclhip_rcc_set_tsb(sim_, &Tsb)); }
  void setModel(const ModelT& model) {
    detail::set_model(sim_, model);
    m_model = std::make_shared<ModelT>(model);
  }
  // rmagine keeps the model as `m_model` (RCCEmbree.cpp:29: m_model->size())
  std::shared_ptr<ModelT> model() const { return m_model; }

  // simulate(Tbm, res): every attribute of `res` must already hold model.size() elements (lidar_corrector_embree_benchmark.cpp:99-100)
  template <typename BundleT>
  void simulate(const Transform& Tbm, BundleT& res) { run(&Tbm, 1u, false, res); }
  // simulate<BundleT>(Tbm): allocates the bundle (scan_map_segmentation_embree.cpp:87)
  template <typename BundleT>
  BundleT simulate(const Transform& Tbm) {
    BundleT res;
    resize_bundle(res, 1u);
    run(&Tbm, 1u, false, res);
    return res;
  }
  // batch forms: results pose-major, index pose * size() + getBufferId(vid, hid)
  template <typename BundleT>
  void simulate(const Memory<Transform, RAM>& Tbm, BundleT& res) { run(Tbm.raw(), static_cast<uint32_t>(Tbm.size()), false, res); }
  template <typename BundleT>
  void simulate(const Memory<Transform, VRAM_HIP>& Tbm, BundleT& res) { run(Tbm.raw(), static_cast<uint32_t>(Tbm.size()), true, res); }
  template <typename BundleT>
  BundleT simulate(const Memory<Transform, RAM>& Tbm) {
    BundleT res;
    resize_bundle(res, Tbm.size());
    run(Tbm.raw(), static_cast<uint32_t>(Tbm.size()), false, res);
    return res;
  }
  template <typename BundleT>
  BundleT simulate(const Memory<Transform, VRAM_HIP>& Tbm) {
    BundleT res;
    resize_bundle(res, Tbm.size());
    run(Tbm.raw(), static_cast<uint32_t>(Tbm.size()), true, res);
    return res;
  }

 protected:
  // the protected-base form: the engine is the operator's handle
  struct Borrowed { rmclhip_rcc* h; HipMapPtr map; };
  explicit SimulatorHip(Borrowed b) : sim_map_(std::move(b.map)), sim_(b.h), sim_owns_(false) {}
  std::shared_ptr<ModelT> m_model;

 private:
  template <typename BundleT>
  void resize_bundle(BundleT& res, size_t nposes) {
    if (!m_model) throw std::runtime_error("simulate: no sensor model (setModel first)");
    resize_memory_bundle<RAM>(res, m_model->getHeight(), m_model->getWidth(), nposes);
    resize_memory_bundle<VRAM_HIP>(res, m_model->getHeight(), m_model->getWidth(), nposes, sim_map_->context());
  }
  // one attribute: the device pointer the kernel writes -- the caller's own memory (VRAM_HIP) or this simulator's staging buffer (RAM)
  template <template <typename> class Attr, typename T, typename BundleT>
  T* bind(BundleT& res, size_t n, Memory<T, VRAM_HIP>& stage) {
    if constexpr (std::is_base_of<Attr<VRAM_HIP>, BundleT>::value) {
      auto& m = static_cast<Attr<VRAM_HIP>&>(res).attr_memory_();
      if (m.size() < n) throw std::runtime_error("simulate: a bundle attribute is smaller than poses * model.size()");
      return m.raw();
    } else if constexpr (std::is_base_of<Attr<RAM>, BundleT>::value) {
      if (static_cast<Attr<RAM>&>(res).attr_memory_().size() < n) throw std::runtime_error("simulate: a bundle attribute is smaller than poses * model.size()");
      if (!stage.context()) stage.setContext(sim_map_->context());
      stage.resize(n);
      return stage.raw();
    } else {
      (void)res; (void)n; (void)stage;
      return nullptr;
    }
  }
  template <template <typename> class Attr, typename T, typename BundleT>
  void fetch(BundleT& res, size_t n, const Memory<T, VRAM_HIP>& stage) {
    if constexpr (!std::is_base_of<Attr<VRAM_HIP>, BundleT>::value && std::is_base_of<Attr<RAM>, BundleT>::value) {
      if (n) check(rmclhip_memcpy_d2h(sim_map_->context()->handle(), static_cast<Attr<RAM>&>(res).attr_memory_().raw(), stage.raw(), n * sizeof(T)));
    } else {
      (void)res; (void)n; (void)stage;
    }
  }
  template <typename BundleT>
  void run(const Transform* Tbm, uint32_t nposes, bool poses_on_device, BundleT& res) {
    static_assert(detail::has_attr<Hits, BundleT>() || detail::has_attr<Ranges, BundleT>() || detail::has_attr<Points, BundleT>() ||
                      detail::has_attr<Normals, BundleT>() || detail::has_attr<FaceIds, BundleT>(),
                  "simulate: the bundle carries none of Hits / Ranges / Points / Normals / FaceIds");
    if (!m_model) throw std::runtime_error("simulate: no sensor model (setModel first)");
    if (nposes == 0u) return;
    const size_t n = m_model->size() * nposes;
    rmclhip_bundle_views v{};
    v.hits_dev = bind<Hits>(res, n, st_hits_);
    v.ranges_dev = bind<Ranges>(res, n, st_ranges_);
    v.points_xyz_dev = reinterpret_cast<float*>(bind<Points>(res, n, st_points_));
    v.normals_xyz_dev = reinterpret_cast<float*>(bind<Normals>(res, n, st_normals_));
    v.face_ids_dev = bind<FaceIds>(res, n, st_face_ids_);
    check(rmclhip_rcc_simulate(sim_, Tbm, nposes, poses_on_device ? 1 : 0, &v));
    fetch<Hits>(res, n, st_hits_);
    fetch<Ranges>(res, n, st_ranges_);
    fetch<Points>(res, n, st_points_);
    fetch<Normals>(res, n, st_normals_);
    fetch<FaceIds>(res, n, st_face_ids_);
  }
  HipMapPtr sim_map_;
  rmclhip_rcc* sim_ = nullptr;
  bool sim_owns_ = false;
  Memory<uint8_t, VRAM_HIP> st_hits_;       // staging of RAM bundles (grow-only)
  Memory<float, VRAM_HIP> st_ranges_;
  Memory<Vector, VRAM_HIP> st_points_, st_normals_;
  Memory<uint32_t, VRAM_HIP> st_face_ids_;
};
using SphereSimulatorHip = SimulatorHip<SphericalModel>;
using O1DnSimulatorHip = SimulatorHip<O1DnModel>;
using PinholeSimulatorHip = SimulatorHip<PinholeModel>;
using OnDnSimulatorHip = SimulatorHip<OnDnModel>;
using SphereSimulatorHipPtr = std::shared_ptr<SphereSimulatorHip>;
using O1DnSimulatorHipPtr = std::shared_ptr<O1DnSimulatorHip>;

// rmcl::RCCEmbree{Spherical, Pinhole, O1Dn, OnDn} (RCCEmbree.hpp:18-83): the correspondence operator + ModelSetter + -- protected --
// the simulator of its model.  find() is the reference's `simulate(Tbm_est, model_buffers_)` (RCCEmbree.cpp:35) on the operator's own
// buffers, which live behind the C handle (rmclhip_rcc_find); like Correspondences_::model_buffers_ they carry {points, normals, hits}
// unless setOutputs() widens the bundle.
template <typename ModelT>
class RCCHip_ : public CorrespondencesHIP, public ModelSetter<ModelT>, protected SimulatorHip<ModelT> {
 public:
  explicit RCCHip_(HipMapPtr map) : CorrespondencesHIP(map), SimulatorHip<ModelT>(typename SimulatorHip<ModelT>::Borrowed{h_, map}) {
    setOutputs(RMCLHIP_OUT_MICP);
  }
  void setModel(const ModelT& sensor_model) override { SimulatorHip<ModelT>::setModel(sensor_model); }
  // RCCEmbree.cpp:15-19: both bases learn Tsb (here they share the engine, so once is enough for it)
  void setTsb(const Transform& Tsb) override {
    CorrespondencesHIP::setTsb(Tsb);
    SimulatorHip<ModelT>::setTsb(Tsb);
  }
};
using RCCHipSpherical = RCCHip_<SphericalModel>;
using RCCHipO1Dn = RCCHip_<O1DnModel>;
using RCCHipPinhole = RCCHip_<PinholeModel>;
using RCCHipOnDn = RCCHip_<OnDnModel>;

// rmcl::CPCEmbree (rmcl/include/rmcl/registration/CPCEmbree.hpp): closest-point correspondences
class CPCHip : public CorrespondencesHIP {
 public:
  explicit CPCHip(HipMapPtr map) : CorrespondencesHIP(std::move(map)) {}
  void find(const Transform& Tbm_est) override {
    bindDataset();   // the closest-point query READS the dataset (CPCEmbree.cpp:27-37), unlike the ray-casting find
    check(rmclhip_rcc_set_params(h_, params.max_dist, adaptive_max_dist_min));
    check(rmclhip_rcc_find_cpc(h_, &Tbm_est));
  }
  // both optional, neither changes a hit: start every query at the triangle it ended on in the previous find (default on) /
  // search only within params.max_dist (default off: points beyond it then carry NaN instead of their gated-out closest point)
  void setTracking(bool on) { check(rmclhip_rcc_set_cpc_tracking(h_, on ? 1 : 0)); }
  void setBounded(bool on) { check(rmclhip_rcc_set_cpc_bounded(h_, on ? 1 : 0)); }
  // the map's near grid seeds points without a tracking seed (default on; a cold query then costs what a tracked one does)
  void setGrid(bool on) { check(rmclhip_rcc_set_cpc_grid(h_, on ? 1 : 0)); }
};

// ---- pose batches over several devices (v1 SphereCorrector::correct shape, lidar_corrector_optix_benchmark.cpp:86-133) ------
// One operator replica per device over one host BVH build; the poses of a batch are block-partitioned, no exchange.  The replicas
// are plain rmclhip_rcc handles (borrowed): configure them with forEach and the C setters, then correctBatch == the unsharded one.
class ShardedCorrectorHip {
 public:
  ShardedCorrectorHip(const std::vector<int>& devices, const float* vertices_xyz, uint32_t n_vertices, const uint32_t* faces_ijk,
                      uint32_t n_faces) {
    check(rmclhip_rcc_sharded_create(devices.data(), static_cast<uint32_t>(devices.size()), vertices_xyz, n_vertices, faces_ijk, n_faces, &h_));
  }
  ~ShardedCorrectorHip() { rmclhip_rcc_sharded_destroy(h_); }
  ShardedCorrectorHip(const ShardedCorrectorHip&) = delete;
  ShardedCorrectorHip& operator=(const ShardedCorrectorHip&) = delete;
  uint32_t size() const { return rmclhip_rcc_sharded_size(h_); }
  rmclhip_rcc* replica(uint32_t rank) const {
    rmclhip_rcc* r = nullptr;
    check(rmclhip_rcc_sharded_replica(h_, rank, &r));
    return r;
  }
  template <typename Fn>
  void forEach(Fn&& fn) const {   // fn(rmclhip_rcc*): e.g. rmclhip_rcc_set_tsb / _set_model_spherical / _set_params / _set_dataset
    for (uint32_t r = 0; r < size(); ++r) fn(replica(r));
  }
  std::vector<Transform> correctBatch(const std::vector<Transform>& Tbm, std::vector<CrossStatistics>* stats = nullptr) {
    std::vector<Transform> out(Tbm.size());
    if (stats) stats->resize(Tbm.size());
    check(rmclhip_rcc_sharded_correct_batch(h_, Tbm.data(), static_cast<uint32_t>(Tbm.size()), out.data(), stats ? stats->data() : nullptr));
    return out;
  }

 private:
  rmclhip_rcc_sharded* h_ = nullptr;
};

// ---- particle filter -----------------------------------------------------------------------------------
struct ParticleUpdateConfig {};
struct ParticleUpdateResults {};

struct SensorUpdaterBase {
  virtual ~SensorUpdaterBase() = default;
  virtual void init() {}
  virtual void reset() {}
};

template <typename MemT>
struct ParticleUpdater;
template <>
struct ParticleUpdater<VRAM_HIP> {
  virtual ~ParticleUpdater() = default;
  virtual ParticleUpdateResults update(DeviceView<Transform> particle_poses, DeviceView<ParticleAttributes> particle_attrs,
                                       const ParticleUpdateConfig& config = {}) = 0;
};

// rmcl::PCDSensorUpdaterOptix on gfx950.  setInput() takes the already sampled measurements (sensor frame)
// and Tsb; sampling from a PointCloud2 and the TF lookup stay with the ROS-side caller
// (PCDSensorUpdaterEmbree.cpp:249-327).
class PCDSensorUpdaterHip : public SensorUpdaterBase, public ParticleUpdater<VRAM_HIP> {
 public:
  rmclhip_pf_params config_{2.0f, 100.0f, 100.0f, 0.0f, {0.05f, 80.0f}, 10000u, 0u};

  explicit PCDSensorUpdaterHip(HipMapPtr map) : map_(std::move(map)) {
    if (!map_) throw std::runtime_error("NO MAP");
  }
  ~PCDSensorUpdaterHip() override { rmclhip_pf_destroy(h_); }
  void init() override {
    if (!h_) check(rmclhip_pf_create(map_->context()->handle(), map_->handle(), &h_));
  }
  void setInput(std::vector<RangeMeasurement> beams, const Transform& Tsb) {
    beams_ = std::move(beams);
    Tsb_ = Tsb;
  }
  // Input<PointCloud2::ConstSharedPtr>::setInput (Input.hpp:7-15) for the raw message: `samples` beams are drawn from the
  // cloud bytes like PCDSensorUpdaterEmbree::update does (:276-327; config_.samples, default 100, :124), with an explicit
  // seed instead of the reference's function-static engine.  Returns the number of beams (< samples: "Point invalid").
  size_t setInput(const uint8_t* cloud_data, size_t nbytes, const rmclhip_pointcloud2_layout& layout, const Transform& Tsb,
                  uint32_t samples = 100, uint64_t seed = 0) {
    beams_.resize(samples);
    uint32_t n = 0;
    check(rmclhip_pf_sample_beams_pointcloud2(cloud_data, nbytes, &layout, samples, seed, beams_.data(), &n));
    beams_.resize(n);
    Tsb_ = Tsb;
    return n;
  }
  const std::vector<RangeMeasurement>& beams() const { return beams_; }
  ParticleUpdateResults update(DeviceView<Transform> poses, DeviceView<ParticleAttributes> attrs,
                               const ParticleUpdateConfig& = {}) override {
    init();
    check(rmclhip_pf_set_params(h_, &config_));
    check(rmclhip_pf_update(h_, poses.raw(), attrs.raw(), static_cast<uint32_t>(poses.size()), beams_.data(),
                            static_cast<uint32_t>(beams_.size()), &Tsb_));
    return {};
  }
  rmclhip_pf* handle() const { return h_; }

 private:
  HipMapPtr map_;
  rmclhip_pf* h_ = nullptr;
  std::vector<RangeMeasurement> beams_;
  Transform Tsb_ = identity();
};

// The particle filter of ONE process over several devices (rmclhip_comm + rmclhip_pf_sharded: RCCL ncclCommInitAll, weight
// all-gather, moment all-reduces): sensor update, pose estimate (RmclNode::estimateStats, rmcl_localization.cpp:642-731) and
// the distributed gladiator tournament for the single-process node (rmcl_localization.cpp:482-552).  The cloud lives in the
// object (block-partitioned over the devices); setParticles / download move it from / to the node's host store.
class PCDSensorUpdaterHipSharded : public SensorUpdaterBase {
 public:
  rmclhip_pf_params config_{2.0f, 100.0f, 100.0f, 0.0f, {0.05f, 80.0f}, 10000u, 0u};

  PCDSensorUpdaterHipSharded(const std::vector<int>& devices, const float* vertices_xyz, uint32_t n_vertices,
                             const uint32_t* faces_ijk, uint32_t n_faces) {
    check(rmclhip_comm_create(devices.data(), static_cast<uint32_t>(devices.size()), &comm_));
    const rmclhip_status st = rmclhip_pf_sharded_create(comm_, vertices_xyz, n_vertices, faces_ijk, n_faces, &h_);
    if (st != RMCLHIP_OK) {
      const std::string msg = rmclhip_last_error();
      rmclhip_comm_destroy(comm_);
      throw std::runtime_error(msg);
    }
  }
  ~PCDSensorUpdaterHipSharded() override {
    rmclhip_pf_sharded_destroy(h_);
    rmclhip_comm_destroy(comm_);
  }
  PCDSensorUpdaterHipSharded(const PCDSensorUpdaterHipSharded&) = delete;
  PCDSensorUpdaterHipSharded& operator=(const PCDSensorUpdaterHipSharded&) = delete;
  void init() override {}
  uint32_t worldSize() const { return rmclhip_comm_size(comm_); }
  void setParticles(const std::vector<Transform>& poses, const std::vector<ParticleAttributes>& attrs) {
    if (poses.size() != attrs.size()) throw std::runtime_error("setParticles: poses.size() != attrs.size()");
    check(rmclhip_pf_sharded_set_particles(h_, poses.data(), attrs.data(), static_cast<uint32_t>(poses.size())));
    n_ = poses.size();
  }
  void download(std::vector<Transform>& poses, std::vector<ParticleAttributes>& attrs) const {
    poses.resize(n_);
    attrs.resize(n_);
    check(rmclhip_pf_sharded_download(h_, poses.data(), attrs.data()));
  }
  void setInput(std::vector<RangeMeasurement> beams, const Transform& Tsb) {
    beams_ = std::move(beams);
    Tsb_ = Tsb;
  }
  // sensor update on every device's block + ONE all-gather of the weights
  ParticleUpdateResults update(const ParticleUpdateConfig& = {}) {
    check(rmclhip_pf_sharded_set_params(h_, &config_));
    check(rmclhip_pf_update_sharded(h_, beams_.data(), static_cast<uint32_t>(beams_.size()), &Tsb_));
    return {};
  }
  std::vector<float> weights(uint32_t rank = 0) const {
    std::vector<float> w(n_);
    check(rmclhip_pf_sharded_get_weights(h_, rank, w.data()));
    return w;
  }
  rmclhip_likelihood_stats computeStats() const {
    rmclhip_likelihood_stats st{};
    check(rmclhip_pf_allreduce_stats(h_, &st));
    return st;
  }
  rmclhip_pose_estimate estimateStats(uint32_t max_induction_particles) const {
    rmclhip_pose_estimate e{};
    check(rmclhip_pf_allreduce_pose_estimate(h_, max_induction_particles, &e));
    return e;
  }
  // MotionUpdater<MemT>::update on every device's block, in place (rmcl_localization.cpp:432-480; particle_motion.cu:11-46 + the collision
  // ray of TFMotionUpdaterCPU.cpp:17-50): the odometry lookup and the combined forget rate stay with the caller, as for TFMotionUpdaterHip
  void motionUpdate(const Transform& T_bnew_bold, double forget_rate, bool check_collision = true) {
    check(rmclhip_pf_sharded_set_params(h_, &config_));
    check(rmclhip_pf_sharded_motion_update(h_, &T_bnew_bold, forget_rate, check_collision ? 1 : 0));
  }
  // one cycle of the node (rmcl_localization.cpp:84, 432-552) in ONE call: motion (T_bnew_bold == nullptr: none) -> sensor update -> weight
  // all-gather -> {sum, max} -> resampling (0 none, 1 gladiator, 2 residual); returns the {sum, max} the resampler normalises with
  rmclhip_likelihood_stats step(const Transform* T_bnew_bold, double forget_rate, bool check_collision, int resample,
                                const rmclhip_gladiator_config& cfg, uint64_t seed, uint32_t step_index) {
    rmclhip_likelihood_stats st{};
    check(rmclhip_pf_sharded_set_params(h_, &config_));
    check(rmclhip_pf_sharded_step(h_, T_bnew_bold, forget_rate, check_collision ? 1 : 0, beams_.data(), static_cast<uint32_t>(beams_.size()), &Tsb_,
                                  resample, &cfg, seed, step_index, &st));
    return st;
  }
  void resample(const rmclhip_gladiator_config& cfg, uint64_t seed, uint32_t step) { check(rmclhip_pf_sharded_resample(h_, &cfg, seed, step)); }
  void resampleResidual(const rmclhip_gladiator_config& cfg, uint64_t seed, uint32_t step) { check(rmclhip_pf_sharded_resample_residual(h_, &cfg, seed, step)); }

 private:
  rmclhip_comm* comm_ = nullptr;
  rmclhip_pf_sharded* h_ = nullptr;
  size_t n_ = 0;
  std::vector<RangeMeasurement> beams_;
  Transform Tsb_ = identity();
};

// rmcl::TFMotionUpdaterGPU (+ the wall-collision test of TFMotionUpdaterCPU) on gfx950: MotionUpdater<MemT>.
// The odometry lookup (TF) and the forget rate (TFMotionUpdaterCPU.cpp:172-174) stay with the caller.
class TFMotionUpdaterHip : public SensorUpdaterBase {
 public:
  bool check_collision = true;
  explicit TFMotionUpdaterHip(HipMapPtr map) : map_(std::move(map)) {
    if (!map_) throw std::runtime_error("NO MAP");
  }
  ~TFMotionUpdaterHip() override { rmclhip_pf_destroy(h_); }
  void init() override {
    if (!h_) check(rmclhip_pf_create(map_->context()->handle(), map_->handle(), &h_));
  }
  ParticleUpdateResults update(DeviceView<Transform> poses, DeviceView<ParticleAttributes> attrs,
                               const Transform& T_bnew_bold, double forget_rate) {
    init();
    check(rmclhip_pf_motion_update(h_, poses.raw(), attrs.raw(), static_cast<uint32_t>(poses.size()), &T_bnew_bold,
                                   forget_rate, check_collision ? 1 : 0));
    return {};
  }

 private:
  HipMapPtr map_;
  rmclhip_pf* h_ = nullptr;
};

// rmcl::GladiatorResamplerGPU on gfx950: Resampler<MemT>::update(poses, attrs, poses_new, attrs_new)
// (GladiatorResamplerGPU.cpp:46-81).  Out of place; `seed` + the running step select the Philox stream.
struct ParticleUpdateDynamicConfig {};
struct ParticleUpdateDynamicResults { size_t n_particles = 0; };
class GladiatorResamplerHip : public SensorUpdaterBase {
 public:
  rmclhip_gladiator_config config_{0.03f, 0.03f, 0.0f, 0.0f, 0.0f, 0.01f, 0.3f, 0.2f, 0u};
  uint64_t seed = 1234;

  explicit GladiatorResamplerHip(ContextPtr ctx) : ctx_(std::move(ctx)) {
    if (!ctx_) throw std::runtime_error("NO CONTEXT");
  }
  ~GladiatorResamplerHip() override { rmclhip_resampler_destroy(h_); }
  void init() override {
    if (!h_) check(rmclhip_resampler_create(ctx_->handle(), &h_));
  }
  void reset() override { step_ = 0; }
  rmclhip_likelihood_stats computeStats(DeviceView<ParticleAttributes> attrs) {
    init();
    rmclhip_likelihood_stats s{};
    check(rmclhip_resampler_compute_stats(h_, attrs.raw(), static_cast<uint32_t>(attrs.size()), &s));
    return s;
  }
  ParticleUpdateDynamicResults update(DeviceView<Transform> poses, DeviceView<ParticleAttributes> attrs,
                                      DeviceView<Transform> poses_new, DeviceView<ParticleAttributes> attrs_new,
                                      const ParticleUpdateDynamicConfig& = {}) {
    init();
    const uint32_t n_new = static_cast<uint32_t>(poses_new.size());
    check(rmclhip_resampler_gladiator(h_, poses.raw(), attrs.raw(), static_cast<uint32_t>(poses.size()), poses_new.raw(),
                                      attrs_new.raw(), 0u, n_new, &config_, seed, step_++));
    return {n_new};
  }

 private:
  ContextPtr ctx_;
  rmclhip_resampler* h_ = nullptr;
  uint32_t step_ = 0;
};

// rmcl::ResidualResamplerCPU (ResidualResamplerCPU.cpp:55-203), the node's other Resampler plugin (rmcl_localization.cpp:567): same
// interface and parameters as the gladiator; the new cloud may have a different size than the old one
class ResidualResamplerHip : public SensorUpdaterBase {
 public:
  rmclhip_gladiator_config config_{0.03f, 0.03f, 0.0f, 0.0f, 0.0f, 0.01f, 0.3f, 0.2f, 1u};
  uint64_t seed = 1234;
  uint64_t last_draws = 0;   // iterations the reference's sequential loop would have run

  explicit ResidualResamplerHip(ContextPtr ctx) : ctx_(std::move(ctx)) {
    if (!ctx_) throw std::runtime_error("NO CONTEXT");
  }
  ~ResidualResamplerHip() override { rmclhip_resampler_destroy(h_); }
  void init() override {
    if (!h_) check(rmclhip_resampler_create(ctx_->handle(), &h_));
  }
  void reset() override { step_ = 0; }
  ParticleUpdateDynamicResults update(DeviceView<Transform> poses, DeviceView<ParticleAttributes> attrs,
                                      DeviceView<Transform> poses_new, DeviceView<ParticleAttributes> attrs_new,
                                      const ParticleUpdateDynamicConfig& = {}) {
    init();
    const uint32_t n_new = static_cast<uint32_t>(poses_new.size());
    check(rmclhip_resampler_residual(h_, poses.raw(), attrs.raw(), static_cast<uint32_t>(poses.size()), poses_new.raw(),
                                     attrs_new.raw(), n_new, 0u, n_new, &config_, seed, step_++, &last_draws));
    return {n_new};
  }

 private:
  ContextPtr ctx_;
  rmclhip_resampler* h_ = nullptr;
  uint32_t step_ = 0;
};

}  // namespace rmcl_hip
