// simulator_cpp_example.cpp -- the rmagine-level Simulator interface of include/rmcl_hip/rmcl_hip.hpp driven the way the reference's
// direct simulator callers drive theirs:
//   * ScanMapSegmentationEmbreeNode (rmcl_ros/src/nodes/filter/scan_map_segmentation_embree.cpp:38-39 construction, :76-87 the
//     per-scan body, :100-185 the classification loop) -- copied call for call below with the type names changed
//     (rm::SphereSimulatorEmbree -> SphereSimulatorHip, rm:: -> rmcl_hip::), the ROS message replaced by two plain structs;
//   * the stale v1 benchmarks' `correct.simulate(T_dest, sim_res)` with Bundle<Ranges<RAM>> and a Memory<Transform> of poses, on the
//     host (lidar_corrector_embree_benchmark.cpp:95-117) and with poses and ranges in device memory (lidar_corrector_optix_benchmark.cpp:
//     96-119);
//   * CorrespondencesCUDA::computeCrossStatistics (rmcl/src/rmcl/registration/CorrespondencesCUDA.cpp:9-30): rm::watch(dataset), a model
//     view {points, mask = hits, normals}, the max_dist interpolation and the FREE function rm::statistics_p2l on those views.
//
//   g++ -std=c++17 -Iinclude examples/simulator_cpp_example.cpp -Lrmcl_amd -lrmclhip -Wl,-rpath,$PWD/rmcl_amd -o simulator_example
//   ./simulator_example mesh.bin scan.bin     (mesh.bin: u32 nv, u32 nf, nv*3 f32, nf*3 u32; scan.bin: 32*32 f32 measured ranges)
//
// Prints one "key value..." line per result; tests/test_cpp_adapters.py compares them with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "rmcl_hip/rmcl_hip.hpp"

namespace rm = rmcl_hip;   // the reference's callers write rm:: for rmagine

static rm::Transform from_rpy(float x, float y, float z, double roll, double pitch, double yaw) {
  const double cr = std::cos(roll / 2), sr = std::sin(roll / 2), cp = std::cos(pitch / 2), sp = std::sin(pitch / 2);
  const double cy = std::cos(yaw / 2), sy = std::sin(yaw / 2);
  rm::Transform T = rm::identity();
  T.R.x = static_cast<float>(sr * cp * cy - cr * sp * sy);
  T.R.y = static_cast<float>(cr * sp * cy + sr * cp * sy);
  T.R.z = static_cast<float>(cr * cp * sy - sr * sp * cy);
  T.R.w = static_cast<float>(cr * cp * cy + sr * sp * sy);
  T.t = {x, y, z};
  return T;
}

// what rmcl_msgs::msg::ScanStamped carries into scanCB, and the cloud it fills
struct Scan {
  rm::SphericalModel info;
  std::vector<float> ranges;
};
struct SegmentationCloud {
  size_t width = 0;
  double sx = 0, sy = 0, sz = 0;
  void push_back(const rm::Vector& p) { ++width; sx += p.x; sy += p.y; sz += p.z; }
};

// ScanMapSegmentationEmbreeNode with the simulator type changed (scan_map_segmentation_embree.cpp)
class ScanMapSegmentationHipNode {
 public:
  float min_dist_outlier_scan_ = 0.15f, min_dist_outlier_map_ = 0.15f;   // map_segmentation.cpp's parameters

  explicit ScanMapSegmentationHipNode(rm::HipMapPtr map) {
    // :38-39
    scan_sim_ = std::make_shared<rm::SphereSimulatorHip>(map);
    scan_sim_->setTsb(rm::identity());
  }

  void scanCB(const Scan& msg, const rm::Transform& T, SegmentationCloud& cloud_outlier_scan2, SegmentationCloud& cloud_outlier_map2) const {
    // :76-87
    rm::SphericalModel model;
    model = msg.info;   // convert(msg->scan.info, model);
    scan_sim_->setModel(model);

    using ResultT = rm::Bundle<
      rm::Ranges<rm::RAM>,
      rm::Normals<rm::RAM>
    >;

    ResultT res = scan_sim_->simulate<ResultT>(T);

    // :89-90
    const rm::MemoryView<float, rm::RAM> ranges = res.ranges;
    const rm::MemoryView<rm::Vector, rm::RAM> normals = res.normals;

    // :108-185
    for (size_t vid = 0; vid < model.getHeight(); vid++) {
      for (size_t hid = 0; hid < model.getWidth(); hid++) {
        const size_t bid = model.getBufferId(vid, hid);

        const float range_real = msg.ranges[bid];
        const float range_sim = ranges[bid];

        const bool range_real_valid = model.range.inside(range_real);
        const bool range_sim_valid = model.range.inside(range_sim);

        if (range_real_valid) {
          rm::Vector preal_s = model.getDirection(vid, hid) * range_real + model.getOrigin(vid, hid);

          if (range_sim_valid) {
            rm::Vector pint_s = model.getDirection(vid, hid) * range_sim;
            rm::Vector nint_s = normals[bid];
            nint_s = nint_s * (1.0f / rm::l2norm(nint_s));   // nint_s.normalizeInplace();

            float signed_plane_dist = rm::dot(preal_s - pint_s, nint_s);
            const rm::Vector pmesh_s = preal_s + nint_s * signed_plane_dist;
            const float plane_distance = rm::l2norm(pmesh_s - preal_s);

            if (range_real < range_sim) {
              // something is in front of surface
              if (plane_distance > min_dist_outlier_scan_) cloud_outlier_scan2.push_back(preal_s);
            } else {
              // ray cutted the surface
              if (plane_distance > min_dist_outlier_map_) cloud_outlier_map2.push_back(pint_s);
            }
          } else {
            // point in real scan but not in simulated
            cloud_outlier_scan2.push_back(preal_s);
          }
        } else {
          if (range_sim_valid) {
            // sim hits surface but real not: map could be wrong
            rm::Vector pint_s = model.getDirection(vid, hid) * range_sim + model.getOrigin(vid, hid);
            cloud_outlier_map2.push_back(pint_s);
          }
        }
      }
    }
  }

 private:
  rm::SphereSimulatorHipPtr scan_sim_;
};

// the v1 correctors expose their simulator base publicly (`correct.simulate(T_dest, sim_res)`); the RCC classes keep it protected
// (RCCEmbree.hpp:18-22) -- a subclass may open it
class SphereCorrectorHip : public rm::RCCHipSpherical {
 public:
  using rm::RCCHipSpherical::RCCHipSpherical;
  using rm::SimulatorHip<rm::SphericalModel>::simulate;
};

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s mesh.bin scan.bin\n", argv[0]); return 2; }
  std::FILE* fh = std::fopen(argv[1], "rb");
  if (!fh) { std::perror("mesh"); return 2; }
  uint32_t nv = 0, nf = 0;
  if (std::fread(&nv, 4, 1, fh) != 1 || std::fread(&nf, 4, 1, fh) != 1) return 2;
  std::vector<float> verts(3 * static_cast<size_t>(nv));
  std::vector<uint32_t> faces(3 * static_cast<size_t>(nf));
  if (std::fread(verts.data(), 4, verts.size(), fh) != verts.size()) return 2;
  if (std::fread(faces.data(), 4, faces.size(), fh) != faces.size()) return 2;
  std::fclose(fh);

  const float pi = 3.14159265358979323846f;
  Scan scan;
  scan.info.phi = {-pi / 4, (pi / 2) / 31, 32};
  scan.info.theta = {-pi, 2 * pi / 32, 32};
  scan.info.range = {0.1f, 100.0f};
  scan.ranges.resize(scan.info.size());
  fh = std::fopen(argv[2], "rb");
  if (!fh) { std::perror("scan"); return 2; }
  if (std::fread(scan.ranges.data(), 4, scan.ranges.size(), fh) != scan.ranges.size()) return 2;
  std::fclose(fh);

  try {
    auto ctx = std::make_shared<rm::Context>(0);
    auto map = std::make_shared<rm::HipMap>(ctx, verts.data(), nv, faces.data(), nf);
    const rm::Transform T_sensor_map = from_rpy(0.5f, -0.3f, 0.2f, 0.02, -0.03, 0.4);

    // ---- the segmentation node ------------------------------------------------------------------------------------------------
    {
      ScanMapSegmentationHipNode node(map);
      SegmentationCloud outlier_scan, outlier_map;
      node.scanCB(scan, T_sensor_map, outlier_scan, outlier_map);
      std::printf("seg_outlier_scan %zu %.9g %.9g %.9g\n", outlier_scan.width, outlier_scan.sx, outlier_scan.sy, outlier_scan.sz);
      std::printf("seg_outlier_map %zu %.9g %.9g %.9g\n", outlier_map.width, outlier_map.sx, outlier_map.sy, outlier_map.sz);
    }

    // ---- the v1 benchmarks' simulate(Memory<Transform>, Bundle<Ranges>) ----------------------------------------------------------
    const rm::SphericalModel model = scan.info;
    const rm::Transform Tsb = from_rpy(0.1f, 0.0f, 0.3f, 0, 0, 10.0 * pi / 180);
    {
      SphereCorrectorHip correct(map);
      correct.setTsb(Tsb);
      correct.setModel(model);

      using ResultT = rm::Bundle<
          rm::Ranges<rm::RAM>
      >;
      rm::Memory<rm::Transform, rm::RAM> T_dest(3);
      for (size_t i = 0; i < T_dest.size(); i++) {
        T_dest[i] = T_sensor_map;
        T_dest[i].t.z += 0.2f * static_cast<float>(i);
      }
      ResultT sim_res;
      sim_res.ranges.resize(model.size() * T_dest.size());
      correct.simulate(T_dest, sim_res);                      // lidar_corrector_embree_benchmark.cpp:117
      for (size_t i = 0; i < T_dest.size(); i++) {
        double s = 0;
        for (size_t k = 0; k < model.size(); k++) s += sim_res.ranges[i * model.size() + k];
        std::printf("batch_ranges_%zu %.9g\n", i, s);
      }
      // lidar_corrector_optix_benchmark.cpp:96-119: poses and results in device memory
      using ResultD = rm::Bundle<
          rm::Ranges<rm::VRAM_HIP>
      >;
      ResultD sim_res_;
      sim_res_.ranges.setContext(ctx);
      sim_res_.ranges.resize(model.size() * T_dest.size());
      rm::Memory<rm::Transform, rm::VRAM_HIP> T_dest_(ctx);
      T_dest_ = T_dest;
      correct.simulate(T_dest_, sim_res_);
      rm::Memory<float, rm::RAM> back;
      sim_res_.ranges.download(back);
      size_t same = 0;
      for (size_t k = 0; k < back.size(); k++) same += (back[k] == sim_res.ranges[k]) ? 1u : 0u;
      std::printf("batch_device_equal %zu %zu\n", same, back.size());

      // ---- CorrespondencesCUDA::computeCrossStatistics written out with the free function -----------------------------------------
      // dataset: the measured scan as MICPSphericalSensorCPU::unpackMessage builds it (:206-226)
      rm::Memory<rm::Vector, rm::RAM> ds_points(model.size());
      rm::Memory<uint8_t, rm::RAM> ds_mask(model.size());
      for (uint32_t vid = 0; vid < model.getHeight(); vid++)
        for (uint32_t hid = 0; hid < model.getWidth(); hid++) {
          const uint32_t bid = model.getBufferId(vid, hid);
          const float r = scan.ranges[bid];
          ds_points[bid] = model.getDirection(vid, hid) * r;
          ds_mask[bid] = model.range.inside(r) ? 1 : 0;
        }
      correct.dataset.points = ds_points;
      correct.dataset.mask = ds_mask;
      correct.params.max_dist = 1.0f;
      correct.adaptive_max_dist_min = 0.15f;
      const rm::Transform Tbm_est = T_sensor_map * from_rpy(0.2f, 0.1f, 0.05f, 0, 0, 2.0 * pi / 180);
      correct.find(Tbm_est);
      const double convergence_progress = 0.25;
      const rm::Transform T_snew_sold = from_rpy(0.01f, -0.02f, 0.005f, 0.001, 0.002, -0.003);

      const rm::PointCloudView_<rm::VRAM_HIP> cloud_dataset = rm::watch(correct.dataset);
      const rm::PointCloudView_<rm::VRAM_HIP> cloud_model = correct.modelView();   // {points, mask = hits, normals}
      rm::UmeyamaReductionConstraints params_local = correct.params;
      params_local.max_dist = static_cast<float>(correct.params.max_dist * (1.0 - convergence_progress) + correct.adaptive_max_dist_min * convergence_progress);
      const rm::CrossStatistics stats_free = rm::statistics_p2l(T_snew_sold, cloud_dataset, cloud_model, params_local);
      const rm::CrossStatistics stats_op = correct.computeCrossStatistics(T_snew_sold, convergence_progress);
      std::printf("p2l_free %u %.9g %.9g %.9g %.9g\n", stats_free.n_meas, stats_free.dataset_mean.x, stats_free.model_mean.z,
                  stats_free.covariance[0], stats_free.covariance[5]);
      std::printf("p2l_operator %u %.9g %.9g %.9g %.9g\n", stats_op.n_meas, stats_op.dataset_mean.x, stats_op.model_mean.z,
                  stats_op.covariance[0], stats_op.covariance[5]);

      // the operator's bundle is {points, normals, hits}: ranges and face ids were never written
      const auto mb = correct.modelBuffers();
      std::printf("operator_bundle %u %d %d %d %d %d\n", correct.outputs(), mb.points != nullptr, mb.normals != nullptr, mb.mask != nullptr,
                  mb.ranges != nullptr, mb.face_ids != nullptr);

      // a simulate() into the caller's bundle between find and computeCrossStatistics leaves the operator's correspondences alone
      using MicpT = rm::Bundle<rm::Points<rm::VRAM_HIP>, rm::Normals<rm::VRAM_HIP>, rm::Hits<rm::VRAM_HIP>>;
      MicpT other;
      rm::resize_memory_bundle<rm::VRAM_HIP>(other, model.getHeight(), model.getWidth(), 1, ctx);
      correct.simulate(T_sensor_map, other);
      const rm::CrossStatistics stats_again = correct.computeCrossStatistics(T_snew_sold, convergence_progress);
      std::printf("p2l_after_simulate %u %.9g\n", stats_again.n_meas, stats_again.covariance[0]);
      // ... and its results serve the free function as a model view of their own (designated-initialiser style of CorrespondencesCUDA.cpp:14-18)
      rm::PointCloudView_<rm::VRAM_HIP> other_view;
      other_view.points = {other.points.raw(), other.points.size()};
      other_view.mask = {other.hits.raw(), other.hits.size()};
      other_view.normals = {other.normals.raw(), other.normals.size()};
      other_view.ctx = ctx->handle();
      const rm::CrossStatistics stats_truth = rm::statistics_p2l(rm::identity(), cloud_dataset, other_view, params_local);
      std::printf("p2l_truth_pose %u %.9g %.9g\n", stats_truth.n_meas, stats_truth.covariance[0] + stats_truth.covariance[4] + stats_truth.covariance[8],
                  rm::l2norm(stats_truth.dataset_mean - stats_truth.model_mean));
    }

    // ---- the other simulators: O1Dn / OnDn fed with the spherical directions, pinhole ----------------------------------------------
    {
      rm::O1DnModel o1;
      o1.width = model.getWidth(); o1.height = model.getHeight(); o1.range = model.range; o1.orig = rm::Vector{0.f, 0.f, 0.f};
      rm::OnDnModel on;
      on.width = o1.width; on.height = o1.height; on.range = model.range;
      for (uint32_t vid = 0; vid < model.getHeight(); vid++)
        for (uint32_t hid = 0; hid < model.getWidth(); hid++) {
          o1.dirs.push_back(model.getDirection(vid, hid));
          on.dirs.push_back(model.getDirection(vid, hid));
          on.origs.push_back(rm::Vector{0.f, 0.f, 0.f});
        }
      using IdsT = rm::Bundle<rm::Hits<rm::RAM>, rm::FaceIds<rm::RAM>>;
      auto tally = [](const IdsT& r, const char* key) {
        unsigned long long hs = 0, fs = 0;
        for (size_t i = 0; i < r.hits.size(); i++) { hs += r.hits[i]; fs += r.hits[i] ? r.face_ids[i] : 0u; }
        std::printf("%s %llu %llu\n", key, hs, fs);
      };
      rm::SphereSimulatorHip ss(map);
      ss.setTsb(Tsb); ss.setModel(model);
      tally(ss.simulate<IdsT>(T_sensor_map), "sim_sphere");
      rm::O1DnSimulatorHip s1(map);
      s1.setTsb(Tsb); s1.setModel(o1);
      tally(s1.simulate<IdsT>(T_sensor_map), "sim_o1dn");
      rm::OnDnSimulatorHip sn(map);
      sn.setTsb(Tsb); sn.setModel(on);
      tally(sn.simulate<IdsT>(T_sensor_map), "sim_ondn");
      rm::PinholeModel ph;
      ph.width = 32; ph.height = 32; ph.range = model.range; ph.f[0] = 20.f; ph.f[1] = 20.f; ph.c[0] = 15.5f; ph.c[1] = 15.5f;
      rm::PinholeSimulatorHip sp(map);
      sp.setTsb(Tsb); sp.setModel(ph);
      tally(sp.simulate<IdsT>(T_sensor_map), "sim_pinhole");
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
