// micp_cpp_example.cpp -- the C++ adapters of include/rmcl_hip/rmcl_hip.hpp driven the way the reference's
// callers drive their operators: MICPSensor_::findCorrespondences / computeCrossStatistics
// (rmcl_ros/include/rmcl_ros/micpl/MICPSensor.hpp:146-184), the correctOnce inner loop
// (rmcl_ros/src/nodes/micp_localization.cpp:915-964) and SensorUpdater::update.
//
//   g++ -std=c++17 -Iinclude examples/micp_cpp_example.cpp -Lrmcl_amd -lrmclhip -Wl,-rpath,$PWD/rmcl_amd -o micp_example
//   ./micp_example mesh.bin       (mesh.bin: u32 nv, u32 nf, nv*3 f32, nf*3 u32)
//
// Prints one "key value" pair per line; tests/test_cpp_adapters.py compares them with the Python path.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rmcl_hip/rmcl_hip.hpp"
#include "rmclhip_bench.h"   // rmclhip_rcc_time_caller_loop: a measurement aid, not part of the boundary

using namespace rmcl_hip;

static Transform from_rpy(float x, float y, float z, double roll, double pitch, double yaw) {
  const double cr = std::cos(roll / 2), sr = std::sin(roll / 2), cp = std::cos(pitch / 2), sp = std::sin(pitch / 2);
  const double cy = std::cos(yaw / 2), sy = std::sin(yaw / 2);
  Transform T = identity();
  T.R.x = static_cast<float>(sr * cp * cy - cr * sp * sy);
  T.R.y = static_cast<float>(cr * sp * cy + sr * cp * sy);
  T.R.z = static_cast<float>(cr * cp * sy - sr * sp * cy);
  T.R.w = static_cast<float>(cr * cp * cy + sr * sp * sy);
  T.t = {x, y, z};
  return T;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s mesh.bin\n", argv[0]); return 2; }
  std::FILE* fh = std::fopen(argv[1], "rb");
  if (!fh) { std::perror("mesh"); return 2; }
  uint32_t nv = 0, nf = 0;
  if (std::fread(&nv, 4, 1, fh) != 1 || std::fread(&nf, 4, 1, fh) != 1) return 2;
  std::vector<float> verts(3 * static_cast<size_t>(nv));
  std::vector<uint32_t> faces(3 * static_cast<size_t>(nf));
  if (std::fread(verts.data(), 4, verts.size(), fh) != verts.size()) return 2;
  if (std::fread(faces.data(), 4, faces.size(), fh) != faces.size()) return 2;
  std::fclose(fh);

  try {
    auto ctx = std::make_shared<Context>(0);
    auto map = std::make_shared<HipMap>(ctx, verts.data(), nv, faces.data(), nf);

    // sensor: 32x32 spherical scanner (config C1), mounted with an offset
    const float pi = 3.14159265358979323846f;
    SphericalModel model{};
    model.phi = {-pi / 4, (pi / 2) / 31, 32};
    model.theta = {-pi, 2 * pi / 32, 32};
    model.range = {0.1f, 100.0f};
    const Transform Tsb = from_rpy(0.1f, 0.0f, 0.3f, 0, 0, 10.0 * pi / 180);
    const Transform Tbo = identity();
    const Transform truth = from_rpy(0.5f, -0.3f, 0.2f, 0.02, -0.03, 0.4);
    const Transform Tom_est = truth * from_rpy(0.2f, 0.1f, 0.05f, 0, 0, 2.0 * pi / 180);

    RCCHipSpherical rcc(map);
    rcc.setTsb(Tsb);
    rcc.setModel(model);
    // the operator's bundle is {points, normals, hits} like Correspondences_::model_buffers_ (Correspondences.hpp:81-85); this example
    // also reads ranges and face ids back
    rcc.setOutputs(RMCLHIP_OUT_ALL);
    rcc.params.max_dist = 1.0f;
    rcc.adaptive_max_dist_min = 0.15f;

    // "measured" scan: simulate at the true pose, then unpackMessage-style dataset from the ranges
    const uint32_t n = model.phi.size * model.theta.size;
    rcc.find(truth * Tbo);
    std::vector<float> ranges(n);
    std::vector<uint32_t> face_ids(n);
    std::vector<uint8_t> hits(n);
    rcc.download(hits.data(), ranges.data(), nullptr, nullptr, face_ids.data());
    const uint32_t valid = rcc.setDatasetFromRanges(ranges.data(), n);
    uint64_t face_sum = 0, hit_sum = 0;
    for (uint32_t i = 0; i < n; ++i) { face_sum += hits[i] ? face_ids[i] : 0; hit_sum += hits[i]; }
    std::printf("hits %llu\nface_sum %llu\nvalid %u\n", (unsigned long long)hit_sum, (unsigned long long)face_sum, valid);

    // MICPSensor_::findCorrespondences + the correctOnce inner loop on the host, 5 iterations
    rcc.find(Tom_est * Tbo);
    Transform T_onew_oold = identity();
    CrossStatistics Cmerged = cross_statistics_identity();
    for (int i = 0; i < 5; ++i) {
      const Transform T_bnew_bold = ~Tbo * T_onew_oold * Tbo;
      const Transform T_snew_sold = ~Tsb * T_bnew_bold * Tsb;
      const CrossStatistics stats_s = rcc.computeCrossStatistics(T_snew_sold, 0.0);
      const CrossStatistics Cs_o = Tbo * (Tsb * stats_s);
      Cmerged = cross_statistics_identity();
      Cmerged += Cs_o;
      T_onew_oold = T_onew_oold * umeyama_transform(Cmerged);
    }
    std::printf("host_loop_n_meas %u\nhost_loop_t %.9g %.9g %.9g\nhost_loop_q %.9g %.9g %.9g %.9g\n", Cmerged.n_meas,
                T_onew_oold.t.x, T_onew_oold.t.y, T_onew_oold.t.z, T_onew_oold.R.x, T_onew_oold.R.y, T_onew_oold.R.z,
                T_onew_oold.R.w);
    {
      // round 4: the five calls above were answered from the moments of the find's correspondences -- one moment pass at the first
      // call, the other four on the host without a launch (rmclhip_rcc_ccs_info); the same unchanged loop timed in C
      const rmclhip_ccs_info ci = rcc.ccsInfo();
      float ms = 0.f;
      Transform Tcl{};
      CrossStatistics scl{};
      check(rmclhip_rcc_time_caller_loop(rcc.handle(), &Tom_est, &Tbo, 5, 0.0, 20, &Tcl, &scl, &ms));
      const rmclhip_ccs_info c2 = rcc.ccsInfo();
      std::printf("caller_loop_served %u %u %u\ncaller_loop_timed %u %u %u %u\ncaller_loop_t %.9g %.9g %.9g\ncaller_loop_n_meas %u\n", ci.calls,
                  ci.from_moments, ci.passes, c2.calls - ci.calls, c2.from_moments - ci.from_moments, c2.passes - ci.passes,
                  c2.speculative_finds, Tcl.t.x, Tcl.t.y, Tcl.t.z, scl.n_meas);
      std::fprintf(stderr, "unchanged caller loop: %.4f ms per 5-iteration correction\n", ms);
    }
    // the other model adapters (RCCEmbree.cpp:57-68, 89-99, 120-130) and the closest-point operator (CPCEmbree.cpp:18-44) on the same scan:
    // O1Dn / OnDn fed with the spherical model's own directions must answer like the spherical operator; the pinhole operator and CPCHip
    // are compared with the oracle by tests/test_cpp_adapters.py
    {
      auto tally = [n](const CorrespondencesHIP& op, const char* key) {
        std::vector<uint8_t> h(n);
        std::vector<uint32_t> ids(n);
        op.download(h.data(), nullptr, nullptr, nullptr, ids.data());
        uint64_t fs = 0, hs = 0;
        for (uint32_t i = 0; i < n; ++i) { fs += h[i] ? ids[i] : 0; hs += h[i]; }
        std::printf("%s %llu %llu\n", key, (unsigned long long)hs, (unsigned long long)fs);
      };
      O1DnModel o1;
      o1.width = model.theta.size; o1.height = model.phi.size; o1.range = model.range; o1.orig = Vector{0.f, 0.f, 0.f};
      OnDnModel on;
      on.width = o1.width; on.height = o1.height; on.range = model.range;
      for (uint32_t vid = 0; vid < model.phi.size; ++vid)
        for (uint32_t hid = 0; hid < model.theta.size; ++hid) {
          o1.dirs.push_back(getDirection(model, vid, hid));
          on.dirs.push_back(getDirection(model, vid, hid));
          on.origs.push_back(Vector{0.f, 0.f, 0.f});
        }
      RCCHipO1Dn r1(map);
      r1.setOutputs(RMCLHIP_OUT_ALL); r1.setTsb(Tsb); r1.setModel(o1); r1.find(truth * Tbo);
      tally(r1, "o1dn");
      RCCHipOnDn rn(map);
      rn.setOutputs(RMCLHIP_OUT_ALL); rn.setTsb(Tsb); rn.setModel(on); rn.find(truth * Tbo);
      tally(rn, "ondn");
      PinholeModel ph;
      ph.width = 32; ph.height = 32; ph.range = model.range; ph.f[0] = 20.f; ph.f[1] = 20.f; ph.c[0] = 15.5f; ph.c[1] = 15.5f;
      RCCHipPinhole rp(map);
      rp.setOutputs(RMCLHIP_OUT_ALL); rp.setTsb(Tsb); rp.setModel(ph); rp.find(truth * Tbo);
      tally(rp, "pinhole");
      CPCHip cpc(map);
      cpc.setTsb(Tsb);
      cpc.params.max_dist = 1.0f;
      cpc.adaptive_max_dist_min = 1.0f;
      std::vector<float> pts(3 * static_cast<size_t>(n));
      std::vector<uint8_t> msk(n);
      for (uint32_t vid = 0; vid < model.phi.size; ++vid)
        for (uint32_t hid = 0; hid < model.theta.size; ++hid) {
          const uint32_t i = getBufferId(model, vid, hid);
          const Vector d = getDirection(model, vid, hid);
          pts[3 * i] = d.x * ranges[i]; pts[3 * i + 1] = d.y * ranges[i]; pts[3 * i + 2] = d.z * ranges[i];
          msk[i] = (ranges[i] >= model.range.min && ranges[i] <= model.range.max) ? 1 : 0;
        }
      cpc.setDataset(pts.data(), msk.data(), n);
      cpc.find(Tom_est * Tbo);
      tally(cpc, "cpc");
      const CrossStatistics cs = cpc.computeCrossStatistics(identity(), 0.0);
      std::printf("cpc_n_meas %u\n", cs.n_meas);
    }
    // Correspondences_::dataset filled the way the reference's device sensors fill it (MICPSphericalSensorCUDA.cpp:207-232):
    // points / mask built on the host per measurement, then `dataset.points = host.points; dataset.mask = host.mask;`
    {
      RCCHipSpherical rcc2(map);
      rcc2.setTsb(Tsb);
      rcc2.setModel(model);
      rcc2.params.max_dist = 1.0f;
      rcc2.adaptive_max_dist_min = 0.15f;
      PointCloud_<RAM> dataset_cpu;
      dataset_cpu.points.resize(n);
      dataset_cpu.mask.resize(n);
      uint32_t valid2 = 0;
      for (uint32_t vid = 0; vid < model.phi.size; ++vid)
        for (uint32_t hid = 0; hid < model.theta.size; ++hid) {
          const uint32_t loc_id = getBufferId(model, vid, hid);
          const float real_range = ranges[loc_id];
          const Vector d = getDirection(model, vid, hid);
          dataset_cpu.points[loc_id] = Vector{d.x * real_range, d.y * real_range, d.z * real_range};
          const bool out_of_range = real_range < model.range.min || real_range > model.range.max;
          dataset_cpu.mask[loc_id] = out_of_range ? 0 : 1;
          valid2 += out_of_range ? 0 : 1;
        }
      rcc2.dataset.points = dataset_cpu.points;   // upload
      rcc2.dataset.mask = dataset_cpu.mask;
      rcc2.find(Tom_est * Tbo);
      const CrossStatistics a = rcc2.computeCrossStatistics(identity(), 0.0);
      rcc.find(Tom_est * Tbo);
      const CrossStatistics b = rcc.computeCrossStatistics(identity(), 0.0);
      const auto dv = rcc2.datasetView();
      std::printf("dataset_member_valid %u\ndataset_member_n_meas %u %u\ndataset_member_cov00 %.9g %.9g\ndataset_view %zu %zu\n", valid2,
                  a.n_meas, b.n_meas, a.covariance[0], b.covariance[0], dv.points.size(), dv.mask.size());
    }
    // modelView() (Correspondences.hpp:47-53) the way MICPSensorCUDA.cpp:66-84 reads it: {points, mask, normals} views on the device
    {
      rcc.find(truth * Tbo);
      const auto mv = rcc.modelView();
      Memory<uint8_t, RAM> mask_h;
      Memory<Vector, RAM> points_h, normals_h;
      download(*ctx, mv.mask, mask_h);
      download(*ctx, mv.points, points_h);
      download(*ctx, mv.normals, normals_h);
      uint64_t mask_sum = 0;
      double range_sum = 0, unit = 0;
      for (size_t i = 0; i < mask_h.size(); ++i) {
        if (!mask_h[i]) continue;
        ++mask_sum;
        range_sum += std::sqrt(points_h[i].x * points_h[i].x + points_h[i].y * points_h[i].y + points_h[i].z * points_h[i].z);
        unit += normals_h[i].x * normals_h[i].x + normals_h[i].y * normals_h[i].y + normals_h[i].z * normals_h[i].z;
      }
      double ranges_sum = 0;
      for (uint32_t i = 0; i < n; ++i) ranges_sum += hits[i] ? ranges[i] : 0.0;
      std::printf("model_view %zu %llu %.9g %.9g %.9g\n", mv.points.size(), (unsigned long long)mask_sum, range_sum, ranges_sum, unit);
    }
    // a scene (rm::import_embree_map of a file with several nodes, micp_localization.cpp:187-195): the same mesh twice, the
    // second copy scaled and out of sight 200 m away -- the scan must not change, face ids stay those of instance 0
    {
      const std::vector<rmclhip_mesh> scene_meshes = {rmclhip_mesh{verts.data(), faces.data(), nv, nf}};
      const std::vector<rmclhip_instance> scene_instances = {
          HipMap::instance(0, identity()), HipMap::instance(0, from_rpy(200.f, 0.f, 0.f, 0, 0, 0.5), Vector{2.f, 1.f, 0.5f})};
      auto scene = std::make_shared<HipMap>(ctx, scene_meshes, scene_instances);
      RCCHipSpherical rcc3(scene);
      rcc3.setOutputs(RMCLHIP_OUT_HITS | RMCLHIP_OUT_FACE_IDS);
      rcc3.setTsb(Tsb);
      rcc3.setModel(model);
      rcc3.find(truth * Tbo);
      std::vector<uint32_t> fid3(n);
      std::vector<uint8_t> hit3(n);
      rcc3.download(hit3.data(), nullptr, nullptr, nullptr, fid3.data());
      uint64_t fs3 = 0, hs3 = 0, inst_sum = 0;
      for (uint32_t i = 0; i < n; ++i) {
        if (!hit3[i]) continue;
        fs3 += fid3[i];
        ++hs3;
        inst_sum += scene->locate(fid3[i]).instance;
      }
      const std::vector<uint32_t> first = scene->sceneInstances();
      std::printf("scene %zu %u %u %u\nscene_hits %llu\nscene_face_sum %llu\nscene_instance_sum %llu\n", first.size() - 1, first[0],
                  first[1], first[2], (unsigned long long)hs3, (unsigned long long)fs3, (unsigned long long)inst_sum);
    }
    // the same loop resident on the device
    CrossStatistics sdev{};
    const Transform Tdev = rcc.correctOnce(Tom_est, Tbo, 5, 0.0, false, &sdev);
    std::printf("device_loop_n_meas %u\ndevice_loop_t %.9g %.9g %.9g\n", sdev.n_meas, Tdev.t.x, Tdev.t.y, Tdev.t.z);
    // repeats of the same correction run in the moment form once the first call has learnt its bounds (rmclhip.h:
    // rmclhip_rcc_set_micp_fast); the result is the same
    {
      Transform Trep = Tdev;
      CrossStatistics srep{};
      for (int k = 0; k < 3; ++k) Trep = rcc.correctOnce(Tom_est, Tbo, 5, 0.0, false, &srep);
      const rmclhip_micp_fast_info info = rcc.micpFastInfo();
      std::printf("moment_form_attempts_done %u %u\nmoment_form_n_meas %u\nmoment_form_t %.9g %.9g %.9g\n", info.attempts, info.done,
                  srep.n_meas, Trep.t.x, Trep.t.y, Trep.t.z);
    }

    // the N-sensor entry point with this one sensor (the node's loop over sensors_vec_, micp_localization.cpp:921-938)
    {
      CrossStatistics sm{};
      const Transform Tm = correctOnce({&rcc}, Tom_est, {Tbo}, {1.0}, 5, 0.0, &sm);
      std::printf("multi_loop_n_meas %u\nmulti_loop_t %.9g %.9g %.9g\n", sm.n_meas, Tm.t.x, Tm.t.y, Tm.t.z);
    }

    // v1 pose batch (lidar_corrector_optix_benchmark.cpp:86-133) through one operator and through TWO replicas on device 0
    // (rmclhip_rcc_sharded_*: poses block-partitioned, no exchange): identical deltas
    {
      std::vector<Transform> batch;
      for (int k = 0; k < 7; ++k) batch.push_back(truth * from_rpy(0.05f * k, -0.03f * k, 0.01f * k, 0, 0, 0.01f * k));
      rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0f;
      const std::vector<Transform> one = rcc.correctBatch(batch);
      ShardedCorrectorHip sh({0, 0}, verts.data(), nv, faces.data(), nf);
      std::vector<Vector> pts(n);
      std::vector<uint8_t> msk(n);
      for (uint32_t vid = 0; vid < model.phi.size; ++vid)
        for (uint32_t hid = 0; hid < model.theta.size; ++hid) {
          const uint32_t id = getBufferId(model, vid, hid);
          const Vector d = getDirection(model, vid, hid);
          pts[id] = Vector{d.x * ranges[id], d.y * ranges[id], d.z * ranges[id]};
          msk[id] = (ranges[id] < model.range.min || ranges[id] > model.range.max) ? 0 : 1;
        }
      sh.forEach([&](rmclhip_rcc* r) {
        check(rmclhip_rcc_set_tsb(r, &Tsb));
        check(rmclhip_rcc_set_model_spherical(r, model.c_model()));
        check(rmclhip_rcc_set_params(r, 1.0f, 1.0f));
        check(rmclhip_rcc_set_dataset(r, &pts[0].x, msk.data(), n, 0));
      });
      const std::vector<Transform> two = sh.correctBatch(batch);
      uint32_t same = 0;
      for (size_t k = 0; k < batch.size(); ++k) same += std::memcmp(&one[k], &two[k], sizeof(Transform)) == 0 ? 1u : 0u;
      std::printf("sharded_batch %u %zu %u\n", sh.size(), batch.size(), same);
      rcc.params.max_dist = 1.0f;
      rcc.adaptive_max_dist_min = 0.15f;
    }

    // particle filter: 4 hypotheses, 3 beams
    std::vector<Transform> poses = {truth, Tom_est, from_rpy(1, 1, 0, 0, 0, 1.0), from_rpy(-2, 0.5f, 0.3f, 0, 0, -2.0)};
    std::vector<ParticleAttributes> attrs(poses.size());
    for (auto& a : attrs) { a = ParticleAttributes{}; a.likelihood.mean = 1.0f; }
    std::vector<RangeMeasurement> beams(3);
    const float dirs[3][3] = {{1, 0, 0}, {0, 1, 0}, {0.6f, 0, 0.8f}};
    for (int b = 0; b < 3; ++b) {
      beams[b] = RangeMeasurement{};
      beams[b].dir = {dirs[b][0], dirs[b][1], dirs[b][2]};
      beams[b].range = 3.0f + b;
    }
    void *d_poses = nullptr, *d_attrs = nullptr;
    check(rmclhip_malloc(ctx->handle(), poses.size() * sizeof(Transform), &d_poses));
    check(rmclhip_malloc(ctx->handle(), attrs.size() * sizeof(ParticleAttributes), &d_attrs));
    check(rmclhip_memcpy_h2d(ctx->handle(), d_poses, poses.data(), poses.size() * sizeof(Transform)));
    check(rmclhip_memcpy_h2d(ctx->handle(), d_attrs, attrs.data(), attrs.size() * sizeof(ParticleAttributes)));
    PCDSensorUpdaterHip upd(map);
    upd.init();
    upd.setInput(beams, Tsb);
    upd.update({static_cast<Transform*>(d_poses), poses.size()}, {static_cast<ParticleAttributes*>(d_attrs), attrs.size()});
    check(rmclhip_memcpy_d2h(ctx->handle(), attrs.data(), d_attrs, attrs.size() * sizeof(ParticleAttributes)));
    for (size_t i = 0; i < attrs.size(); ++i)
      std::printf("pf_%zu %.9g %.9g %u\n", i, attrs[i].likelihood.mean, attrs[i].likelihood.sigma, attrs[i].likelihood.n_meas);
    // beams drawn from a raw PointCloud2 (x y z float32 + one more float32 field, point_step 16) like
    // PCDSensorUpdaterEmbree::update does: the simulated scan at the true pose, misses as NaN points
    {
      std::vector<float> pts3(3 * static_cast<size_t>(n));
      rcc.find(truth * Tbo);
      rcc.download(nullptr, nullptr, pts3.data(), nullptr, nullptr);
      std::vector<float> cloud(4 * static_cast<size_t>(n), 0.0f);
      for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) cloud[4 * i + k] = pts3[3 * i + k];
      rmclhip_pointcloud2_layout L{};
      L.width = n; L.height = 1; L.point_step = 16; L.row_step = 16 * n; L.offset_x = 0; L.offset_y = 4; L.offset_z = 8; L.datatype = 7;
      PCDSensorUpdaterHip upd2(map);
      const size_t nb = upd2.setInput(reinterpret_cast<const uint8_t*>(cloud.data()), cloud.size() * 4, L, Tsb, 40, 1234);
      double range_sum = 0;
      for (const auto& bm : upd2.beams()) range_sum += bm.range;
      std::printf("sampled_beams %zu %.9g\n", nb, range_sum);
    }
    // the same update through the multi-device form (one process, here one device): weights, {sum, max}, pose estimate
    {
      std::vector<ParticleAttributes> attrs0(poses.size());
      for (auto& a : attrs0) { a = ParticleAttributes{}; a.likelihood.mean = 1.0f; }
      PCDSensorUpdaterHipSharded sharded({0}, verts.data(), nv, faces.data(), nf);
      sharded.setParticles(poses, attrs0);
      sharded.setInput(beams, Tsb);
      sharded.update();
      const std::vector<float> w = sharded.weights();
      const rmclhip_likelihood_stats sst = sharded.computeStats();
      const rmclhip_pose_estimate est = sharded.estimateStats(1000000);
      std::printf("sharded_world %u\nsharded_w %.9g %.9g %.9g %.9g\nsharded_stats %.9g %.9g\nsharded_pose_t %.9g %.9g %.9g\n",
                  sharded.worldSize(), w[0], w[1], w[2], w[3], sst.sum, sst.max, est.pose.t.x, est.pose.t.y, est.pose.t.z);
      // the node's cycle on the sharded cloud (rmcl_localization.cpp:84, 432-552): a motion update on its own, then motion -> sensor
      // update -> weight all-gather -> {sum, max} in ONE call (no resampling here: the clouds below are compared particle by particle)
      const Transform T_bnew_bold = from_rpy(0.3f, 0, 0, 0, 0, 0.05f);
      sharded.motionUpdate(T_bnew_bold, 0.01, true);
      rmclhip_gladiator_config no_cfg{};
      const rmclhip_likelihood_stats cst = sharded.step(&T_bnew_bold, 0.01, true, 0, no_cfg, 42, 0);
      std::vector<Transform> pc;
      std::vector<ParticleAttributes> ac;
      sharded.download(pc, ac);
      std::printf("sharded_cycle_stats %.9g %.9g\n", cst.sum, cst.max);
      for (size_t i = 0; i < ac.size(); ++i)
        std::printf("sharded_cycle_%zu %.9g %u %.9g %.9g %.9g\n", i, ac[i].likelihood.mean, ac[i].likelihood.n_meas, pc[i].t.x, pc[i].t.y, pc[i].t.z);
    }
    // motion update (30 cm forward, 1 % forgetting, wall-collision test) and one gladiator tournament
    TFMotionUpdaterHip motion(map);
    const DeviceView<Transform> vposes{static_cast<Transform*>(d_poses), poses.size()};
    const DeviceView<ParticleAttributes> vattrs{static_cast<ParticleAttributes*>(d_attrs), attrs.size()};
    motion.update(vposes, vattrs, from_rpy(0.3f, 0, 0, 0, 0, 0.05f), 0.01);
    void *d_poses_new = nullptr, *d_attrs_new = nullptr;
    check(rmclhip_malloc(ctx->handle(), poses.size() * sizeof(Transform), &d_poses_new));
    check(rmclhip_malloc(ctx->handle(), attrs.size() * sizeof(ParticleAttributes), &d_attrs_new));
    GladiatorResamplerHip resampler(ctx);
    resampler.seed = 42;
    const rmclhip_likelihood_stats st = resampler.computeStats(vattrs);
    std::printf("stats %.9g %.9g\n", st.sum, st.max);
    const auto res = resampler.update(vposes, vattrs, {static_cast<Transform*>(d_poses_new), poses.size()},
                                      {static_cast<ParticleAttributes*>(d_attrs_new), attrs.size()});
    check(rmclhip_memcpy_d2h(ctx->handle(), poses.data(), d_poses_new, poses.size() * sizeof(Transform)));
    check(rmclhip_memcpy_d2h(ctx->handle(), attrs.data(), d_attrs_new, attrs.size() * sizeof(ParticleAttributes)));
    std::printf("resampled %zu\n", res.n_particles);
    {
      // the node's other resampler plugin (rmcl_localization.cpp:567): residual resampling of the same cloud into a larger one
      const size_t n_big = 3 * poses.size();
      void *d_pr = nullptr, *d_ar = nullptr;
      check(rmclhip_malloc(ctx->handle(), n_big * sizeof(Transform), &d_pr));
      check(rmclhip_malloc(ctx->handle(), n_big * sizeof(ParticleAttributes), &d_ar));
      ResidualResamplerHip residual(ctx);
      residual.seed = 42;
      const auto rr = residual.update(vposes, vattrs, {static_cast<Transform*>(d_pr), n_big}, {static_cast<ParticleAttributes*>(d_ar), n_big});
      std::vector<ParticleAttributes> ar(n_big);
      check(rmclhip_memcpy_d2h(ctx->handle(), ar.data(), d_ar, n_big * sizeof(ParticleAttributes)));
      double lsum = 0;
      for (const auto& a : ar) lsum += a.likelihood.mean;
      std::printf("residual %zu %llu %.9g\n", rr.n_particles, (unsigned long long)residual.last_draws, lsum);
      check(rmclhip_free(ctx->handle(), d_pr));
      check(rmclhip_free(ctx->handle(), d_ar));
    }
    for (size_t i = 0; i < attrs.size(); ++i)
      std::printf("rs_%zu %.9g %u %.9g %.9g %.9g\n", i, attrs[i].likelihood.mean, attrs[i].likelihood.n_meas, poses[i].t.x,
                  poses[i].t.y, poses[i].t.z);
    check(rmclhip_free(ctx->handle(), d_poses_new));
    check(rmclhip_free(ctx->handle(), d_attrs_new));
    check(rmclhip_free(ctx->handle(), d_poses));
    check(rmclhip_free(ctx->handle(), d_attrs));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
