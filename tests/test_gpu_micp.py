"""GPU parity of the MICP-L correction loop (micp_localization.cpp:900-964, MICPSensor.hpp:146-184):
the on-device loop (rmclhip_rcc_correct_once), the host loop over find/computeCrossStatistics
(rmcl_amd.micp.MICPLocalization.correctOnce) and the v1 batch corrector
(lidar_corrector_embree_benchmark.cpp:127-135), all against the oracle's restatement and the
committed G5 trajectories.  Pose deltas within 1e-5 (north_star).
"""
import math

import numpy as np
import pytest

import oracle_micp as om
from conftest import golden_path
from test_gpu_reduce import _transform_close

pytestmark = pytest.mark.gpu


def _sphere_setup(ra, orc, ctx, meshes):
    from rmcl_amd import synthetic as syn
    v, f = meshes("sphere20k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_vlp16_900(0.0)
    return m, hm, model


def test_g5_sphere_scenario_converges(ra, orc, ctx, meshes):
    """stale-benchmark scenario (lidar_corrector_embree_benchmark.cpp:84-135): sphere map, start at z+0.2,
    10 iterations; analytic known answer z -> 0; both schedules vs the committed oracle trajectories."""
    from rmcl_amd import types as T
    g = np.load(golden_path("g5_micp_sphere20k.npz"))
    m, hm, model = _sphere_setup(ra, orc, ctx, meshes)
    ident = T.identity()
    meas = m.simulate_spherical(model, ident, ident, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    Tom = T.transform((0, 0, 0, 1), (0.0, 0.0, 0.2))
    for name, refind in (("R", False), ("B", True)):
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(ident)
        rcc.setModel(model)
        rcc.set_dataset(ds, mask)
        rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
        traj = g["traj_" + name].view(T.TRANSFORM)
        for k in (1, 3, 10):
            Tk, stats = rcc.correct_once(Tom, ident, k, 0.0, refind)
            _transform_close(Tk, traj[k - 1], 1e-5)
        T_new = T.mult(Tom, Tk)
        # the (B) schedule re-raycasts and converges to the truth; (R) keeps the initial correspondences
        # (a fixed correspondence set only reaches the point-to-plane optimum of THOSE correspondences)
        # z is weakly observable for a +-15 deg sensor (Kabsch on plane-projected points contracts it
        # slowly), so the analytic check is direction + x/y, not full convergence after 10 steps
        assert 0.0 < float(T_new["t"]["z"]) < 0.19
        assert abs(float(T_new["t"]["x"])) < 1e-3 and abs(float(T_new["t"]["y"])) < 1e-3
        assert int(stats["n_meas"]) == int(g["stats_" + name].view(T.CROSS_STATISTICS)[0]["n_meas"])
        rcc.close()


def test_device_loop_equals_host_loop_with_frames(ra, orc, ctx, meshes):
    """non-trivial Tsb / Tbo / Tom and a convergence_progress: device loop == host loop == oracle == G5."""
    from rmcl_amd import synthetic as syn, types as T
    g = np.load(golden_path("g5_micp_sphere20k.npz"))
    m, hm, model = _sphere_setup(ra, orc, ctx, meshes)
    Tsb, Tbo, Tom2 = (g[k].view(T.TRANSFORM)[0] for k in ("Tsb", "Tbo", "Tom2"))
    meas = m.simulate_spherical(model, Tsb, T.mult(T.identity(), Tbo), bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc = ra.RCCHipSpherical(hm)
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
    sensor = ra.MICPSensor("lidar", rcc, Tsb=Tsb, Tbo=Tbo)
    sensor.valid_dataset_measurements = int(mask.sum())
    sensor.total_dataset_measurements = len(mask)
    # device loop
    Tdev, sdev = rcc.correct_once(Tom2, Tbo, 10, 0.3, False)
    traj = g["traj_frames"].view(T.TRANSFORM)
    _transform_close(Tdev, traj[-1], 1e-5)
    # host loop (the reference's call pattern: find once, computeCrossStatistics per iteration)
    loc = ra.MICPLocalization([sensor], optimization_iterations=10)
    loc.Tom_ = Tom2
    loc.convergence_progress_ = 0.3
    rec = []
    Thost = loc.correctOnce(record=rec)
    for k in range(10):
        _transform_close(rec[k], traj[k], 1e-5)
    _transform_close(Thost, Tdev, 1e-5)
    assert loc.correction_stats_latest_["valid_matches"] == int(sdev["n_meas"])
    assert 0.0 <= loc.convergence_progress_ <= 1.0
    # and the oracle run live
    Tor, _, _ = om.correct_once(m, model, Tsb, Tbo, Tom2, ds, mask, 10, 1.0, adaptive_min=0.15, convergence_progress=0.3, nthreads=8)
    _transform_close(Tdev, Tor, 1e-5)


def test_c3_full_size_inner_loop(ra, orc, ctx, meshes):
    """config C3: 128x1024 scan, 100k-triangle mesh, 10 ICP iterations, both schedules vs the oracle."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("sphere100k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c2()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    ident = T.identity()
    meas = m.simulate_spherical(model, ident, truth, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(ident)
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
    for refind in (False, True):
        Tg, sg = rcc.correct_once(est, ident, 10, 0.0, refind)
        To, so, _ = om.correct_once(m, model, ident, ident, est, ds, mask, 10, 1.0, adaptive_min=0.15, refind=refind, nthreads=8)
        _transform_close(Tg, To, 1e-5)
        assert int(sg["n_meas"]) == int(so["n_meas"])
    # on a sphere only the translation is observable; (B) must pull the sensor back towards the truth
    Tnew = T.mult(est, Tg)
    d0 = math.dist([float(est["t"][k]) for k in "xyz"], [float(truth["t"][k]) for k in "xyz"])
    d1 = math.dist([float(Tnew["t"][k]) for k in "xyz"], [float(truth["t"][k]) for k in "xyz"])
    assert d1 < 0.2 * d0


def test_correct_batch_v1_api(ra, orc, ctx, meshes):
    """v1 SphereCorrector::correct over a pose batch: Tdelta per pose vs the oracle, with Tsb != I;
    then T_curr = multNxN(T_curr, Tdelta) drives every hypothesis towards the truth."""
    from rmcl_amd import synthetic as syn, types as T
    m, hm, model = _sphere_setup(ra, orc, ctx, meshes)
    Tsb = syn.tsb_offset()
    ident = T.identity()
    meas = m.simulate_spherical(model, Tsb, ident, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rng = np.random.RandomState(11)
    poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-0.3, 0.3, 3)), (0.0, 0.0, rng.uniform(-0.05, 0.05)))
                      for _ in range(37)], dtype=T.TRANSFORM)
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(Tsb)
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
    Td, st = rcc.correct_batch(poses)
    Tr, sr = om.correct_batch(m, model, Tsb, poses, ds, mask, 1.0, nthreads=8)
    for i in range(len(poses)):
        assert int(st[i]["n_meas"]) == int(sr[i]["n_meas"])
        _transform_close(Td[i], Tr[i], 1e-5)
    # batch model buffers: pose-major, identical to single-pose finds
    mv = rcc.modelView()
    n = model.phi.size * model.theta.size
    rcc.find(poses[5])
    single = rcc.modelView()
    assert np.array_equal(mv["face_ids"][5 * n:6 * n], single["face_ids"])
    assert np.array_equal(mv["ranges"][5 * n:6 * n], single["ranges"])
    cur = poses.copy()
    for _ in range(8):
        Td, _ = rcc.correct_batch(cur)
        cur = np.array([T.mult(cur[i], Td[i]) for i in range(len(cur))], dtype=T.TRANSFORM)
    t1 = np.array([[float(c["t"][k]) for k in "xyz"] for c in cur])
    t0 = np.array([[float(c["t"][k]) for k in "xyz"] for c in poses])
    # x/y are well observable (rotation is not, on a sphere: the base may keep |yaw error| * |Tsb.t|);
    # z contracts slowly for a +-15 deg sensor but must not grow
    assert np.abs(t1[:, :2]).max() < 2e-2
    assert np.all(np.abs(t1[:, 2]) <= np.abs(t0[:, 2]) + 1e-3)


@pytest.mark.parametrize("variant_bits", [0, 1 << 8, 1 << 9, (1 << 8) | (1 << 9), 1 << 10, (1 << 10) | (1 << 9), 4 << 10,
                                          (6 << 10) | (1 << 8), 7 << 10])
def test_loop_variants_agree(ra, orc, ctx, meshes, variant_bits):
    """A/B code paths of the MICP loop (fused last-block reduction tail, hipGraph replay on/off, loop form: one
    launch per iteration / reduce + solve launches / persistent kernel with a grid barrier) give the same
    statistics and pose as the default path; repeated calls with changing inputs exercise graph replay with
    fresh per-call parameters (pose, Tbo, max_dist, Tsb)."""
    from rmcl_amd import synthetic as syn, types as T
    m, hm, model = _sphere_setup(ra, orc, ctx, meshes)
    ident = T.identity()
    meas = m.simulate_spherical(model, ident, ident, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc = ra.RCCHipSpherical(hm)
    rcc.set_variant(2 | variant_bits)   # 14 400 rays: the quad traversal, the automatic choice for this scan
    rcc.setTsb(ident)
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    cases = [(T.transform((0, 0, 0, 1), (0.0, 0.0, 0.2)), ident, 1.0, 0.0, ident),
             (T.transform_from_rpy((0.1, -0.05, 0.1), (0, 0, 0.02)), T.transform_from_rpy((0.2, 0.1, 0.0), (0, 0, 0.3)), 0.7, 0.4, syn.tsb_offset()),
             (T.transform((0, 0, 0, 1), (0.05, 0.0, -0.1)), ident, 1.0, 0.0, ident)]
    for Tom, Tbo, md, p, Tsb in cases:
        rcc.setTsb(Tsb)
        rcc.params.max_dist, rcc.adaptive_max_dist_min = md, 0.15
        Tg, sg = rcc.correct_once(Tom, Tbo, 4, p, False)
        To, so, _ = om.correct_once(m, model, Tsb, Tbo, Tom, ds, mask, 4, md, adaptive_min=0.15, convergence_progress=p, nthreads=8)
        _transform_close(Tg, To, 1e-5)
        assert int(sg["n_meas"]) == int(so["n_meas"])
        s1 = rcc.computeCrossStatistics(ident, p)
        assert int(s1["n_meas"]) > 0
    Td, st = rcc.correct_batch(np.array([c[0] for c in cases], dtype=T.TRANSFORM))
    assert len(Td) == 3 and int(st[0]["n_meas"]) > 0


def test_find_batch_o1dn_pose_major(ra, orc, ctx, meshes):
    """Simulator::simulate(Memory<Transform>, Bundle&) for an O1Dn model: pose-major buffers == single finds."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    sm = syn.model_pf16()
    dirs = syn.model_directions(sm)
    rcc = ra.RCCHipO1Dn(hm)
    rcc.setTsb(syn.tsb_offset())
    rcc.setModel(16, 16, 0.05, 80.0, (0.01, 0.02, 0.03), dirs)
    poses, _ = syn.uniform_particles(33, seed=9, bb_min=(-8, -8, 0.5, 0, 0, -3), bb_max=(8, 8, 2.5, 0, 0, 3))
    for variant in (23, 0, 2, 24):
        rcc.set_traversal(variant)
        rcc.find_batch(poses)
        mv = rcc.modelView()
        ref = m.simulate_o1dn(16, 16, 0.05, 80.0, (0.01, 0.02, 0.03), dirs, syn.tsb_offset(), poses, bvh=True, nthreads=4)
        assert np.array_equal(mv["face_ids"], ref["face_ids"]) and np.array_equal(mv["hits"], ref["hits"])
        assert np.allclose(mv["ranges"], ref["ranges"], rtol=1e-5)


def test_two_sensor_device_loop_equals_host_loop(ra, orc, ctx, meshes):
    """MICPLocalizationNode::correctOnce with TWO sensors (micp_localization.cpp:921-938: per-sensor statistics, optimal +
    weighted merge, one solve per iteration): the device-resident loop (rmclhip_micp_correct_once) must equal the host loop
    that calls computeCrossStatistics per sensor and iteration -- different models, mounts, odometry stamps (Tbo) and merge
    weights, including a weight that truncates (n_meas *= 0.37)."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    truth = T.transform_from_rpy((1.0, -2.0, 1.4), (0.02, -0.03, 0.4))
    est = T.mult(truth, T.transform_from_rpy((0.15, -0.1, 0.04), (0.01, 0.0, 0.03)))
    specs = [("lidar", syn.model_vlp16_900(0.3), syn.tsb_offset(), T.identity(), 1.0),
             ("lidar2", syn.model_c1(), T.transform_from_rpy((-0.2, 0.1, 0.5), (0.0, 0.1, -1.0)),
              T.transform_from_rpy((0.01, 0.0, 0.0), (0.0, 0.0, 0.002)), 0.37)]

    def build():
        sensors = []
        for name, model, Tsb, Tbo, w in specs:
            meas = m.simulate_spherical(model, Tsb, T.mult(truth, Tbo), bvh=True, nthreads=8)
            ds, mask = om.dataset_from_ranges(model, meas["ranges"])
            rcc = ra.RCCHipSpherical(hm)
            rcc.setModel(model)
            rcc.set_dataset(ds, mask)
            rcc.params.max_dist, rcc.adaptive_max_dist_min = 0.8, 0.2
            s = ra.MICPSensor(name, rcc, Tsb=Tsb, Tbo=Tbo, merge_weight_multiplier=w)
            s.valid_dataset_measurements = int(mask.sum())
            sensors.append(s)
        return sensors

    est_far = est
    est_near = T.mult(truth, T.transform_from_rpy((0.03, -0.02, 0.01), (0.002, 0.0, 0.006)))
    for n_iter, progress, est, want_done in ((5, 0.0, est_far, False), (10, 0.4, est_far, False), (10, 0.1, est_near, True)):
        host = ra.MICPLocalization(build(), optimization_iterations=n_iter)
        host.Tom_, host.convergence_progress_ = est, progress
        rec = []
        Th = host.correctOnce(record=rec)
        dev = ra.MICPLocalization(build(), optimization_iterations=n_iter)
        # the same correction three times: the first call of the moment form of the N-sensor loop learns its bounds (falls
        # back to one streaming launch per sensor and iteration), the repeats run in the single-workgroup loop
        for rep in range(3):
            dev.Tom_, dev.convergence_progress_ = est, progress
            Td = dev.correctOnce(device_loop=True)
            _transform_close(Td, Th, 1e-5)
        infos = [s.correspondences_.micp_fast_info() for s in dev.sensors_vec_]
        # a correction that is large against the gate (est_far: the points move up to 0.9 m, max_dist is 0.8) leaves every
        # correspondence undecided and always takes the fallback; the small one must complete in the moment form
        assert all(i["attempts"] == 3 for i in infos) and (not want_done or all(i["done"] >= 1 for i in infos)), infos
        # ... and with the moment form switched off on one sensor the whole loop takes the per-iteration form
        dev.sensors_vec_[1].correspondences_.set_micp_fast(0)
        dev.Tom_, dev.convergence_progress_ = est, progress
        Td = dev.correctOnce(device_loop=True)
        _transform_close(Td, Th, 1e-5)
        assert dev.sensors_vec_[0].correspondences_.micp_fast_info()["attempts"] == 3
        assert dev.correction_stats_latest_["valid_matches"] == host.correction_stats_latest_["valid_matches"] > 1000
        assert abs(dev.convergence_progress_ - host.convergence_progress_) < 1e-6
        _transform_close(dev.Tom_, host.Tom_, 1e-5)
        # the correction really moves towards the truth
        d0 = np.linalg.norm([est["t"][k] - truth["t"][k] for k in "xyz"])
        d1 = np.linalg.norm([dev.Tom_["t"][k] - truth["t"][k] for k in "xyz"])
        assert d1 < 0.5 * d0
        for s in host.sensors_vec_ + dev.sensors_vec_:
            s.correspondences_.close()
    # one sensor through the N-sensor entry point == the single-sensor graph path
    sensors = build()[:1]
    loc = ra.MICPLocalization(sensors, optimization_iterations=7)
    loc.Tom_ = est
    Tm = loc.correctOnce(device_loop=True)
    Tg, _ = sensors[0].correspondences_.correct_once(est, sensors[0].Tbo, 7, 0.0, False)
    _transform_close(Tm, Tg, 1e-5)
    sensors[0].correspondences_.close()


def test_six_sensor_device_loop_equals_host_loop(ra, orc, ctx, meshes):
    """the N-sensor device loop with SIX sensors (more than the four waves of its workgroup: wave w evaluates sensors w and w + 4;
    lanes 0..5 of wave 0 own one sensor each): different models, mounts, odometry offsets and merge weights (one of them 0:
    the sensor counts in the optimal merge only).  Equal to the host loop -- per sensor and iteration one computeCrossStatistics --
    in the moment form (small correction) and in its fallback (large correction)."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    truth = T.transform_from_rpy((1.0, -2.0, 1.4), (0.02, -0.03, 0.4))
    rng = np.random.RandomState(5)
    models = [syn.model_c1(), syn.model_vlp16_900(0.3), syn.model_pf16(), syn.model_c1(), syn.model_pf16(), syn.model_c1()]
    weights = [1.0, 0.37, 2.0, 0.0, 1.0, 0.5]
    specs = []
    for k, (model, w) in enumerate(zip(models, weights)):
        Tsb = T.transform_from_rpy(tuple(rng.uniform(-0.3, 0.3, 3)), tuple(rng.uniform(-0.5, 0.5, 3)))
        Tbo = T.transform_from_rpy(tuple(rng.uniform(-0.01, 0.01, 3)), (0.0, 0.0, float(rng.uniform(-0.003, 0.003))))
        specs.append(("s%d" % k, model, Tsb, Tbo, w))

    def build():
        sensors = []
        for name, model, Tsb, Tbo, w in specs:
            meas = m.simulate_spherical(model, Tsb, T.mult(truth, Tbo), bvh=True, nthreads=8)
            ds, mask = om.dataset_from_ranges(model, meas["ranges"])
            rcc = ra.RCCHipSpherical(hm)
            rcc.setModel(model)
            rcc.set_dataset(ds, mask)
            rcc.params.max_dist, rcc.adaptive_max_dist_min = 0.8, 0.2
            s = ra.MICPSensor(name, rcc, Tsb=Tsb, Tbo=Tbo, merge_weight_multiplier=w)
            s.valid_dataset_measurements = int(mask.sum())
            sensors.append(s)
        return sensors

    est_far = T.mult(truth, T.transform_from_rpy((0.15, -0.1, 0.04), (0.01, 0.0, 0.03)))
    est_near = T.mult(truth, T.transform_from_rpy((0.02, -0.015, 0.01), (0.001, 0.0, 0.004)))
    for n_iter, progress, est, want_done in ((6, 0.2, est_far, False), (9, 0.1, est_near, True)):
        host = ra.MICPLocalization(build(), optimization_iterations=n_iter)
        host.Tom_, host.convergence_progress_ = est, progress
        Th = host.correctOnce()
        dev = ra.MICPLocalization(build(), optimization_iterations=n_iter)
        for rep in range(3):
            dev.Tom_, dev.convergence_progress_ = est, progress
            Td = dev.correctOnce(device_loop=True)
            _transform_close(Td, Th, 1e-5)
        infos = [s.correspondences_.micp_fast_info() for s in dev.sensors_vec_]
        assert all(i["attempts"] == 3 for i in infos) and (not want_done or all(i["done"] >= 1 for i in infos)), infos
        assert dev.correction_stats_latest_["valid_matches"] == host.correction_stats_latest_["valid_matches"] > 1000
        _transform_close(dev.Tom_, host.Tom_, 1e-5)
        for s in host.sensors_vec_ + dev.sensors_vec_:
            s.correspondences_.close()


def _room_case(ra, orc, ctx, meshes, model):
    from rmcl_amd import types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    truth = T.transform_from_rpy((1.0, -1.5, 1.2), (0.01, -0.02, 0.5))
    meas = m.simulate_spherical(model, T.identity(), truth, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    return m, hm, truth, ds, mask


def test_moment_form_equals_per_iteration_form(ra, orc, ctx, meshes):
    """The moment form of the schedule-(R) loop (rmclhip.h: rmclhip_rcc_set_micp_fast) against the per-iteration form and the
    oracle on a room with occluders and an open ceiling (gated-out correspondences, misses, dist values on both sides of
    max_dist): same n_meas, pose within 1e-6; the first call learns the bounds (falls back), repeats run the moment form."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c1()
    m, hm, truth, ds, mask = _room_case(ra, orc, ctx, meshes, model)
    Tsb, Tbo = syn.tsb_offset(), T.transform_from_rpy((0.3, -0.1, 0.0), (0.0, 0.0, 0.2))
    meas = m.simulate_spherical(model, Tsb, T.mult(truth, T.identity()), bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    done_total, unc_seen, host_loops = 0, {1: 0, 4: 0}, 0
    for pert, rpy, prog in (((0.03, -0.02, 0.01), (0.0, 0.0, 0.01), 0.0), ((0.12, 0.08, -0.03), (0.01, -0.01, 0.03), 0.0),
                            ((0.01, 0.0, 0.0), (0.0, 0.0, 0.002), 0.9), ((0.3, -0.2, 0.1), (0.02, 0.0, -0.06), 0.2)):
        # the localisation state: Tom * Tbo = truth * perturbation
        est_bm = T.mult(truth, T.transform_from_rpy(pert, rpy))
        Tom = T.mult(est_bm, T.inv(Tbo))
        res = {}
        for mode in (0, 1, 4):      # 1: iterations on the host (default), 4: round 3's device loop
            rcc = ra.RCCHipSpherical(hm)
            rcc.setTsb(Tsb)
            rcc.setModel(model)
            rcc.set_dataset(ds, mask)
            rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
            rcc.set_micp_fast(mode)
            out = [rcc.correct_once(Tom, Tbo, 8, prog, False) for _ in range(3)]
            res[mode] = out
            if mode != 0:
                info = rcc.micp_fast_info()
                assert info["attempts"] == 3
                done_total += info["done"]
                if info["last_code"] == 0:
                    unc_seen[mode] = max(unc_seen[mode], info["last_uncertain"])
                if mode == 1:
                    host_loops += info["host_loops"]
                else:
                    assert info["host_loops"] == 0
            else:
                assert rcc.micp_fast_info()["attempts"] == 0
            rcc.close()
        To, so, _ = om.correct_once(m, model, Tsb, Tbo, Tom, ds, mask, 8, 1.0, adaptive_min=0.15, convergence_progress=prog, nthreads=8)
        for k in range(3):
            for mode in (1, 4):
                Tf, sf = res[mode][k]
                Tc, sc = res[0][k]
                assert int(sf["n_meas"]) == int(sc["n_meas"]) == int(so["n_meas"])
                _transform_close(Tf, Tc, 1e-6)
                _transform_close(Tf, To, 1e-5)
                assert np.allclose(sf["covariance"], sc["covariance"], rtol=1e-5, atol=1e-6)
    assert done_total >= 8          # the moment form did run (not only its fallback), in both forms
    assert host_loops >= 4          # ... and in mode 1 the HOST ran the iterations
    assert unc_seen[1] > 0 and unc_seen[4] > 0   # ... including the re-evaluation of undecided correspondences on either side


@pytest.mark.parametrize("fast", [1, 4])
def test_moment_form_fallbacks_keep_the_result(ra, orc, ctx, meshes, fast):
    """Both exits of the moment form: (a) a correction far larger than the learnt bounds (pre-transform leaves the caps),
    (b) a gate so tight that most correspondences sit near it (more than 4096 undecided): the per-iteration form takes
    over and the result equals the one with the moment form switched off."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c2()
    v, f = meshes("sphere100k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    ident = T.identity()
    truth = syn.pose_c2_truth()
    meas = m.simulate_spherical(model, ident, truth, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    small = T.mult(truth, T.transform_from_rpy((0.01, 0.0, 0.0), (0.0, 0.0, 0.001)))
    large = T.mult(truth, syn.pose_c2_perturbation())
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(ident)
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    rcc.set_micp_fast(fast)
    ref = ra.RCCHipSpherical(hm)
    ref.setTsb(ident)
    ref.setModel(model)
    ref.set_dataset(ds, mask)
    ref.set_micp_fast(0)
    seq = [(small, 1.0), (small, 1.0), (large, 1.0), (large, 1.0), (small, 1.0), (large, 0.21), (large, 0.21), (large, 0.21), (small, 1.0)]
    codes = []
    for est, md in seq:
        for r in (rcc, ref):
            r.params.max_dist = r.adaptive_max_dist_min = md
        Tf, sf = rcc.correct_once(est, ident, 10, 0.0, False)
        Tc, sc = ref.correct_once(est, ident, 10, 0.0, False)
        assert int(sf["n_meas"]) == int(sc["n_meas"])
        _transform_close(Tf, Tc, 1e-6)
        codes.append(rcc.micp_fast_info()["last_code"])
    info = rcc.micp_fast_info()
    assert info["cap_exits"] >= 1 and info["done"] >= 2, (codes, info)
    assert 2 in codes or info["overflows"] >= 0      # the tight gate may or may not overflow on this mesh: reported either way
    rcc.close()
    ref.close()


def test_moment_form_o1dn_unmasked_dataset_and_short_dataset(ra, orc, ctx, meshes):
    """The moment form behind the other entry conditions of correct_once: an O1Dn model with NaN directions (misses),
    a dataset handed over WITHOUT a mask (NaN points stay in and are gated out by the reduction's own comparison), and a
    dataset shorter than the model (the reduction runs over min(n_dataset, n_model)): equal to the per-iteration form."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    dirs = syn.model_directions(model).copy()
    dirs[5::97] = np.nan
    H, W = model.phi.size, model.theta.size
    truth = T.transform_from_rpy((1.0, -1.5, 1.2), (0.0, 0.0, 0.5))
    est = T.mult(truth, T.transform_from_rpy((0.04, -0.03, 0.01), (0.0, 0.0, 0.012)))
    res = {}
    for mode in (0, 1, 4):
        rcc = ra.RCCHipO1Dn(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(W, H, 0.1, 100.0, (0.0, 0.0, 0.0), dirs)
        rcc.find(truth)
        mv = rcc.modelView()
        pts = (dirs * mv["ranges"].reshape(-1, 1)).astype(np.float32)     # NaN where the direction is NaN
        pts[mv["hits"].reshape(-1) == 0] = np.nan                            # ... and where the scan missed
        rcc.params.max_dist, rcc.adaptive_max_dist_min = 0.6, 0.2
        rcc.set_micp_fast(mode)
        out = []
        rcc.set_dataset(pts, None)
        out += [rcc.correct_once(est, T.identity(), 6, 0.1, False) for _ in range(3)]
        rcc.set_dataset(pts[: (H * W) // 2 + 7], None)
        out += [rcc.correct_once(est, T.identity(), 6, 0.1, False) for _ in range(3)]
        res[mode] = out
        if mode != 0:
            assert rcc.micp_fast_info()["done"] >= 2
        rcc.close()
    for mode in (1, 4):
        for (Tf, sf), (Tc, sc) in zip(res[mode], res[0]):
            assert int(sf["n_meas"]) == int(sc["n_meas"]) > 100
            _transform_close(Tf, Tc, 1e-6)
    assert int(res[1][0][1]["n_meas"]) > int(res[1][3][1]["n_meas"])        # the short dataset really is shorter


@pytest.mark.parametrize("shape", [(128, 1024, 23), (100, 1000, 23), (64, 512, 23), (64, 512, 2), (16, 900, 2), (30, 500, 2), (128, 1024, 32), (100, 1000, 32)])
def test_moments_formed_in_the_find_epilogue_equal_the_separate_pass(ra, orc, ctx, meshes, shape):
    """rmclhip_rcc_set_micp_fast 1 (the find of kind 23 forms the moments in its epilogue: f64 MFMA over the wave's 64 correspondences,
    one partial row per workgroup, mask words in tile order, rows folded by eight workgroups) against mode 3 (k_micp_moments, a pass of
    its own): the same 82 moments (1e-10 relative: the summation order differs), the same undecided correspondences, the same
    statistics and pose.  100 x 1000 has ragged tiles, lanes without a ray and workgroups without a tile; 64 x 512 (the per-ray
    traversal forced: the rule would take the quad kind) leaves 128 rows, which ONE workgroup folds; kind 2 = the quad traversal the
    rule takes for small scans (four lanes per ray, a tile per workgroup, one mask word per workgroup; 30 x 500: ragged); the room leaves
    correspondences undecided, the O1Dn model has NaN directions and an unmasked dataset."""
    from rmcl_amd import synthetic as syn, types as T
    H, W, want_kind = shape
    v, f = meshes("room100k")
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c2()
    model.phi.inc = model.phi.inc * 128.0 / H
    model.phi.size = H
    model.theta.inc = model.theta.inc * 1024.0 / W
    model.theta.size = W
    truth = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    est = T.mult(truth, T.transform_from_rpy((0.03, -0.02, 0.015), (0.004, -0.003, 0.008)))
    dirs = syn.model_directions(model).copy()
    dirs[11::131] = np.nan
    kinds = ("spherical", "o1dn") if (H, W) not in ((128, 1024), (16, 900)) else ("spherical", "o1dn", "pinhole", "ondn")
    for kind in kinds:
        res = {}
        for mode in (3, 1, 4):
            if kind == "pinhole":
                rcc = ra.RCCHipPinhole(hm)
                rcc.setTsb(T.identity())
                rcc.setModel(W, H, 0.1, 100.0, 600.0, 90.0, W / 2 - 0.5, H / 2 - 0.5)
                rcc.find(truth)
                mv = rcc.modelView()
                pts = mv["points"].reshape(-1, 3).copy()
                pts[mv["hits"].reshape(-1) == 0] = np.nan
                rcc.set_dataset(pts, (mv["hits"].reshape(-1) > 0).astype(np.uint8))
            elif kind == "ondn":
                rcc = ra.RCCHipOnDn(hm)
                rcc.setTsb(T.identity())
                d2 = syn.model_directions(model).copy()
                origs = (0.05 * np.stack([np.sin(np.arange(H * W) * 0.37), np.cos(np.arange(H * W) * 0.11), np.sin(np.arange(H * W) * 0.05)], -1)).astype(np.float32)
                rcc.setModel(W, H, 0.1, 100.0, origs, d2)
                rcc.find(truth)
                mv = rcc.modelView()
                pts = mv["points"].reshape(-1, 3).copy()
                pts[mv["hits"].reshape(-1) == 0] = np.nan
                rcc.set_dataset(pts, None)
            elif kind == "spherical":
                rcc = ra.RCCHipSpherical(hm)
                rcc.setTsb(T.identity())
                rcc.setModel(model)
                rcc.find(truth)
                rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
            else:
                rcc = ra.RCCHipO1Dn(hm)
                rcc.setTsb(T.identity())
                rcc.setModel(W, H, 0.1, 100.0, (0.01, -0.02, 0.03), dirs)
                rcc.find(truth)
                mv = rcc.modelView()
                pts = (dirs * mv["ranges"].reshape(-1, 1) + np.float32([0.01, -0.02, 0.03])).astype(np.float32)
                pts[mv["hits"].reshape(-1) == 0] = np.nan
                rcc.set_dataset(pts, None)
            if want_kind == 32 or (H * W <= 57344 and want_kind == 23):   # (32: the cooperative descent below the frontier, what autotune may choose)
                rcc.set_traversal(want_kind)
            assert rcc.find_variant(1) == want_kind
            rcc.params.max_dist, rcc.adaptive_max_dist_min = 0.5, 0.2
            rcc.set_micp_fast(3)
            for _ in range(3):
                rcc.correct_once(est, T.identity(), 8, 0.0, False)       # the bounds are learnt with the separate pass in both runs
            rcc.set_micp_fast(mode)
            Tc, st = rcc.correct_once(est, T.identity(), 8, 0.0, False)
            tot, rows, unc = rcc.debug_micp_moments()
            info = rcc.micp_fast_info()
            assert info["last_code"] == 0, info
            res[mode] = (Tc, st, tot, rows, unc, info["last_uncertain"])
            rcc.close()
        for other in (1, 4):     # (the host's iterations and the device loop, both behind the find's epilogue)
            (Ta, sa, ma, rows_a, unc_a, lu_a), (Tb, sb, mb, rows_b, unc_b, lu_b) = res[3], res[other]
            ntiles = -(-H // 4) * -(-W // 16)
            assert rows_b % 8 == 0 and rows_b == ((ntiles if want_kind == 2 else -(-ntiles // 4)) + 7) // 8 * 8   # one row per workgroup of the find
            assert unc_a == unc_b == lu_a == lu_b
            assert ma[0] > 1000 and ma[0] == mb[0]                           # the count of certainly gated-in correspondences: exact
            assert np.allclose(ma[:82], mb[:82], rtol=1e-10, atol=1e-7)
            assert np.all(mb[82:] == 0.0)
            assert int(sa["n_meas"]) == int(sb["n_meas"])
            _transform_close(Ta, Tb, 1e-6)
            assert np.allclose(sa["covariance"], sb["covariance"], rtol=1e-6, atol=1e-8)
    hm.release()


@pytest.mark.parametrize("fast", [1, 4])
def test_moment_form_randomised_against_per_iteration_form(ra, orc, ctx, meshes, fast):
    """120 random corrections (mesh, mount, odometry frame, perturbation from millimetres to decimetres, gate from 5 cm to
    2 m, 2..12 iterations, progress) through two operators that differ only in rmclhip_rcc_set_micp_fast: n_meas must be
    IDENTICAL (a gate decision taken from the moments that the per-iteration arithmetic takes differently would show here)
    and the pose equal to 1e-6 -- whichever exit the moment form takes."""
    from rmcl_amd import synthetic as syn, types as T
    rng = np.random.RandomState(2024)
    model = syn.model_c1()
    codes = {0: 0, 1: 0, 2: 0}
    for mesh in ("cube", "room30k"):
        v, f = meshes(mesh)
        hm = ra.import_hip_map(ctx, v, f)
        truth = T.transform_from_rpy((1.0, -1.5, 1.2), (0.0, 0.0, 0.5))
        pair = []
        for mode in (fast, 0):
            rcc = ra.RCCHipSpherical(hm)
            rcc.setModel(model)
            rcc.set_micp_fast(mode)
            pair.append(rcc)
        for case in range(60):
            Tsb = T.transform_from_rpy(tuple(rng.uniform(-0.3, 0.3, 3)), tuple(rng.uniform(-0.5, 0.5, 3)))
            Tbo = T.transform_from_rpy(tuple(rng.uniform(-0.2, 0.2, 3)), tuple(rng.uniform(-0.1, 0.1, 3)))
            scale = 10.0 ** rng.uniform(-3.0, -0.5)
            est = T.mult(truth, T.transform_from_rpy(tuple(rng.uniform(-1, 1, 3) * scale), tuple(rng.uniform(-1, 1, 3) * scale * 0.3)))
            Tom = T.mult(est, T.inv(Tbo))
            md = float(10.0 ** rng.uniform(-1.3, 0.3))
            n_iter = int(rng.randint(2, 13))
            prog = float(rng.uniform(0.0, 1.0))
            out = []
            for rcc in pair:
                rcc.setTsb(Tsb)
                if case % 7 == 0 or case == 0:
                    # a fresh measured scan every few cases (taken at the truth through this mount)
                    rcc.find(T.mult(truth, T.identity()))
                    rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
                rcc.params.max_dist, rcc.adaptive_max_dist_min = md, 0.5 * md
                out.append(rcc.correct_once(Tom, Tbo, n_iter, prog, False))
            (Tf, sf), (Tc, sc) = out
            assert int(sf["n_meas"]) == int(sc["n_meas"]), (mesh, case, md, scale, pair[0].micp_fast_info())
            _transform_close(Tf, Tc, 1e-6)
            codes[pair[0].micp_fast_info()["last_code"]] += 1
        for rcc in pair:
            rcc.close()
    assert codes[0] >= 30 and codes[1] >= 5, codes      # all the exits were taken


def test_long_loops_stay_within_tolerance(ra, orc, ctx, meshes):
    """40 iterations with non-trivial frames: the sensor-frame form of the iteration (one product per iteration instead of the
    reference's frame-by-frame conjugations) and the moment form must not drift away from the oracle's frame-by-frame loop."""
    from rmcl_amd import synthetic as syn, types as T
    g = np.load(golden_path("g5_micp_sphere20k.npz"))
    m, hm, model = _sphere_setup(ra, orc, ctx, meshes)
    Tsb, Tbo, Tom2 = (g[k].view(T.TRANSFORM)[0] for k in ("Tsb", "Tbo", "Tom2"))
    meas = m.simulate_spherical(model, Tsb, T.mult(T.identity(), Tbo), bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    To, so, _ = om.correct_once(m, model, Tsb, Tbo, Tom2, ds, mask, 40, 1.0, adaptive_min=0.15, convergence_progress=0.3, nthreads=8)
    for mode in (0, 1, 4):
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(Tsb)
        rcc.setModel(model)
        rcc.set_dataset(ds, mask)
        rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
        rcc.set_micp_fast(mode)
        for _ in range(3):
            Tg, sg = rcc.correct_once(Tom2, Tbo, 40, 0.3, False)
            _transform_close(Tg, To, 1e-5)
            assert int(sg["n_meas"]) == int(so["n_meas"])
        rcc.close()


def _caller_loop_pair(ra, hm, model, Tsb, Tbo, ds, mask, modes=(1, 0), n_iter=10):
    out = []
    for mode in modes:
        rcc = ra.RCCHipSpherical(hm)
        rcc.setModel(model)
        rcc.set_dataset(ds, mask)
        rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
        rcc.set_micp_fast(mode)
        sensor = ra.MICPSensor("lidar", rcc, Tsb=Tsb, Tbo=Tbo)
        sensor.valid_dataset_measurements = int(mask.sum())
        sensor.total_dataset_measurements = len(mask)
        out.append((rcc, ra.MICPLocalization([sensor], optimization_iterations=n_iter)))
    return out


def test_unchanged_caller_loop_is_served_from_the_moments(ra, orc, ctx, meshes):
    """The reference's own loop -- find() once, then computeCrossStatistics() + umeyama_transform per iteration on the HOST
    (micp_localization.cpp:900-964; rmcl_amd.micp.MICPLocalization.correctOnce restates it call for call) -- through the
    unchanged entry points.  Round 4: the first correction's first computeCrossStatistics runs ONE moment pass, every later call of
    that correction is answered on the host from the published moments; from the second correction on the find itself forms them
    (speculating on max_dist' within +-8 % of the last one) and NO call launches anything.  Against the same loop with the moment
    form off (a streaming reduction per call) and the oracle's frame-by-frame loop: every iteration's T_onew_oold within 1e-6 /
    1e-5, valid_matches identical, on a room with occluders (undecided correspondences, gated-out ones, misses)."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c1()
    m, hm, truth, ds0, mask0 = _room_case(ra, orc, ctx, meshes, model)
    Tsb, Tbo = syn.tsb_offset(), T.transform_from_rpy((0.3, -0.1, 0.0), (0.0, 0.0, 0.2))
    meas = m.simulate_spherical(model, Tsb, T.mult(truth, T.identity()), bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    (rcc, loc), (rcc0, loc0) = _caller_loop_pair(ra, hm, model, Tsb, Tbo, ds, mask)
    est_bm = T.mult(truth, T.transform_from_rpy((0.05, -0.03, 0.01), (0.0, 0.005, 0.02)))
    loc.Tom_ = T.mult(est_bm, T.inv(Tbo))
    loc0.Tom_ = loc.Tom_.copy()
    unc = 0
    for corr in range(6):
        Tom_before, prog = loc.Tom_.copy(), loc.convergence_progress_
        rec, rec0 = [], []
        loc.correctOnce(record=rec)
        loc0.correctOnce(record=rec0)
        To, so, _ = om.correct_once(m, model, Tsb, Tbo, Tom_before, ds, mask, 10, 1.0, adaptive_min=0.15, convergence_progress=prog, nthreads=8)
        assert loc.correction_stats_latest_["valid_matches"] == loc0.correction_stats_latest_["valid_matches"] == int(so["n_meas"])
        for a, b in zip(rec, rec0):
            _transform_close(a, b, 1e-6)
        _transform_close(rec[-1], To, 1e-5)
        assert abs(loc.convergence_progress_ - loc0.convergence_progress_) < 1e-6
        # both localisations continue from the SAME state (the comparison is per correction, not of two diverging trajectories)
        loc0.Tom_, loc0.convergence_progress_ = loc.Tom_.copy(), loc.convergence_progress_
        unc = max(unc, rcc.micp_fast_info()["last_uncertain"])
    info = rcc.ccs_info()
    assert info["calls"] == 60
    assert info["from_moments"] >= 55, info            # everything but what a band / cap miss sent to the streaming reduction
    assert info["speculative_finds"] >= 3, info        # the finds of the later corrections formed the moments themselves
    assert info["passes"] <= 4, info
    assert rcc0.ccs_info()["from_moments"] == 0
    rcc.close()
    rcc0.close()


def test_caller_loop_moment_cache_is_dropped_by_what_invalidates_it(ra, orc, ctx, meshes):
    """The published moments belong to ONE find and ONE dataset: a new dataset, a new find, a max_dist' outside the band, a
    pre-transform outside the caps and a gate so tight that more than 256 correspondences are undecided must all reach the same
    numbers the streaming reduction gives."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c1()
    m, hm, truth, ds0, mask0 = _room_case(ra, orc, ctx, meshes, model)
    ident = T.identity()
    meas = m.simulate_spherical(model, ident, truth, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc, ref = ra.RCCHipSpherical(hm), ra.RCCHipSpherical(hm)
    for r, mode in ((rcc, 1), (ref, 0)):
        r.setTsb(ident)
        r.setModel(model)
        r.set_dataset(ds, mask)
        r.params.max_dist, r.adaptive_max_dist_min = 1.0, 0.15
        r.set_micp_fast(mode)
    est = T.mult(truth, T.transform_from_rpy((0.04, -0.02, 0.01), (0.0, 0.0, 0.01)))
    small = T.transform_from_rpy((0.002, 0.001, 0.0), (0.0, 0.0, 0.001))
    big = T.transform_from_rpy((0.4, 0.0, 0.0), (0.0, 0.0, 0.1))

    def both(Tpre, prog):
        a, b = rcc.computeCrossStatistics(Tpre, prog), ref.computeCrossStatistics(Tpre, prog)
        assert int(a["n_meas"]) == int(b["n_meas"]), (int(a["n_meas"]), int(b["n_meas"]), rcc.ccs_info())
        assert np.allclose(a["covariance"], b["covariance"], rtol=2e-5, atol=2e-6)
        for k in "xyz":
            assert abs(float(a["model_mean"][k]) - float(b["model_mean"][k])) < 2e-5
        return a

    for r in (rcc, ref):
        r.find(est)
    both(ident, 0.0)                      # pass
    both(small, 0.0)                      # from the moments
    i0 = rcc.ccs_info()
    assert i0["passes"] == 1 and i0["from_moments"] == 2
    both(small, 0.5)                      # max_dist' 0.575: outside the band -> second pass
    both(small, 0.9)                      # max_dist' 0.235: outside the band again and no pass left -> streaming
    i1 = rcc.ccs_info()
    assert i1["passes"] == 2 and i1["from_moments"] == 3, i1
    both(small, 0.5)                      # (the set of the second pass still answers its own band)
    assert rcc.ccs_info()["from_moments"] == 4
    # a new dataset (shifted points): the old moments must not answer
    ds2 = (ds + np.float32([0.01, 0.0, 0.0])).astype(np.float32)
    for r in (rcc, ref):
        r.set_dataset(ds2, mask)
    both(small, 0.5)
    # a new find (speculating now: the last find was followed by calls) at another pose
    est2 = T.mult(truth, T.transform_from_rpy((-0.03, 0.02, 0.0), (0.0, 0.0, -0.015)))
    for r in (rcc, ref):
        r.find(est2)
    assert rcc.ccs_info()["speculative_finds"] == 1
    before = rcc.ccs_info()["from_moments"]
    both(ident, 0.5)
    both(small, 0.52)                     # within the band of the speculation (centre 0.575 * ...)
    assert rcc.ccs_info()["from_moments"] == before + 2 and rcc.ccs_info()["passes"] == i1["passes"] + 1
    both(big, 0.5)                        # a pre-transform outside the caps: a pass with wider caps (everything undecided) -> streaming
    # a gate at 3 cm on a 4-6 cm perturbation: most correspondences are undecided -> streaming, same numbers
    for r in (rcc, ref):
        r.params.max_dist = r.adaptive_max_dist_min = 0.03
        r.find(est)
    both(ident, 0.0)
    both(small, 0.0)
    # modelView after a speculating find is the find's own output
    mv, mv0 = rcc.modelView(), ref.modelView()
    assert np.array_equal(mv["face_ids"], mv0["face_ids"]) and np.array_equal(mv["hits"], mv0["hits"])
    assert np.array_equal(mv["points"], mv0["points"], equal_nan=True)
    rcc.close()
    ref.close()


@pytest.mark.parametrize("pert,lo,hi", [(((0.02, -0.015, 0.005), (0.0, 0.001, 0.005)), 1, 256), (((0.09, -0.05, 0.01), (0.0, 0.002, 0.016)), 257, 1024)],
                         ids=["tracking-size", "10 cm / 1 deg"])
def test_c3_full_size_room_host_forms_with_undecided_correspondences(ra, orc, ctx, meshes, pert, lo, hi):
    """C3 at full size (128 x 1024, 100 000 triangles) on the occluded room: a tracking-size correction leaves a handful of
    correspondences undecided, a 10 cm / 1 deg one several hundred (round 4: up to 1024 go to the host, summed in eight interleaved
    partial sums -- AVX2 where the CPU has it), which the HOST re-evaluates per iteration -- rmclhip_rcc_correct_once (iterations on
    the host) and the reference's unchanged caller loop (find + 10 x computeCrossStatistics, served from the published moments)
    against the oracle's frame-by-frame loop to 1e-5 and against the streaming form (moment form off) to 1e-6; n_meas identical."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room100k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c2()
    truth = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    Tsb, Tbo = syn.tsb_offset(), T.transform_from_rpy((0.3, -0.1, 0.0), (0.0, 0.0, 0.2))
    meas = m.simulate_spherical(model, Tsb, truth, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    est_bm = T.mult(truth, T.transform_from_rpy(*pert))
    Tom = T.mult(est_bm, T.inv(Tbo))
    To, so, _ = om.correct_once(m, model, Tsb, Tbo, Tom, ds, mask, 10, 1.0, adaptive_min=0.15, convergence_progress=0.2, nthreads=8)
    (rcc, loc), (rcc0, loc0) = _caller_loop_pair(ra, hm, model, Tsb, Tbo, ds, mask)
    unc = []
    for rep in range(3):       # (the first correction learns the caps; the repeats run on speculating finds)
        for L in (loc, loc0):
            L.Tom_, L.convergence_progress_ = Tom, 0.2
        rec, rec0 = [], []
        loc.correctOnce(record=rec)
        loc0.correctOnce(record=rec0)
        assert loc.correction_stats_latest_["valid_matches"] == loc0.correction_stats_latest_["valid_matches"] == int(so["n_meas"])
        for a, b in zip(rec, rec0):
            _transform_close(a, b, 1e-6)
        _transform_close(rec[-1], To, 1e-5)
    info = rcc.ccs_info()
    assert info["from_moments"] >= 28 and info["speculative_finds"] >= 2, info
    # the one-call form on the same operator
    for rep in range(3):
        Tg, sg = rcc.correct_once(Tom, Tbo, 10, 0.2, False)
        T0, s0 = rcc0.correct_once(Tom, Tbo, 10, 0.2, False)
        assert int(sg["n_meas"]) == int(s0["n_meas"]) == int(so["n_meas"])
        _transform_close(Tg, T0, 1e-6)
        _transform_close(Tg, To, 1e-5)
        fi = rcc.micp_fast_info()
        if fi["last_code"] == 0:
            unc.append(fi["last_uncertain"])
    assert unc and lo <= max(unc) <= hi, unc      # undecided correspondences were present (in this case's range) and the host took them
    assert rcc.micp_fast_info()["host_loops"] >= 2
    rcc.close()
    rcc0.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_caller_calls_in_random_order_match_the_streaming_reduction(ra, orc, ctx, meshes, seed):
    """The moment sets of round 4 live across calls (a find speculates on what the calls after the LAST find looked like), so the
    state machine is fuzzed: a seeded random sequence of find / computeCrossStatistics / new dataset / new mask / new max_dist /
    new adaptive_max_dist_min / find at far poses on two operators -- one with the moment form (default), one answering every call
    with the streaming reduction -- must give the same statistics call for call, whatever answered them."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c1()
    m, hm, truth, ds, mask = _room_case(ra, orc, ctx, meshes, model)
    ident = T.identity()
    rng = np.random.RandomState(seed)
    rcc, ref = ra.RCCHipSpherical(hm), ra.RCCHipSpherical(hm)
    for r, mode in ((rcc, 1), (ref, 0)):
        r.setTsb(ident)
        r.setModel(model)
        r.set_dataset(ds, mask)
        r.params.max_dist, r.adaptive_max_dist_min = 1.0, 0.15
        r.set_micp_fast(mode)

    def small_pose(scale):
        return T.transform_from_rpy(tuple(rng.normal(0, 0.03 * scale, 3)), tuple(rng.normal(0, 0.01 * scale, 3)))

    est = T.mult(truth, small_pose(1.0))
    for r in (rcc, ref):
        r.find(est)
    prog, n_ccs = 0.0, 0
    for step in range(70):
        op = rng.choice(["ccs", "ccs", "ccs", "ccs", "find", "find_far", "dataset", "mask", "maxd", "amin", "progress"])
        if op == "ccs":
            Tpre = small_pose(rng.choice([0.02, 0.2, 1.0, 6.0]))
            a, b = rcc.computeCrossStatistics(Tpre, prog), ref.computeCrossStatistics(Tpre, prog)
            assert int(a["n_meas"]) == int(b["n_meas"]), (step, int(a["n_meas"]), int(b["n_meas"]), rcc.ccs_info())
            assert np.allclose(a["covariance"], b["covariance"], rtol=2e-5, atol=2e-6), (step, rcc.ccs_info())
            for k in "xyz":
                assert abs(float(a["model_mean"][k]) - float(b["model_mean"][k])) < 2e-5
                assert abs(float(a["dataset_mean"][k]) - float(b["dataset_mean"][k])) < 2e-5
            n_ccs += 1
        elif op in ("find", "find_far"):
            est = T.mult(truth, small_pose(1.0 if op == "find" else 8.0))
            for r in (rcc, ref):
                r.find(est)
        elif op == "dataset":
            ds = (ds + rng.normal(0, 0.004, 3).astype(np.float32)).astype(np.float32)
            for r in (rcc, ref):
                r.set_dataset(ds, mask)
        elif op == "mask":
            mask = (mask & (rng.rand(mask.size) > 0.05).astype(mask.dtype).reshape(mask.shape)).astype(mask.dtype)
            for r in (rcc, ref):
                r.set_dataset(ds, mask)
        elif op == "maxd":
            v = float(rng.choice([0.05, 0.3, 1.0, 2.5]))
            for r in (rcc, ref):
                r.params.max_dist = v
        elif op == "amin":
            v = float(rng.choice([0.02, 0.15, 0.5]))
            for r in (rcc, ref):
                r.adaptive_max_dist_min = v
        else:
            prog = float(rng.choice([0.0, 0.3, 0.5, 0.52, 0.9, 1.0]))
    info = rcc.ccs_info()
    assert info["calls"] == n_ccs and n_ccs > 10
    assert info["from_moments"] > 0, info          # the fuzz does exercise the moment path
    assert ref.ccs_info()["from_moments"] == 0
    rcc.close()
    ref.close()


@pytest.mark.parametrize("mesh", ["cube", "room30k"])
def test_correct_once_above_262144_rays_walks_the_tree_its_tables_belong_to(ra, orc, ctx, meshes, mesh):
    """for scans above 262 144 rays the automatic rule picks kind 24 (the FILTER's tree), while the correction's find
    with the moment epilogue runs kind 23 on the MAP's tree.  The frontier table / stack bound handed to it must be those of the tree it
    walks: on a small map the two cuts number their nodes differently right below the root, so tables of the wrong tree start rays at
    wrong nodes and the correspondences are silently wrong.  256 x 2048 rays: the moment form (default) against the per-iteration form
    (plain kind-24 find + streaming reductions) and against the oracle's correspondences."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes(mesh)
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    f32 = np.float32
    H, W = 256, 2048
    model = T.spherical_model(f32(-0.6), f32(1.2 / (H - 1)), H, f32(-math.pi), f32(2 * math.pi / W), W, f32(0.1), f32(60.0))
    truth = T.transform_from_rpy((0.4, -0.3, 0.8 if mesh == "cube" else 1.4), (0.01, -0.02, 0.3))
    est = T.mult(truth, T.transform_from_rpy((0.05, -0.04, 0.03), (0.0, 0.0, 0.01)))
    ident = T.identity()
    meas = m.simulate_spherical(model, ident, truth, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(ident)
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    rcc.params.max_dist, rcc.adaptive_max_dist_min = 0.5, 0.5
    assert rcc.find_variant(1) == 24
    ref = m.simulate_spherical(model, ident, est, bvh=True, nthreads=8)
    out = {}
    for mode in (1, 0):
        rcc.set_micp_fast(mode)
        out[mode] = rcc.correct_once(est, ident, 5, 0.0, False)
        mv = rcc.modelView()    # the correspondences the correction's own find left
        assert np.array_equal(mv["hits"], ref["hits"]) and np.array_equal(mv["face_ids"], ref["face_ids"]), "mode %d" % mode
    _transform_close(out[1][0], out[0][0], 1e-5)
    assert int(out[1][1]["n_meas"]) == int(out[0][1]["n_meas"]) > 100000
    To, so, _ = om.correct_once(m, model, ident, ident, est, ds, mask, 5, 0.5, adaptive_min=0.5, nthreads=8)
    _transform_close(out[1][0], To, 1e-5)
    rcc.close()
