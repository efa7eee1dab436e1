"""GPU parity of closest-point correspondences: CPCEmbree::find
(rmcl/src/rmcl/registration/CPCEmbree.cpp:18-44) vs the oracle (brute-force nearest triangle, Ericson's
closest point on triangle).  Face ids bit-exact, distances / points / normals within 1e-5."""
import numpy as np
import pytest

import oracle_micp as om
from conftest import assert_close_rel

pytestmark = pytest.mark.gpu


def _cmp(gpu, ref, what):
    assert np.array_equal(gpu["hits"], ref["hits"]), what
    assert np.array_equal(gpu["face_ids"], ref["face_ids"]), what
    assert_close_rel(gpu["ranges"], ref["ranges"], 1e-5, 1e-7, what + " distances")
    assert_close_rel(gpu["points"], ref["points"], 1e-5, 1e-6, what + " points")
    assert_close_rel(gpu["normals"], ref["normals"], 1e-5, 1e-6, what + " normals")


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("mesh_name", ["cube", "room30k"])
def test_cpc_find_matches_oracle(ra, orc, ctx, meshes, mesh_name, variant):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes(mesh_name)
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    Tsb = syn.tsb_offset()
    truth = T.transform_from_rpy((0.5, -0.3, 1.2), (0.02, -0.03, 0.4))
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, Tsb, truth, bvh=True)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    ds[5] = np.nan                                # invalid point of an organised cloud
    cpc = ra.CPCHip(hm)
    cpc.set_variant(variant)          # 1: one lane per point, 2: four lanes per point
    cpc.setTsb(Tsb)
    cpc.params.max_dist = 0.3
    cpc.adaptive_max_dist_min = 0.3
    cpc.set_dataset(ds, mask)
    cpc.find(est)
    gpu = cpc.modelView()
    ref = m.cpc_find(Tsb, est, ds, 0.3, bvh=False)
    _cmp(gpu, ref, "cpc " + mesh_name)
    assert gpu["hits"].any() and (gpu["hits"] == 0).any()
    # the reduction + Umeyama on closest-point correspondences (classic ICP step)
    s = cpc.computeCrossStatistics(T.identity())
    r = orc.statistics_p2l_f64(T.identity(), ds, mask, ref["points"], ref["normals"], ref["hits"], 0.3)
    assert int(s["n_meas"]) == r["n_meas"] > 50
    assert np.allclose(s["covariance"].reshape(3, 3), r["covariance"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("variant", [1, 2])
def test_cpc_points_on_surface_and_ties(ra, orc, ctx, meshes, variant):
    """query points exactly on shared edges / vertices (distance 0, several equidistant triangles): the
    (min distance, min face id) tie-break must agree with the brute-force oracle."""
    from rmcl_amd import types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    pts = np.concatenate([v[::7], (v[f[::11, 0]] + v[f[::11, 1]]) * np.float32(0.5), np.zeros((1, 3), np.float32)]).astype(np.float32)
    cpc = ra.CPCHip(hm)
    cpc.set_variant(variant)
    cpc.setTsb(T.identity())
    cpc.params.max_dist = 10.0
    cpc.set_dataset(pts, None)
    cpc.find(T.identity())
    gpu = cpc.modelView()
    ref = m.cpc_find(T.identity(), T.identity(), pts, 10.0, bvh=False)
    _cmp(gpu, ref, "ties")
    assert np.sum(gpu["ranges"] == 0) > 50
    assert abs(gpu["ranges"][-1] - 5.0) < 1e-6     # centre of the 10 m cube room


@pytest.mark.parametrize("n_particles,n_beams", [(500, 100), (131, 9)])
def test_pf_update_with_closest_point_errors(ra, orc, ctx, meshes, n_particles, n_beams):
    """sensor_update.correspondence_type = 1 (evaluate_cpc, PCDSensorUpdaterEmbree.cpp:88-95,219-222): the beam
    error is the distance of the measured point to the surface."""
    import math
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    truth = T.transform_from_rpy((1.0, -1.5, 1.2), (0.0, 0.0, 0.5))
    cloud = m.simulate_spherical(syn.model_c1(), T.identity(), truth, bvh=True)["points"]
    beams = ra.sample_beams(cloud, n_beams, seed=7)
    poses, attrs = syn.uniform_particles(n_particles, seed=6, bb_min=(-9, -9, 0.2, 0, 0, -math.pi), bb_max=(9, 9, 3.0, 0, 0, math.pi))
    Tsb = syn.tsb_offset()
    upd = ra.PCDSensorUpdaterHip(hm)
    upd.config = T.pf_params(correspondence_type=1)
    upd.init()
    upd.setInput(beams, Tsb)
    d_poses, d_attrs = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    d_err = ra.DeviceArray(ctx, np.float32, n_particles * n_beams)
    upd.set_error_output(d_err)
    upd.update(d_poses, d_attrs)
    a_gpu, e_gpu = d_attrs.download(), d_err.download().reshape(n_particles, n_beams)
    a_ref = attrs.copy()
    e_ref = m.pf_update(poses, a_ref, beams, Tsb, orc.pf_params(correspondence_type=1), bvh=True, nthreads=8, want_errors=True)
    assert_close_rel(e_gpu, e_ref, 1e-5, 1e-6, "cpc errors")
    assert np.array_equal(a_gpu["likelihood"]["n_meas"], a_ref["likelihood"]["n_meas"])
    assert_close_rel(a_gpu["likelihood"]["mean"], a_ref["likelihood"]["mean"], 1e-5, 1e-12, "cpc mean")
    assert_close_rel(a_gpu["likelihood"]["sigma"], a_ref["likelihood"]["sigma"], 1e-4, 1e-10, "cpc sigma")
    assert e_ref.max() < 20 and e_ref.min() >= 0       # distances, never the 100 m miss penalty
    # round 4: the queries above started from the map's near grid (the default); without the seed (rmclhip_pf_set_mapping bit 8)
    # attributes and errors are the same bit for bit -- a seed only bounds the search
    upd.set_mapping(256, 0, None)
    d_attrs2 = ra.DeviceArray.from_host(ctx, attrs)
    d_err2 = ra.DeviceArray(ctx, np.float32, n_particles * n_beams)
    upd.set_error_output(d_err2)
    upd.update(d_poses, d_attrs2)
    assert d_attrs2.download().tobytes() == a_gpu.tobytes()
    assert d_err2.download().tobytes() == e_gpu.tobytes()
    upd.close()
    with pytest.raises(ra.RmclHipError):
        upd.config = T.pf_params(correspondence_type=4)
        upd.update(d_poses, d_attrs)


def test_tracking_gives_the_cold_result_bit_for_bit(ra, orc, ctx, meshes):
    """rmclhip_rcc_set_cpc_tracking: a query that starts from the triangle the point was closest to in the previous call returns
    exactly what a cold query returns -- over a sequence of poses that drift (small steps, one large jump, a repeat), with
    the dataset replaced in between (the records of the old dataset must not be used), four lanes and one lane per point."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    hm = ra.import_hip_map(ctx, v, f)
    rng = np.random.RandomState(4)
    pts_a = rng.uniform(-8, 8, (5000, 3)).astype(np.float32)
    pts_a[:, 2] = rng.uniform(0.1, 3.0, 5000)
    pts_b = (pts_a[::-1] * np.float32(0.9)).copy()
    poses = [T.transform_from_rpy((0.02 * k, -0.015 * k, 0.01 * k), (0.0, 0.0, 0.004 * k)) for k in range(6)]
    poses += [T.transform_from_rpy((2.0, -1.0, 0.5), (0.1, 0.0, 1.3)), poses[2], poses[2]]
    for variant in (2, 1):
        # cold: no tracking, seeded from the map's near grid (round 4 default); bare: no seed at all (the reference's rtcPointQuery);
        # warm: tracking (+ the grid for the points a previous call left without a record)
        cold, warm, bare = ra.CPCHip(hm), ra.CPCHip(hm), ra.CPCHip(hm)
        for c in (cold, warm, bare):
            c.set_variant(variant)
            c.setTsb(syn.tsb_offset())
            c.params.max_dist = 0.4
        cold.set_tracking(False)
        bare.set_tracking(False)
        bare.set_grid(False)
        for pts in (pts_a, pts_b, pts_a):
            for c in (cold, warm, bare):
                c.set_dataset(pts, None)
            for P in poses:
                for c in (cold, warm, bare):
                    c.find(P)
                a, b, z = cold.modelView(), warm.modelView(), bare.modelView()
                for k in ("hits", "ranges", "face_ids", "points", "normals"):
                    assert a[k].tobytes() == b[k].tobytes() == z[k].tobytes(), (variant, k)
        for c in (cold, warm, bare):
            c.close()
    # points far outside the map's box (clamped to the nearest cell) and NaN points through the grid seed
    far = (pts_a * np.float32(40.0)).astype(np.float32)
    far[::7] = np.nan
    cold, bare = ra.CPCHip(hm), ra.CPCHip(hm)
    bare.set_grid(False)
    for c in (cold, bare):
        c.set_tracking(False)
        c.setTsb(syn.tsb_offset())
        c.params.max_dist = 0.4
        c.set_dataset(far, None)
        c.find(poses[3])
    a, z = cold.modelView(), bare.modelView()
    for k in ("hits", "ranges", "face_ids", "points", "normals"):
        assert a[k].tobytes() == z[k].tobytes(), k
    cold.close()
    bare.close()


def test_bounded_search_keeps_every_hit_bit_for_bit(ra, orc, ctx, meshes):
    """rmclhip_rcc_set_cpc_bounded: with the search limited to params.max_dist the hit mask is the unbounded one and every hit
    point carries the unbounded answer bit for bit; points with no surface within max_dist carry NaN / 0xFFFFFFFF.  Three
    gates (most points out, half, all in), tracking on and off, four lanes and one lane per point, points exactly at
    max_dist from a wall."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")           # walls at +-5
    hm = ra.import_hip_map(ctx, v, f)
    rng = np.random.RandomState(9)
    pts = rng.uniform(-4.9, 4.9, (6000, 3)).astype(np.float32)
    pts[:64] = np.float32(0.0)
    pts[:64, 0] = np.float32(5.0) - np.float32(0.25)      # exactly 0.25 from the wall x = 5 (before the pose)
    Tsb = T.identity()
    poses = [T.identity(), T.transform_from_rpy((0.01, -0.02, 0.015), (0.0, 0.0, 0.003)), T.identity()]
    for variant in (2, 1):
        for tracking in (True, False):
            for md in (0.25, 1.0, 20.0):
                ref, bnd = ra.CPCHip(hm), ra.CPCHip(hm)
                for c in (ref, bnd):
                    c.set_variant(variant)
                    c.setTsb(Tsb)
                    c.params.max_dist = md
                    c.set_tracking(tracking)
                    c.set_dataset(pts, None)
                bnd.set_bounded(True)
                for P in poses:
                    ref.find(P)
                    bnd.find(P)
                    a, b = ref.modelView(), bnd.modelView()
                    assert a["hits"].tobytes() == b["hits"].tobytes(), (variant, tracking, md)
                    hit = a["hits"].reshape(-1) > 0
                    for k in ("ranges", "face_ids", "points", "normals"):
                        ak, bk = a[k].reshape(len(hit), -1), b[k].reshape(len(hit), -1)
                        assert ak[hit].tobytes() == bk[hit].tobytes(), (variant, tracking, md, k)
                    # beyond the gate: either the unbounded answer (found inside the slightly inflated bound) or "not found"
                    out = ~hit
                    nf = b["face_ids"].reshape(-1)[out] == 0xFFFFFFFF
                    assert np.isnan(b["points"].reshape(-1, 3)[out][nf]).all() and np.isnan(b["ranges"].reshape(-1)[out][nf]).all()
                    same = a["face_ids"].reshape(-1)[out][~nf] == b["face_ids"].reshape(-1)[out][~nf]
                    assert same.all()
                    if md == 0.25 and P is poses[0]:
                        assert hit[:64].all()                       # d == max_dist is a hit (<=) in both forms
                    if md == 20.0:
                        assert hit.all() and not nf.any()
                    if md == 0.25:
                        assert nf.sum() > 0.5 * out.sum() > 0       # the bound really cut the search
                ref.close()
                bnd.close()


def test_operators_of_one_map_from_several_threads(ra, orc, ctx, meshes):
    """include/rmclhip.h: a handle is thread-compatible, operators of ONE map may be used from different threads at the same time.
    Six threads, each with its own ray-casting operator, closest-point operator and particle-filter updater (closest-point mode) on
    a shared map whose near grids do not exist yet -- so the first queries of several threads race for the grid build (slot 0 for
    the scan points, slot 1 for the filter) -- must each reproduce what the same calls give one after the other."""
    import threading
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    model = syn.model_c1()
    truth = T.transform_from_rpy((1.0, -1.5, 1.2), (0.01, -0.02, 0.5))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::16] * np.float32(3.0))
    poses, attrs = syn.uniform_particles(200, seed=5, bb_min=(-8, -8, 0.3, 0, 0, -3.14), bb_max=(8, 8, 2.5, 0, 0, 3.14))

    def work(hm, k, out):
        est = T.mult(truth, T.transform_from_rpy((0.02 * k, -0.01 * k, 0.0), (0.0, 0.0, 0.01 * k)))
        rcc = ra.RCCHipSpherical(hm); rcc.setTsb(T.identity()); rcc.setModel(model)
        res = []
        for _ in range(3):
            rcc.find(est)
            mv = rcc.modelView()
            cpc = ra.CPCHip(hm); cpc.setTsb(T.identity()); cpc.params.max_dist = 1.0
            cpc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
            cpc.find(T.mult(est, T.transform_from_rpy((0.05, 0.0, 0.0), (0.0, 0.0, 0.02))))
            cv = cpc.modelView()
            upd = ra.PCDSensorUpdaterHip(hm)
            upd.config = T.pf_params(correspondence_type=1)
            upd.init(); upd.setInput(beams, T.identity())
            d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
            upd.update(d_p, d_a)
            res.append((mv["face_ids"].tobytes(), mv["ranges"].tobytes(), cv["face_ids"].tobytes(), cv["points"].tobytes(), d_a.download().tobytes()))
            upd.close(); cpc.close()
        rcc.close()
        out[k] = res

    serial, threaded = {}, {}
    hm1 = ra.import_hip_map(ctx, v, f)
    for k in range(6):
        work(hm1, k, serial)
    hm2 = ra.import_hip_map(ctx, v, f)      # a fresh map: no grid yet
    errs = []

    def guarded(k):
        try:
            work(hm2, k, threaded)
        except Exception as e:   # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=guarded, args=(k,)) for k in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(6):
        assert threaded[k] == serial[k], k
