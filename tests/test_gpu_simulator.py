"""GPU parity: the rmagine-level Simulator interface through the C ABI -- bundle attribute selection (rmclhip_rcc_set_outputs),
Simulator::simulate into caller-owned bundles (rmclhip_rcc_simulate) and the free rm::statistics_p2l on caller-owned views
(rmclhip_statistics_p2l) -- against the CPU oracle.

Reference call sites: RCCEmbree.hpp:18-22 + RCCEmbree.cpp:35 (find == simulate(Tbm_est, model_buffers_), bundle {points, normals,
hits}: Correspondences.hpp:81-85), scan_map_segmentation_embree.cpp:80-87 (simulate<Bundle<Ranges, Normals>>),
lidar_corrector_embree_benchmark.cpp:117 / lidar_corrector_optix_benchmark.cpp:119 (simulate(Memory<Transform>, Bundle<Ranges>)),
CorrespondencesCUDA.cpp:28 (statistics_p2l).  Bar: hits / face ids bit-exact, floats within 1e-5 relative.
"""
import numpy as np
import pytest

from conftest import assert_close_rel, find_kinds

pytestmark = pytest.mark.gpu

ATTRS = ("hits", "ranges", "points", "normals", "face_ids")
BUNDLES = {
    "ranges_normals": ("ranges", "normals"),            # scan_map_segmentation_embree.cpp:82-85
    "micp": ("points", "normals", "hits"),              # Correspondences.hpp:81-85
    "ranges": ("ranges",),                              # lidar_corrector_embree_benchmark.cpp:95-97
    "hits_face_ids": ("hits", "face_ids"),
}


def _check_attr(name, got, ref, what):
    if name in ("hits", "face_ids"):
        assert np.array_equal(got, ref[name]), "%s: %s differ" % (what, name)
    else:
        assert_close_rel(got, ref[name], 1e-5, 0 if name == "ranges" else 1e-6, "%s %s" % (what, name))


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


@pytest.mark.parametrize("bundle", sorted(BUNDLES))
@pytest.mark.parametrize("config", ["c1_cube", "c2_sphere100k"])
def test_find_writes_only_the_selected_bundle(ra, orc, ctx, meshes, config, bundle):
    """set_outputs(subset): the selected attributes equal the oracle at the NEW pose; every deselected buffer still holds, bit for bit,
    what the find at the OLD pose left there."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube" if config == "c1_cube" else "sphere100k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1() if config == "c1_cube" else syn.model_c2()
    Tsb = syn.tsb_offset()
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(Tsb)
    rcc.setModel(model)
    old_pose = T.transform_from_rpy((-1.1, 0.7, -0.4), (0.1, -0.05, 2.0))
    new_pose = syn.pose_c2_truth()
    assert rcc.outputs() == 31
    rcc.find(old_pose)
    old = rcc.modelView()
    rcc.set_outputs(BUNDLES[bundle])
    assert rcc.outputs() == sum({"hits": 1, "ranges": 2, "points": 4, "normals": 8, "face_ids": 16}[a] for a in BUNDLES[bundle])
    rcc.find(new_pose)
    got = rcc.modelView(attributes=ATTRS)
    ref = m.simulate_spherical(model, Tsb, new_pose, bvh=(config != "c1_cube"), nthreads=8)
    for a in ATTRS:
        if a in BUNDLES[bundle]:
            _check_attr(a, got[a], ref, "%s %s" % (config, bundle))
        else:
            assert _same_bits(got[a], old[a]), "%s %s: deselected buffer %s was written" % (config, bundle, a)
    # the default read-back follows the selection
    assert sorted(k for k in rcc.modelView() if k != "mask") == sorted(BUNDLES[bundle])
    rcc.close()


@pytest.mark.parametrize("variant", find_kinds(0, 2, 23, 24, 32))
def test_selection_holds_for_every_product_kind(ra, orc, ctx, meshes, variant):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    rcc = ra.RCCHipSpherical(hm)
    rcc.set_traversal(variant)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    rcc.find(T.transform_from_rpy((1.0, 1.0, 0.5), (0, 0, 1.0)))
    old = rcc.modelView()
    rcc.set_outputs(("ranges", "normals"))
    pose = syn.pose_c2_truth()
    rcc.find(pose)
    got = rcc.modelView(attributes=ATTRS)
    ref = m.simulate_spherical(model, T.identity(), pose, bvh=False)
    for a in ("ranges", "normals"):
        _check_attr(a, got[a], ref, "kind %d" % variant)
    for a in ("hits", "points", "face_ids"):
        assert _same_bits(got[a], old[a])
    rcc.close()


def test_never_selected_attributes_have_nothing_to_hand_out(ra, ctx, meshes):
    """an operator whose bundle never carried ranges / face ids: no buffers for them (25 B/ray), download of them is an error, the
    reduction still works; deselecting part of {hits, points, normals} makes computeCrossStatistics an error, not garbage"""
    from rmcl_amd import synthetic as syn, types as T, _capi
    v, f = meshes("cube")
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    rcc.set_outputs(_capi.OUT_MICP)
    pose = syn.pose_c2_truth()
    rcc.find(pose)
    with pytest.raises(ra.RmclHipError) as e:
        rcc.modelView(attributes=("ranges",))
    assert e.value.status == _capi.ERR_INVALID
    mv = rcc.modelView()
    assert sorted(k for k in mv if k != "mask") == ["hits", "normals", "points"]
    ds = (mv["points"] + np.float32(0.01)).astype(np.float32)
    rcc.set_dataset(ds, mv["hits"])
    rcc.params.max_dist = 1.0
    st = rcc.computeCrossStatistics(T.identity(), 0.0)
    assert int(st["n_meas"]) == int(mv["hits"].sum()) > 0
    To, so = rcc.correct_once(pose, T.identity(), 3)
    assert int(so["n_meas"]) > 0
    for bad in (0, 32, 64 + 1):
        with pytest.raises(ra.RmclHipError):
            rcc.set_outputs(bad)
    rcc.set_outputs(("ranges", "normals"))
    rcc.find(pose)
    for call in (lambda: rcc.computeCrossStatistics(T.identity(), 0.0), lambda: rcc.correct_once(pose, T.identity(), 3),
                 lambda: rcc.correct_batch(np.array([pose, pose], dtype=T.TRANSFORM))):
        with pytest.raises(ra.RmclHipError) as e:
            call()
        assert e.value.status == _capi.ERR_INVALID and "deselected" in str(e.value)
    rcc.set_outputs(_capi.OUT_ALL)
    rcc.find(pose)
    assert int(rcc.computeCrossStatistics(T.identity(), 0.0)["n_meas"]) == int(st["n_meas"])
    rcc.close()


@pytest.mark.parametrize("bundle", sorted(BUNDLES))
def test_simulate_into_caller_bundle(ra, orc, ctx, meshes, bundle):
    """Simulator::simulate(Tbm, Bundle&): caller-owned device memory receives exactly the bundle's attributes; the operator's own
    model buffers, and the statistics served from them, do not change"""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    Tsb = syn.tsb_offset()
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(Tsb)
    rcc.setModel(model)
    est = syn.pose_c2_truth()
    rcc.find(est)
    own = rcc.modelView()
    rcc.set_dataset((own["points"] + np.float32(0.02)).astype(np.float32), own["hits"])
    rcc.params.max_dist = 1.0
    s0 = rcc.computeCrossStatistics(T.identity(), 0.0)
    other = T.transform_from_rpy((-2.1, 1.3, -0.7), (0.3, -0.2, 2.5))
    # guard band: the caller's buffers are 2x too large and pre-filled; only the first n elements of the bundle may change
    n = 32 * 32
    into = {}
    for a in BUNDLES[bundle]:
        k = 3 if a in ("points", "normals") else 1
        dt = {"hits": np.uint8, "face_ids": np.uint32}.get(a, np.float32)
        into[a] = ra.DeviceArray.from_host(ctx, np.full(2 * n * k, 7, dt))
    res = ra.CorrespondencesHIP.download_bundle(rcc.simulate(other, attributes=BUNDLES[bundle], into=into))
    ref = m.simulate_spherical(model, Tsb, other, bvh=False)
    for a in BUNDLES[bundle]:
        got = res[a]
        _check_attr(a, got[:n], ref, "simulate " + bundle)
        assert (got[n:] == 7).all(), "simulate wrote past the bundle's %s" % a
    after = rcc.modelView()
    for a in ATTRS:
        assert _same_bits(after[a], own[a]), "simulate touched the operator's own %s" % a
    s1 = rcc.computeCrossStatistics(T.identity(), 0.0)
    assert s1.tobytes() == s0.tobytes()
    # simulate<Bundle>(T): a fresh bundle
    fresh = ra.CorrespondencesHIP.download_bundle(rcc.simulate(other, attributes=BUNDLES[bundle]))
    for a in BUNDLES[bundle]:
        assert _same_bits(fresh[a], res[a][:n])
    rcc.close()


def test_simulate_c2_full_size_ranges_normals(ra, orc, ctx, meshes):
    """the segmentation node's bundle at config C2's size (131 072 rays, 100k triangles) against the oracle's walk on every ray"""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("sphere100k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c2()
    sim = ra.RCCHipSpherical(hm)
    sim.setTsb(T.identity())
    sim.setModel(model)
    pose = syn.pose_c2_truth()
    res = ra.CorrespondencesHIP.download_bundle(sim.simulate(pose, attributes=("ranges", "normals")))
    ref = m.simulate_spherical(model, T.identity(), pose, bvh=True, nthreads=8)
    _check_attr("ranges", res["ranges"], ref, "C2 simulate")
    _check_attr("normals", res["normals"], ref, "C2 simulate")
    sim.close()


def test_simulate_batch_host_and_device_poses(ra, orc, ctx, meshes):
    """simulate(Memory<Transform>, Bundle<Ranges>&): pose-major results; poses from the host (embree benchmark :117) and from device
    memory (optix benchmark :119) give the same bits, each pose equals the oracle, and equals find_batch's own buffers"""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("sphere20k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_pf16()
    Tsb = syn.tsb_offset()
    sim = ra.RCCHipSpherical(hm)
    sim.setTsb(Tsb)
    sim.setModel(model)
    rs = np.random.RandomState(5)
    poses = np.array([T.transform_from_rpy(tuple(rs.uniform(-3, 3, 3)), (0.0, 0.0, float(rs.uniform(-3, 3)))) for _ in range(37)],
                     dtype=T.TRANSFORM)
    n = int(model.phi.size) * int(model.theta.size)
    host = ra.CorrespondencesHIP.download_bundle(sim.simulate(poses, attributes=("ranges", "hits", "face_ids")))
    d_poses = ra.DeviceArray.from_host(ctx, poses.view(np.float32))
    dev = ra.CorrespondencesHIP.download_bundle(sim.simulate(len(poses), attributes=("ranges", "hits", "face_ids"), poses_dev=d_poses))
    for a in ("ranges", "hits", "face_ids"):
        assert _same_bits(host[a], dev[a]), "device-resident poses: %s differ" % a
    for i in (0, 11, 36):
        ref = m.simulate_spherical(model, Tsb, poses[i], bvh=False)
        for a in ("ranges", "hits", "face_ids"):
            _check_attr(a, host[a][i * n:(i + 1) * n], ref, "batch pose %d" % i)
    sim.find_batch(poses)
    own = sim.modelView(attributes=("ranges", "hits", "face_ids"))
    for a in ("ranges", "hits", "face_ids"):
        assert _same_bits(own[a], host[a])
    # one pose held in device memory
    one = ra.CorrespondencesHIP.download_bundle(sim.simulate(1, attributes=("ranges",), poses_dev=d_poses))
    assert _same_bits(one["ranges"], host["ranges"][:n])
    sim.close()


def test_simulate_every_model(ra, orc, ctx, meshes):
    """O1Dn / OnDn / pinhole simulators (RCCEmbree.hpp:36-83's protected bases) into a {ranges, normals} bundle vs the oracle"""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    Tsb = syn.tsb_offset()
    pose = syn.pose_c2_truth()
    dirs = orc.spherical_directions(model)
    origs = (np.random.RandomState(2).uniform(-0.05, 0.05, dirs.shape)).astype(np.float32)
    o1 = ra.RCCHipO1Dn(hm)
    o1.setTsb(Tsb)
    o1.setModel(32, 32, 0.1, 100.0, (0.01, -0.02, 0.03), dirs)
    r1 = ra.CorrespondencesHIP.download_bundle(o1.simulate(pose, attributes=("ranges", "normals")))
    ref1 = m.simulate_o1dn(32, 32, 0.1, 100.0, (0.01, -0.02, 0.03), dirs, Tsb, pose, bvh=False)
    on = ra.RCCHipOnDn(hm)
    on.setTsb(Tsb)
    on.setModel(32, 32, 0.1, 100.0, origs, dirs)
    rn = ra.CorrespondencesHIP.download_bundle(on.simulate(pose, attributes=("ranges", "normals")))
    refn = m.simulate_ondn(32, 32, 0.1, 100.0, origs, dirs, Tsb, pose, bvh=False)
    ph = ra.RCCHipPinhole(hm)
    ph.setTsb(Tsb)
    ph.setModel(32, 32, 0.1, 100.0, 20.0, 20.0, 15.5, 15.5)
    rp = ra.CorrespondencesHIP.download_bundle(ph.simulate(pose, attributes=("ranges", "normals")))
    refp = m.simulate_pinhole(32, 32, 0.1, 100.0, (20.0, 20.0), (15.5, 15.5), Tsb, pose, bvh=False)
    for got, ref, what in ((r1, ref1, "o1dn"), (rn, refn, "ondn"), (rp, refp, "pinhole")):
        _check_attr("ranges", got["ranges"], ref, what)
        _check_attr("normals", got["normals"], ref, what)
    for op in (o1, on, ph):
        op.close()


def test_simulate_edge_cases(ra, ctx, meshes):
    from rmcl_amd import synthetic as syn, types as T, _capi
    v, f = meshes("cube")
    hm = ra.import_hip_map(ctx, v, f)
    sim = ra.RCCHipSpherical(hm)
    # no model yet: a no-op like RCCOptix.cpp:30-34's find
    sim._model_shape = (0, 0)
    assert sim.simulate(syn.pose_c2_truth(), attributes=("ranges",)) is not None
    sim.setModel(syn.model_c1())
    sim.setTsb(T.identity())
    # zero poses: nothing written
    d = ra.DeviceArray.from_host(ctx, np.full(1024, 3.0, np.float32))
    sim.simulate(np.zeros(0, dtype=T.TRANSFORM), attributes=("ranges",), into={"ranges": d})
    assert (d.download() == 3.0).all()
    # a bundle without attributes is an error
    with pytest.raises(ra.RmclHipError) as e:
        sim.simulate(syn.pose_c2_truth(), attributes=())
    assert e.value.status == _capi.ERR_INVALID
    # every ray misses (sensor far outside, looking away is impossible for a full sphere: use a tiny range)
    far = _capi.SphericalModel.from_buffer_copy(bytes(syn.model_c1()))
    far.range.max = 0.2
    sim.setModel(far)
    res = ra.CorrespondencesHIP.download_bundle(sim.simulate(T.identity(), attributes=ATTRS))
    assert not res["hits"].any() and (res["face_ids"] == 0xFFFFFFFF).all()
    assert np.allclose(res["ranges"], np.float32(0.2) + np.float32(1.0)) and np.isnan(res["points"]).all() and np.isnan(res["normals"]).all()
    sim.close()


@pytest.mark.parametrize("n_elems", [1, 63, 1024, 131072])
def test_free_statistics_p2l_on_caller_views(ra, orc, ctx, n_elems):
    """rm::statistics_p2l(Tpre, dataset, model, params) on device memory the caller owns, vs the f64 two-pass oracle on the same f32
    inputs; with and without masks"""
    from rmcl_amd import types as T
    rs = np.random.RandomState(100 + n_elems % 97)
    mp = rs.uniform(-10, 10, (n_elems, 3)).astype(np.float32)
    mn = rs.normal(size=(n_elems, 3))
    mn = (mn / np.linalg.norm(mn, axis=1, keepdims=True)).astype(np.float32)
    dp = (mp + rs.normal(scale=0.3, size=(n_elems, 3))).astype(np.float32)
    dm = (rs.uniform(size=n_elems) > 0.2).astype(np.uint8)
    mm = (rs.uniform(size=n_elems) > 0.1).astype(np.uint8)
    Tpre = T.transform_from_rpy((0.05, -0.02, 0.01), (0.01, 0.02, -0.03))
    d = {k: ra.DeviceArray.from_host(ctx, a) for k, a in dict(dp=dp, dm=dm, mp=mp, mn=mn, mm=mm).items()}
    for use_dm, use_mm in ((True, True), (False, True), (True, False), (False, False)):
        got = ra.statistics_p2l(ctx, Tpre, d["dp"], d["dm"] if use_dm else None, d["mp"], d["mn"], d["mm"] if use_mm else None, n_elems, 0.5)
        ref = orc.statistics_p2l_f64(Tpre, dp, dm if use_dm else np.ones(n_elems, np.uint8), mp, mn, mm if use_mm else np.ones(n_elems, np.uint8), 0.5)
        assert int(got["n_meas"]) == ref["n_meas"]
        if ref["n_meas"] == 0:
            continue
        assert np.allclose([got["dataset_mean"][k] for k in "xyz"], ref["dataset_mean"], rtol=1e-5, atol=1e-6)
        assert np.allclose([got["model_mean"][k] for k in "xyz"], ref["model_mean"], rtol=1e-5, atol=1e-6)
        assert np.allclose(got["covariance"].reshape(3, 3), ref["covariance"], rtol=1e-5, atol=1e-5)
    # n == 0: CrossStatistics::Identity()
    z = ra.statistics_p2l(ctx, Tpre, d["dp"], None, d["mp"], d["mn"], None, 0, 0.5)
    assert int(z["n_meas"]) == 0 and not z["covariance"].any()


def test_free_statistics_p2l_equals_the_operator(ra, orc, ctx, meshes):
    """CorrespondencesCUDA::computeCrossStatistics written out (CorrespondencesCUDA.cpp:9-30): watch(dataset), a model view over the
    operator's buffers, the interpolated max_dist, the free function -- the operator's own answer, and a simulate()d bundle as the model"""
    from rmcl_amd import synthetic as syn, types as T, _capi
    import ctypes as C
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    Tsb = syn.tsb_offset()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, Tsb, truth, bvh=False)
    ds = (orc.spherical_directions(model) * meas["ranges"][:, None]).astype(np.float32)
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(Tsb)
    rcc.setModel(model)
    rcc.set_micp_fast(0)            # the operator's streaming reduction: the same kernels, so the same bits
    d_ds, d_mask = ra.DeviceArray.from_host(ctx, ds), ra.DeviceArray.from_host(ctx, meas["hits"])
    rcc.set_dataset(d_ds, d_mask, device=True)
    rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
    rcc.find(est)
    p = 0.3
    Tpre = T.transform_from_rpy((0.01, -0.02, 0.005), (0.001, 0.002, -0.003))
    s_op = rcc.computeCrossStatistics(Tpre, p)
    hits, ranges, points, normals, fids = (C.c_void_p() for _ in range(5))
    nn = C.c_uint32(0)
    _capi.check(_capi.lib().rmclhip_rcc_device_views(rcc._h, C.byref(hits), C.byref(ranges), C.byref(points), C.byref(normals), C.byref(fids), C.byref(nn)))
    maxd = np.float32(np.float64(np.float32(1.0)) * (1.0 - p) + np.float64(np.float32(0.15)) * p)
    s_free = ra.statistics_p2l(ctx, Tpre, d_ds, d_mask, points.value, normals.value, hits.value, nn.value, float(maxd))
    assert s_free.tobytes() == s_op.tobytes()
    mv = rcc.modelView()
    ref = orc.statistics_p2l_f64(Tpre, ds, meas["hits"], mv["points"], mv["normals"], mv["hits"], float(maxd))
    assert int(s_free["n_meas"]) == ref["n_meas"] > 500
    assert np.allclose(s_free["covariance"].reshape(3, 3), ref["covariance"], rtol=1e-5, atol=1e-6)
    # the model side from a simulate()d MICP bundle at the same pose: the same correspondences, hence the same statistics
    b = rcc.simulate(est, attributes=("points", "normals", "hits"))
    s_b = ra.statistics_p2l(ctx, Tpre, d_ds, d_mask, b["points"], b["normals"], b["hits"], nn.value, float(maxd))
    assert s_b.tobytes() == s_op.tobytes()
    rcc.close()


def test_find_async_fn_after_close_raises(ra, ctx, meshes):
    """the pre-bound callable keeps its operator alive and refuses a closed handle"""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    hm = ra.import_hip_map(ctx, v, f)
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_c1())
    fn = rcc.find_async_fn(syn.pose_c2_truth())
    rcc.find_batch(np.array([syn.pose_c2_truth()] * 2, dtype=T.TRANSFORM))
    assert rcc._last_nposes == 2
    fn()
    rcc.sync()
    assert rcc._last_nposes == 1           # set by the call, not by the creation
    assert rcc.modelView()["hits"].size == 1024
    rcc.close()
    with pytest.raises(RuntimeError):
        fn()
    # a callable made from a temporary keeps the temporary alive
    fn2 = ra.RCCHipSpherical(hm).find_async_fn(syn.pose_c2_truth())
    import gc
    gc.collect()
    fn2()                                  # no model: a no-op inside the library, but on a live handle
