"""GPU parity: Correspondences*::computeCrossStatistics -> rm::statistics_p2l -> rm::umeyama_transform
(rmcl/src/rmcl/registration/CorrespondencesCPU.cpp:10-39, micp_localization.cpp:952-953) vs the oracle.

Bar: n_meas bit-exact (the f32 gate is evaluated in the same operation order), means / covariance /
pose deltas within 1e-5 relative of the oracle's double-precision reduction.
"""
import numpy as np
import pytest

import oracle_micp as om
from conftest import golden_path

pytestmark = pytest.mark.gpu


def _stats_close(s_gpu, ref64, scale=None):
    assert int(s_gpu["n_meas"]) == ref64["n_meas"]
    dm = np.array([s_gpu["dataset_mean"][k] for k in "xyz"], dtype=np.float64)
    mm = np.array([s_gpu["model_mean"][k] for k in "xyz"], dtype=np.float64)
    C = s_gpu["covariance"].astype(np.float64).reshape(3, 3)
    sc = scale if scale is not None else max(1.0, np.abs(ref64["covariance"]).max())
    assert np.allclose(dm, ref64["dataset_mean"], rtol=1e-5, atol=1e-6)
    assert np.allclose(mm, ref64["model_mean"], rtol=1e-5, atol=1e-6)
    assert np.allclose(C, ref64["covariance"], rtol=1e-5, atol=1e-5 * sc)


def _transform_close(a, b, tol=1e-5, atol_t=1e-6, atol_r=2e-7):
    """pose deltas within 1e-5 RELATIVE (north_star): |ta - tb| <= tol * |tb| and the angle of the residual rotation
    <= tol * the angle of b's rotation; the absolute floors (1e-6 m, 2e-7 rad) are the f32 resolution of the
    metre-scale means / unit quaternions both sides round through, not slack on the deltas."""
    qa = np.array([a["R"][k] for k in "xyzw"], dtype=np.float64)
    qb = np.array([b["R"][k] for k in "xyzw"], dtype=np.float64)
    qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
    if np.dot(qa, qb) < 0:
        qb = -qb
    ta = np.array([a["t"][k] for k in "xyz"], dtype=np.float64)
    tb = np.array([b["t"][k] for k in "xyz"], dtype=np.float64)
    ang_b = 2.0 * np.arctan2(np.linalg.norm(qb[:3]), abs(qb[3]))
    # residual rotation qa^-1 * qb: its vector part has norm sin(angle / 2)
    w = qa[3] * qb[3] + np.dot(qa[:3], qb[:3])
    vec = qa[3] * qb[:3] - qb[3] * qa[:3] - np.cross(qa[:3], qb[:3])
    ang_res = 2.0 * np.arctan2(np.linalg.norm(vec), abs(w))
    assert ang_res <= tol * ang_b + atol_r, (ang_res, ang_b, qa, qb)
    assert np.linalg.norm(ta - tb) <= tol * np.linalg.norm(tb) + atol_t, (ta, tb)


def _setup(ra, orc, ctx, meshes, mesh_name, model, Tsb, truth, est):
    v, f = meshes(mesh_name)
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    meas = m.simulate_spherical(model, Tsb, truth, bvh=True, nthreads=8)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(Tsb)
    rcc.setModel(model)
    rcc.set_dataset(ds, mask)
    rcc.find(est)
    sim = m.simulate_spherical(model, Tsb, est, bvh=True, nthreads=8)
    return m, rcc, ds, mask, sim


def test_golden_g3_cube(ra, orc, ctx, meshes):
    """committed fixture G3: masked dataset, non-identity Tpre, max_dist 0.8, f32 + f64 oracle values."""
    from rmcl_amd import synthetic as syn, types as T
    g = np.load(golden_path("g3_stats_cube.npz"))
    truth, est = g["truth"].view(T.TRANSFORM)[0], g["est"].view(T.TRANSFORM)[0]
    Tpre = g["Tpre"].view(T.TRANSFORM)[0]
    model, Tsb = syn.model_c1(), syn.tsb_offset()
    m, rcc, ds, mask, sim = _setup(ra, orc, ctx, meshes, "cube", model, Tsb, truth, est)
    mask = g["ds_mask"]
    rcc.set_dataset(ds, mask)
    rcc.params.max_dist = float(g["max_dist"])
    rcc.adaptive_max_dist_min = float(g["max_dist"])
    s = rcc.computeCrossStatistics(Tpre, 0.0)
    ref64 = dict(dataset_mean=g["f64_dataset_mean"], model_mean=g["f64_model_mean"], covariance=g["f64_covariance"],
                 n_meas=int(g["f64_n"]))
    _stats_close(s, ref64)
    s32 = g["stats_f32"].view(T.CROSS_STATISTICS)[0]
    assert int(s["n_meas"]) == int(s32["n_meas"])
    assert np.allclose(s["covariance"], s32["covariance"], rtol=1e-4, atol=1e-4)  # the f32 sequential merge itself drifts
    _transform_close(T.umeyama_transform(s), g["umeyama"].view(T.TRANSFORM)[0], 1e-5)


@pytest.mark.parametrize("mesh_name,model_name", [("sphere100k", "c2"), ("room30k", "c2")])
def test_full_size_reduction(ra, orc, ctx, meshes, mesh_name, model_name):
    """C2/C3 size (131 072 elements): GPU statistics vs the oracle (f64 two-pass and the reference-faithful
    f32 sequential merge), several pre-transforms and gates, then Umeyama on both."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c2()
    truth = syn.pose_c2_truth() if mesh_name == "sphere100k" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    est = T.mult(truth, syn.pose_c2_perturbation())
    Tsb = T.identity()
    m, rcc, ds, mask, sim = _setup(ra, orc, ctx, meshes, mesh_name, model, Tsb, truth, est)
    pres = [T.identity(), T.transform_from_rpy((0.05, -0.02, 0.01), (0.002, -0.001, 0.01))]
    for Tpre in pres:
        for md, amin, p in ((1.0, 0.15, 0.0), (1.0, 0.15, 0.6), (0.05, 0.05, 0.0)):
            rcc.params.max_dist, rcc.adaptive_max_dist_min = md, amin
            s = rcc.computeCrossStatistics(Tpre, p)
            md_eff = orc.adaptive_max_dist(md, amin, p)
            ref64 = orc.statistics_p2l_f64(Tpre, ds, mask, sim["points"], sim["normals"], sim["hits"], md_eff)
            assert ref64["n_meas"] > 1000
            _stats_close(s, ref64)
            # pose delta: 1e-5 against Umeyama of the double-precision statistics ...
            s64 = np.zeros((), dtype=orc.CROSS_STATISTICS)
            for i, k in enumerate("xyz"):
                s64["dataset_mean"][k] = ref64["dataset_mean"][i]
                s64["model_mean"][k] = ref64["model_mean"][i]
            s64["covariance"] = ref64["covariance"].reshape(9)
            s64["n_meas"] = ref64["n_meas"]
            _transform_close(T.umeyama_transform(s), orc.umeyama(s64), 1e-5)
            # ... and a sanity band against the reference-faithful f32 sequential merge, whose own
            # rounding drift over 131 072 merges is ~1e-4 (the reference's OpenMP / CUDA tree orders differ
            # from each other by as much)
            ref32 = orc.statistics_p2l(Tpre, ds, mask, sim["points"], sim["normals"], sim["hits"], md_eff)
            assert int(ref32["n_meas"]) == int(s["n_meas"])
            # (absolute floors widened too: the f32 drift of that reference is absolute, ~1e-6, not relative to the delta)
            _transform_close(T.umeyama_transform(s), orc.umeyama(ref32), 5e-4, atol_t=2e-5, atol_r=2e-6)


def test_linearity_of_statistics(ra, orc, ctx, meshes):
    """size-independent property: statistics of the whole scan == CrossStatistics::operator+= of the
    statistics of its two halves (dataset masks select the halves)."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c2()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    m, rcc, ds, mask, sim = _setup(ra, orc, ctx, meshes, "sphere100k", model, T.identity(), truth, est)
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
    whole = rcc.computeCrossStatistics(T.identity())
    half = np.arange(len(mask)) % 2 == 0
    rcc.set_dataset(ds, (mask * half).astype(np.uint8))
    a = rcc.computeCrossStatistics(T.identity())
    rcc.set_dataset(ds, (mask * ~half).astype(np.uint8))
    b = rcc.computeCrossStatistics(T.identity())
    merged = T.cross_statistics_merge(a, b)
    assert int(merged["n_meas"]) == int(whole["n_meas"]) == int(a["n_meas"]) + int(b["n_meas"])
    assert np.allclose(merged["covariance"], whole["covariance"], rtol=1e-5, atol=1e-5)
    for k in "xyz":
        assert abs(float(merged["dataset_mean"][k]) - float(whole["dataset_mean"][k])) < 1e-5
        assert abs(float(merged["model_mean"][k]) - float(whole["model_mean"][k])) < 1e-5


def test_empty_and_degenerate(ra, orc, ctx, meshes):
    """no valid correspondence -> n_meas 0, Umeyama identity (rm::umeyama_transform with n_meas == 0);
    a single correspondence; max_dist is a STRICT bound (MICPSensorCPU.cpp:78)."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c1()
    truth = syn.pose_c2_truth()
    m, rcc, ds, mask, sim = _setup(ra, orc, ctx, meshes, "cube", model, T.identity(), truth, truth)
    rcc.set_dataset(ds, np.zeros_like(mask))
    s = rcc.computeCrossStatistics(T.identity())
    assert int(s["n_meas"]) == 0 and not np.any(s["covariance"])
    Tu = T.umeyama_transform(s)
    assert float(Tu["R"]["w"]) == 1.0 and float(Tu["t"]["x"]) == 0.0
    one = np.zeros_like(mask)
    one[517] = 1
    rcc.set_dataset(ds, one)
    s = rcc.computeCrossStatistics(T.identity())
    assert int(s["n_meas"]) == 1 and np.allclose(s["covariance"], 0, atol=1e-6)
    # strict gate: shift the dataset along the normals by exactly d; |d| < max_dist keeps, == rejects
    rcc.set_dataset(ds, mask)
    ref = orc.statistics_p2l_f64(T.identity(), ds, mask, sim["points"], sim["normals"], sim["hits"], 1e-30)
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 1e-30
    s = rcc.computeCrossStatistics(T.identity())
    assert int(s["n_meas"]) == ref["n_meas"]


def test_dataset_from_ranges_matches_unpack_message(ra, orc, ctx, meshes):
    """rmclhip_rcc_set_dataset_from_ranges == MICPSphericalSensorCPU::unpackMessage (points = dir*range,
    mask = range in [min, max]): statistics from it equal statistics from the oracle-built dataset."""
    from rmcl_amd import synthetic as syn, types as T
    model = syn.model_c1()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    m, rcc, ds, mask, sim = _setup(ra, orc, ctx, meshes, "cube", model, syn.tsb_offset(), truth, est)
    meas = m.simulate_spherical(model, syn.tsb_offset(), truth, bvh=False)
    ranges = meas["ranges"].copy()
    ranges[::9] = 0.01      # below range.min
    ranges[5::11] = 500.0   # above range.max
    ds2, mask2 = om.dataset_from_ranges(model, ranges)
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
    rcc.set_dataset(ds2, mask2)
    a = rcc.computeCrossStatistics(T.identity())
    nv = rcc.set_dataset_from_ranges(ranges)
    assert nv == int(mask2.sum())
    b = rcc.computeCrossStatistics(T.identity())
    assert int(a["n_meas"]) == int(b["n_meas"])
    assert np.array_equal(a["covariance"], b["covariance"])
