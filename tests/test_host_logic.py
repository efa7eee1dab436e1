"""CPU tests of the product's HOST logic (rmcl_amd.micp, rmcl_amd.pf helpers, rmcl_amd.synthetic): the
correction loop is driven with a test-only correspondence operator backed by the oracle, so the loop
itself (frame conjugation, merge, weighted merge, Umeyama, composition, convergence heuristic) is checked
against the oracle-side restatement and the committed G5 trajectory without a GPU.
"""
import math

import numpy as np

import oracle_micp as om
from conftest import golden_path


class OracleCorrespondences:
    """TEST STUB with the Correspondences_ interface (setTsb / find / computeCrossStatistics), CPU oracle inside."""

    def __init__(self, orc, mesh, model, ds, mask, max_dist, adaptive_min):
        self.orc, self.mesh, self.model, self.ds, self.mask = orc, mesh, model, ds, mask
        self.max_dist, self.adaptive_min = max_dist, adaptive_min
        self.outdated = True
        self.Tsb = orc.transform()
        self.sim = None
        self.n_find = 0

    def setTsb(self, Tsb):
        self.Tsb = Tsb

    def find(self, Tbm):
        self.sim = self.mesh.simulate_spherical(self.model, self.Tsb, Tbm, bvh=True, nthreads=4)
        self.n_find += 1

    def computeCrossStatistics(self, T_snew_sold, convergence_progress=0.0):
        md = self.orc.adaptive_max_dist(self.max_dist, self.adaptive_min, convergence_progress)
        return self.orc.statistics_p2l_exact(T_snew_sold, self.ds, self.mask, self.sim["points"], self.sim["normals"],
                                             self.sim["hits"], md)


def test_correct_once_host_loop_matches_oracle_and_golden(ra, orc, meshes):
    from rmcl_amd import synthetic as syn, types as T
    g = np.load(golden_path("g5_micp_sphere20k.npz"))
    v, f = meshes("sphere20k")
    m = orc.Mesh(v, f)
    model = syn.model_vlp16_900(0.0)
    Tsb, Tbo, Tom2 = (g[k].view(T.TRANSFORM)[0] for k in ("Tsb", "Tbo", "Tom2"))
    meas = m.simulate_spherical(model, Tsb, T.mult(T.identity(), Tbo), bvh=True, nthreads=4)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    corr = OracleCorrespondences(orc, m, model, ds, mask, 1.0, 0.15)
    sensor = ra.MICPSensor("lidar", corr, Tsb=Tsb, Tbo=Tbo)
    sensor.valid_dataset_measurements = int(mask.sum())
    loc = ra.MICPLocalization([sensor], optimization_iterations=10)
    loc.Tom_ = Tom2
    loc.convergence_progress_ = 0.3
    rec = []
    loc.correctOnce(record=rec)
    assert corr.n_find == 1                                   # inner iterations do not re-raycast (App. B.4)
    traj = g["traj_frames"].view(T.TRANSFORM)
    assert np.array(rec, dtype=T.TRANSFORM).tobytes() == traj.tobytes()
    # state update: Tom' = Tom * T_onew_oold, quaternion renormalised (micp_localization.cpp:972-984)
    exp = T.mult(Tom2, rec[-1])
    assert np.allclose([loc.Tom_["t"][k] for k in "xyz"], [exp["t"][k] for k in "xyz"], atol=1e-7)
    q = np.array([loc.Tom_["R"][k] for k in "xyzw"], dtype=np.float64)
    assert abs(np.linalg.norm(q) - 1.0) < 1e-6
    # convergence heuristic (:988-1007)
    t = np.array([exp["t"][k] for k in "xyz"], dtype=np.float64)
    n_meas = loc.correction_stats_latest_["valid_matches"]
    expect = (1.0 / math.exp(10.0 * np.linalg.norm(t))) * float(exp["R"]["w"]) ** 2 * (n_meas / mask.sum())
    assert abs(loc.convergence_progress_ - expect) < 1e-6
    assert loc.correction_stats_latest_["cov_trace"] > 0


def test_two_sensors_merge_and_weights(ra, orc, meshes):
    """two sensors with different Tsb and merge weights: the merged statistics are the count-weighted merge
    of both (micp_localization.cpp:931-937); weight 0 on one sensor removes its influence from the solve."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    model = syn.model_c1()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    sensors = []
    for name, Tsb in (("a", T.identity()), ("b", syn.tsb_offset())):
        meas = m.simulate_spherical(model, Tsb, truth, bvh=False)
        ds, mask = om.dataset_from_ranges(model, meas["ranges"])
        corr = OracleCorrespondences(orc, m, model, ds, mask, 1.0, 1.0)
        s = ra.MICPSensor(name, corr, Tsb=Tsb, Tbo=T.identity())
        s.valid_dataset_measurements = int(mask.sum())
        sensors.append(s)
    loc = ra.MICPLocalization(sensors, optimization_iterations=3, adaptive_max_dist=False)
    loc.Tom_ = est
    d_both = loc.correctOnce()
    assert loc.correction_stats_latest_["valid_matches"] > 1500     # both sensors contributed
    assert loc.convergence_progress_ == 0.0                         # adaptive_max_dist off
    sensors[1].merge_weight_multiplier = 0.0
    loc2 = ra.MICPLocalization(sensors, optimization_iterations=3, adaptive_max_dist=False)
    loc2.Tom_ = est
    d_a = loc2.correctOnce()
    To, _, _ = om.correct_once(m, model, T.identity(), T.identity(), est, sensors[0].correspondences_.ds,
                               sensors[0].correspondences_.mask, 3, 1.0, nthreads=1)
    assert np.allclose([d_a["t"][k] for k in "xyz"], [To["t"][k] for k in "xyz"], atol=1e-6)
    assert not np.allclose([d_a["t"][k] for k in "xyz"], [d_both["t"][k] for k in "xyz"], atol=1e-6)
    # disable_correction: state untouched
    loc3 = ra.MICPLocalization(sensors, optimization_iterations=3, disable_correction=True)
    loc3.Tom_ = est
    loc3.correctOnce()
    assert loc3.Tom_.tobytes() == np.asarray(est).tobytes()


def test_beam_helpers(ra):
    """beams_from_points == PCDSensorUpdaterEmbree.cpp:313-327; sample_beams skips NaN points and is seeded."""
    pts = np.array([[3, 0, 4], [0, 0, 2], [np.nan, 1, 1], [1, 1, 1]], dtype=np.float32)
    b = ra.beams_from_points(pts[[0, 1]])
    assert np.allclose(b["range"], [5, 2]) and np.allclose([b["dir"]["x"][0], b["dir"]["z"][0]], [0.6, 0.8])
    assert np.allclose(b["cov"][0], np.eye(3).ravel() * 0.1) and np.all(b["orig"]["x"] == 0)
    s1, s2 = ra.sample_beams(pts, 50, seed=1), ra.sample_beams(pts, 50, seed=1)
    assert len(s1) == 50 and s1.tobytes() == s2.tobytes() and np.isfinite(s1["range"]).all()
    assert ra.sample_beams(pts, 50, seed=2).tobytes() != s1.tobytes()


def test_synthetic_generators(ra):
    from rmcl_amd import synthetic as syn
    v, f = syn.uv_sphere(100000)
    assert len(f) == 100000 and np.allclose(np.linalg.norm(v, axis=1), 10.0, atol=1e-4)
    assert syn.find_abc(50000) == (250, 200)        # the reference's factorisation rule
    v, f = syn.cube_room()
    assert len(f) == 972 and np.abs(v).max() == 5.0
    v, f = syn.noisy_room(30000)
    assert 25000 < len(f) < 36000 and f.max() < len(v)
    m = syn.model_c2()
    assert (m.phi.size, m.theta.size) == (128, 1024) and abs(m.theta.inc * 1024 - 2 * math.pi) < 1e-5
    poses, attrs = syn.uniform_particles(1000, seed=42)
    q = np.stack([poses["R"][k] for k in "xyzw"], -1)
    assert np.allclose(np.linalg.norm(q, axis=1), 1, atol=1e-6) and np.all(q[:, 0] == 0) and np.all(q[:, 1] == 0)
    assert np.all(attrs["likelihood"]["mean"] == 1.0) and np.all(attrs["likelihood"]["n_meas"] == 0)
    assert syn.uniform_particles(10, seed=42)[0].tobytes() == syn.uniform_particles(10, seed=42)[0].tobytes()


def test_device_loop_refuses_what_it_cannot_deliver(ra):
    """correctOnce(device_loop=True) returns only the final transform (rmclhip_micp_correct_once): a `record` list or more than
    8 sensors must be an error, not a silently shorter answer"""
    import pytest
    from rmcl_amd import types as T

    class _Corr:
        outdated = True

        def setTsb(self, Tsb):
            pass

    sensors = [ra.MICPSensor("s%d" % i, _Corr(), Tsb=T.identity(), Tbo=T.identity()) for i in range(9)]
    loc = ra.MICPLocalization(sensors[:1], optimization_iterations=3)
    with pytest.raises(ValueError, match="record"):
        loc.correctOnce(record=[], device_loop=True)
    loc9 = ra.MICPLocalization(sensors, optimization_iterations=3)
    with pytest.raises(ValueError, match="8 sensors"):
        loc9.correctOnce(device_loop=True)


# ---- round 4: the host half of the gate-stable moment form (rmcl_amd/csrc/micp_host.h) ---------------------------------------
def _host_moment_stats(ra, D, I, N, valid, lo, hi, rho_cap, tau_cap, Tpre, maxd):
    import ctypes as C
    from rmcl_amd import types as T
    L = ra._capi.lib()
    out = np.zeros(1, dtype=T.CROSS_STATISTICS)
    nu, cov = C.c_uint32(0), C.c_int(0)
    Tp = np.ascontiguousarray(Tpre, dtype=T.TRANSFORM).reshape(1)
    vp = None if valid is None else valid.ctypes.data
    st = L.rmclhip_host_moment_statistics(D.ctypes.data, I.ctypes.data, N.ctypes.data, vp, len(D), lo, hi, rho_cap, tau_cap,
                                          Tp.ctypes.data, maxd, out.ctypes.data, C.byref(nu), C.byref(cov))
    assert st == 0
    return out[0], nu.value, bool(cov.value)


def _random_correspondences(n, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    D = (rng.normal(size=(n, 3)) * 6.0).astype(np.float32)
    N = rng.normal(size=(n, 3))
    N = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float32)
    # model point = dataset point + an offset whose component along N is spread around the gate
    off = rng.normal(size=(n, 3)) * 0.3 + N * (rng.uniform(-1.6, 1.6, size=(n, 1)) * spread)
    I = (D + off).astype(np.float32)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    return np.ascontiguousarray(D), np.ascontiguousarray(I), np.ascontiguousarray(N), valid


def test_host_moment_form_equals_the_per_element_loop(ra, orc):
    """statistics_p2l evaluated from the 82 moments + the undecided correspondences == the oracle's per-element loop
    (MICPSensorCPU.cpp:70-84) for pre-transforms inside the caps: identical n_meas, means / covariance to f32 rounding."""
    from rmcl_amd import types as T
    D, I, N, valid = _random_correspondences(6000, 7, spread=1.0)
    rng = np.random.default_rng(11)
    maxd = 1.0
    hits = 0
    for trial in range(12):
        ang = rng.uniform(0, 0.0019)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        tr = rng.normal(size=3)
        tr = tr / np.linalg.norm(tr) * rng.uniform(0, 0.0019)
        q = np.concatenate([axis * math.sin(ang / 2), [math.cos(ang / 2)]])
        Tpre = T.transform(q=tuple(float(x) for x in q), t=tuple(float(x) for x in tr))
        s, nu, cov = _host_moment_stats(ra, D, I, N, valid, maxd, maxd, 0.002, 0.002, Tpre, maxd)
        assert cov and nu <= 256
        ref = orc.statistics_p2l_f64(Tpre, D, valid, I, N, np.ones(len(D), np.uint8), maxd)
        assert int(s["n_meas"]) == ref["n_meas"] and ref["n_meas"] > 1500
        hits += nu
        for i, k in enumerate("xyz"):
            assert abs(float(s["dataset_mean"][k]) - ref["dataset_mean"][i]) < 2e-6 * (1 + abs(ref["dataset_mean"][i]))
            assert abs(float(s["model_mean"][k]) - ref["model_mean"][i]) < 2e-6 * (1 + abs(ref["model_mean"][i]))
        assert np.allclose(s["covariance"].reshape(3, 3), ref["covariance"], rtol=2e-6, atol=2e-5)
    assert hits > 0   # the undecided path was exercised


def test_host_moment_form_band_and_caps(ra, orc):
    """a set formed for a BAND of max_dist' answers every value in the band exactly (n_meas identical to the per-element loop) and
    refuses (covered = 0) values outside it and pre-transforms outside the caps; NaN / masked-out points contribute nothing."""
    from rmcl_amd import types as T
    D, I, N, valid = _random_correspondences(1200, 3, spread=0.4)
    D[5] = np.nan
    valid[5] = 1
    Tpre = T.identity()
    lo, hi = 0.5, 0.58
    for maxd in (0.5, 0.53, 0.58):
        s, nu, cov = _host_moment_stats(ra, D, I, N, valid, lo, hi, 0.001, 0.001, Tpre, maxd)
        assert cov
        ref = orc.statistics_p2l_f64(Tpre, D, valid, I, N, np.ones(len(D), np.uint8), maxd)
        assert int(s["n_meas"]) == ref["n_meas"]
        assert np.allclose(s["covariance"].reshape(3, 3), ref["covariance"], rtol=2e-6, atol=2e-5)
    for maxd in (0.49, 0.6):
        assert not _host_moment_stats(ra, D, I, N, valid, lo, hi, 0.001, 0.001, Tpre, maxd)[2]
    far = T.transform(t=(0.05, 0.0, 0.0))
    assert not _host_moment_stats(ra, D, I, N, valid, lo, hi, 0.001, 0.001, far, 0.53)[2]
    # more undecided correspondences than the host takes: not covered
    D2, I2, N2, v2 = _random_correspondences(20000, 5, spread=0.05)
    s, nu, cov = _host_moment_stats(ra, D2, I2, N2, v2, 0.05, 0.08, 0.05, 0.05, Tpre, 0.06)
    assert nu > 256 and not cov


def test_undecided_sums_do_not_depend_on_the_cpu():
    """micp_host.h sums the undecided correspondences in eight interleaved partial sums -- one AVX2 register of floats -- and compiles
    the SAME body twice (target avx2 / portable).  The two must agree bit for bit (RMCLHIP_NO_AVX2 selects the portable one in a fresh
    process), with several hundred undecided correspondences, a count that is not a multiple of eight, and gated-out ones among them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, ctypes as C
sys.path.insert(0, %r)
import rmcl_amd as ra
from rmcl_amd import types as T
rng = np.random.RandomState(5)
n = 1500
D = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
N = rng.normal(size=(n, 3)); N = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float32)
I = (D + N * rng.uniform(-1.6, 1.6, (n, 1))).astype(np.float32)
ok = (rng.rand(n) > 0.1).astype(np.uint8)
Tp = np.ascontiguousarray(T.transform_from_rpy((0.05, -0.02, 0.01), (0.004, -0.003, 0.01)), dtype=T.TRANSFORM).reshape(1)
out = np.zeros(1, dtype=T.CROSS_STATISTICS); nu, cov = C.c_uint32(0), C.c_int(0)
st = ra._capi.lib().rmclhip_host_moment_statistics(D.ctypes.data, I.ctypes.data, N.ctypes.data, ok.ctypes.data, n, 0.95, 1.05, 0.03, 0.08,
                                                   Tp.ctypes.data, 1.0, out.ctypes.data, C.byref(nu), C.byref(cov))
print(st, nu.value, cov.value, out.tobytes().hex())
""" % root
    res = []
    for env_extra in ({}, {"RMCLHIP_NO_AVX2": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(r.stdout.strip().split())
    assert res[0] == res[1], (res[0][:3], res[1][:3])
    assert res[0][0] == "0" and res[0][2] == "1" and 200 < int(res[0][1]) <= 1024, res[0][:3]   # a few hundred undecided ones, covered
