"""CPU tests of the oracle itself: it is pinned against every convention the reference states in-repo
(SURVEY.md 8(c) pins 1-8; the reference has no tests or golden vectors of its own -> parity unpinned),
against independent libraries (scipy rotations, numpy SVD), against its own brute-force intersector,
and against the committed golden fixtures.
"""
import ctypes as C
import hashlib
import json
import math

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import golden_path


def _q(T):
    return np.array([T["R"][k] for k in "xyzw"], dtype=np.float64)


def _t(T):
    return np.array([T["t"][k] for k in "xyz"], dtype=np.float64)


# ---- transform algebra (pin 5: Transform = {R, t, stamp}; ~T, T*T, T*v) ---------------------------------
def test_transform_algebra_vs_scipy(orc):
    rng = np.random.RandomState(0)
    for _ in range(50):
        qa, qb = rng.normal(size=4), rng.normal(size=4)
        qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
        ta, tb, p = rng.uniform(-5, 5, 3), rng.uniform(-5, 5, 3), rng.uniform(-5, 5, 3)
        A, B = orc.transform(qa, ta), orc.transform(qb, tb)
        Ra, Rb = Rotation.from_quat(qa), Rotation.from_quat(qb)
        AB = orc.tmult(A, B)
        assert np.allclose(Rotation.from_quat(_q(AB)).as_matrix(), (Ra * Rb).as_matrix(), atol=1e-6)
        assert np.allclose(_t(AB), Ra.apply(tb) + ta, atol=1e-5)
        assert np.allclose(orc.tapply(A, p), Ra.apply(p) + ta, atol=1e-5)
        I = orc.tmult(A, orc.tinv(A))
        assert np.allclose(np.abs(_q(I)), [0, 0, 0, 1], atol=1e-6) and np.allclose(_t(I), 0, atol=1e-5)


def test_euler_convention(orc):
    q = orc.euler_to_quat(0.1, -0.2, 0.3)
    ref = Rotation.from_euler("ZYX", [0.3, -0.2, 0.1]).as_quat()
    assert np.allclose(q, ref, atol=1e-6)


def test_micp_frame_conjugation(orc):
    """MICPSensor.hpp:178: T_snew_sold = ~Tsb * T_bnew_bold * Tsb must map sensor-frame points consistently:
    Tsb * (T_snew_sold * p) == T_bnew_bold * (Tsb * p)."""
    Tsb = orc.transform_from_rpy((0.1, 0.0, 0.3), (0.0, 0.0, 0.17))
    Tb = orc.transform_from_rpy((0.02, -0.01, 0.03), (0.01, 0.02, -0.03))
    Ts = orc.tmult(orc.tmult(orc.tinv(Tsb), Tb), Tsb)
    p = np.array([1.0, -2.0, 0.5])
    assert np.allclose(orc.tapply(Tsb, orc.tapply(Ts, p)), orc.tapply(Tb, orc.tapply(Tsb, p)), atol=1e-5)


# ---- spherical model (pins 3, 4, 7) ----------------------------------------------------------------------
def test_spherical_direction_convention(orc):
    """polar2cartesian (rmcl_ros/src/util/conversions.cpp:174-188): (cos phi cos theta, cos phi sin theta,
    sin phi); phi = rows / height, theta = cols / width; buffer id = vid * width + hid."""
    from rmcl_amd import synthetic as syn
    model = syn.model_c1()
    d = orc.spherical_directions(model)
    H, W = model.phi.size, model.theta.size
    assert d.shape == (H * W, 3)
    for vid, hid in ((0, 0), (5, 7), (31, 31), (16, 0)):
        phi = model.phi.min + vid * model.phi.inc
        th = model.theta.min + hid * model.theta.inc
        exp = [math.cos(phi) * math.cos(th), math.cos(phi) * math.sin(th), math.sin(phi)]
        assert np.allclose(d[vid * W + hid], exp, atol=1e-6)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6)


# ---- ray / triangle unit cases (G1) -----------------------------------------------------------------------
def _one_tri(orc):
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    return orc.Mesh(v, np.array([[0, 1, 2]], dtype=np.uint32))


def test_g1_ray_triangle_cases(orc):
    m = _one_tri(orc)
    assert m.intersect((0.2, 0.2, 1), (0, 0, -1)) == (True, 1.0, 0)          # front hit
    assert m.intersect((0.2, 0.2, -2), (0, 0, 1)) == (True, 2.0, 0)          # back face: two-sided
    assert m.intersect((0.8, 0.8, 1), (0, 0, -1))[0] is False                # outside (u+v > 1)
    assert m.intersect((0.2, 0.2, 1), (0, 0, 1))[0] is False                 # behind the origin (t < tnear)
    assert m.intersect((0.2, 0.2, 1), (1, 0, 0))[0] is False                 # parallel: den == 0
    assert m.intersect((0.2, 0.2, 1), (0, 0, -1), 0.0, 0.5)[0] is False      # beyond tfar
    assert m.intersect((0.2, 0.2, 1), (0, 0, -1), 0.0, 1.0)[0] is True       # t == tfar accepted
    assert m.intersect((0.5, 0.0, 1), (0, 0, -1))[0] is True                 # on an edge: inclusive
    assert m.intersect((0.0, 0.0, 1), (0, 0, -1))[0] is True                 # on a vertex
    assert m.intersect((0.2, 0.2, 1), (np.nan, 0, -1), bvh=True)[0] is False  # NaN direction
    # Embree's near side is strict (absDen * tnear < T): a ray that STARTS on the triangle does not hit it
    assert m.intersect((0.2, 0.2, 0), (0, 0, -1))[0] is False
    assert m.intersect((0.2, 0.2, 0), (0.3, 0.1, 1))[0] is False
    assert m.intersect((0.2, 0.2, 1e-6), (0, 0, -1))[0] is True              # just in front of it: hit


def test_tie_break_min_t_then_min_face(orc):
    """two coplanar triangles sharing an edge, and two stacked duplicates: (min t, then min face id)."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], dtype=np.float32)
    f = np.array([[1, 3, 2], [0, 1, 2], [0, 1, 2]], dtype=np.uint32)
    m = orc.Mesh(v, f)
    for bvh in (False, True):
        assert m.intersect((0.5, 0.5, 1), (0, 0, -1), bvh=bvh) == (True, 1.0, 0)   # shared edge -> smaller id
        assert m.intersect((0.2, 0.2, 1), (0, 0, -1), bvh=bvh) == (True, 1.0, 1)   # duplicates 1 and 2 -> 1


def test_degenerate_triangles_never_hit(orc):
    v = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 1, 0]], dtype=np.float32)
    m = orc.Mesh(v, np.array([[0, 1, 2], [0, 1, 3]], dtype=np.uint32))
    assert m.intersect((0.5, 0.0, 1), (0, 0, -1)) == (True, 1.0, 1)
    assert np.array_equal(m.face_normals()[0], [0, 0, 0])


@pytest.mark.parametrize("name", ["cube", "sphere20k", "room30k"])
def test_bvh_equals_brute_force(orc, meshes, name):
    v, f = meshes(name)
    m = orc.Mesh(v, f)
    rng = np.random.RandomState(3)
    n_hit = 0
    for _ in range(400):
        O = rng.uniform(-3, 3, 3).astype(np.float32)
        O[2] = abs(O[2]) + 0.1
        D = rng.normal(size=3).astype(np.float32)
        D /= np.linalg.norm(D)
        a, b = m.intersect(O, D, 0.0, 1e4), m.intersect(O, D, 0.0, 1e4, bvh=True)
        assert a == b
        n_hit += a[0]
    assert n_hit > 200


def test_simulate_outputs_sensor_frame_and_normal_flip(orc, meshes):
    """pin 2 (outputs in the SENSOR frame, CPCEmbree.cpp:27-41) and pin 6 ((p_real - p_int) . n usable,
    scan_map_segmentation_embree.cpp:125-135): points = dir * range, normals face the sensor, and mapping
    the point to the map frame lands on the cube wall."""
    from rmcl_amd import synthetic as syn
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    model = syn.model_c1()
    Tsb, Tbm = syn.tsb_offset(), syn.pose_c2_truth()
    out = m.simulate_spherical(model, Tsb, Tbm, bvh=False)
    d = orc.spherical_directions(model)
    assert out["hits"].all()
    assert np.allclose(out["points"], d * out["ranges"][:, None], rtol=1e-6)
    assert np.all(np.einsum("ij,ij->i", d, out["normals"]) <= 1e-6)
    assert np.allclose(np.linalg.norm(out["normals"], axis=1), 1, atol=1e-5)
    Tsm = orc.tmult(Tbm, Tsb)
    pm = np.array([orc.tapply(Tsm, p) for p in out["points"][::37]])
    assert np.allclose(np.abs(pm).max(axis=1), 5.0, atol=1e-4)


def test_sphere_analytic_ranges(orc, meshes):
    """pin 8: sensor at the centre of the radius-10 sphere -> every range is 10 up to the chord sag."""
    from rmcl_amd import synthetic as syn
    v, f = meshes("sphere20k")
    m = orc.Mesh(v, f)
    # a generic sensor orientation: the Moeller-Trumbore test (like Embree's default, non-"robust" mode) is
    # not watertight at vertices, and an axis-aligned sensor at the exact centre aims rays AT mesh vertices
    Tbm = orc.transform_from_rpy((0.0, 0.0, 0.0), (0.0123, -0.0217, 0.0311))
    out = m.simulate_spherical(syn.model_c1(), orc.transform(), Tbm, bvh=True)
    assert out["hits"].all()
    assert np.all(out["ranges"] <= 10.0 + 1e-4) and np.all(out["ranges"] >= 10.0 - 0.05)


def test_miss_values(orc, meshes):
    """miss: hits 0, range = range.max + 1 (the sentinel the reference itself uses, scan_operations.cpp:36),
    NaN point / normal, face id 0xFFFFFFFF."""
    from rmcl_amd import types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    f32 = np.float32
    model = T.spherical_model(f32(0), f32(0), 1, f32(0), f32(0.1), 4, f32(0.1), f32(2.0))  # walls are 5 m away
    out = m.simulate_spherical(model, orc.transform(), orc.transform(), bvh=False)
    assert not out["hits"].any() and np.all(out["ranges"] == f32(3.0))
    assert np.isnan(out["points"]).all() and np.isnan(out["normals"]).all()
    assert np.all(out["face_ids"] == 0xFFFFFFFF)


# ---- golden fixtures ---------------------------------------------------------------------------------------
def test_golden_g2_reproduces(orc, meshes):
    from rmcl_amd import synthetic as syn, types as T
    g = np.load(golden_path("g2_cube_32x32.npz"))
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    for bvh in (False, True):
        out = m.simulate_spherical(syn.model_c1(), g["Tsb"].view(T.TRANSFORM)[0], g["Tbm"].view(T.TRANSFORM), bvh=bvh)
        for k in ("hits", "ranges", "points", "normals", "face_ids"):
            assert np.array_equal(out[k], g[k], equal_nan=True), k


def test_golden_g7_digest_reproduces(orc, meshes):
    from rmcl_amd import synthetic as syn, types as T
    with open(golden_path("g7_digests.json")) as fh:
        dig = json.load(fh)
    v, f = meshes("sphere100k")
    m = orc.Mesh(v, f)
    out = m.simulate_spherical(syn.model_c2(), T.identity(), syn.pose_c2_truth(), bvh=True, nthreads=4)
    assert hashlib.sha256(out["face_ids"].tobytes()).hexdigest() == dig["c2_sphere100k_face_ids_sha256"]
    assert hashlib.sha256(out["ranges"].tobytes()).hexdigest() == dig["c2_sphere100k_ranges_sha256"]


# ---- statistics / umeyama (pin 1) -------------------------------------------------------------------------
def test_p2l_gate_and_projection_pin(orc):
    """MICPSensorCPU.cpp:70-84: d = (Ii - Di).Ni; keep iff |d| < max_dist (strict); Mi = Di + Ni d;
    masks compared > 0; Tpre applied to the dataset point."""
    ident = orc.transform()
    D = np.array([[0, 0, 1.0], [0, 0, 1.0], [1, 1, 1], [2, 2, 2]], dtype=np.float32)
    I = np.array([[0, 0, 1.5], [0, 0, 3.0], [1, 1, 1], [2, 2, 2]], dtype=np.float32)
    N = np.array([[0, 0, 1.0]] * 4, dtype=np.float32)
    dm = np.array([1, 1, 0, 1], dtype=np.uint8)
    mm = np.array([1, 1, 1, 0], dtype=np.uint8)
    s = orc.statistics_p2l(ident, D, dm, I, N, mm, 1.0)
    assert int(s["n_meas"]) == 1                       # element 1 gated out (|d| = 2), 2 and 3 masked
    assert np.allclose([s["model_mean"][k] for k in "xyz"], [0, 0, 1.5])
    s = orc.statistics_p2l(ident, D, dm, I, N, mm, 0.5)
    assert int(s["n_meas"]) == 0                       # strict: |d| == max_dist rejected
    Tpre = orc.transform((0, 0, 0, 1), (0, 0, 0.25))
    s = orc.statistics_p2l(Tpre, D, dm, I, N, mm, 0.5)
    assert int(s["n_meas"]) == 1 and abs(float(s["dataset_mean"]["z"]) - 1.25) < 1e-6


def test_cross_statistics_merge_matches_direct(orc):
    rng = np.random.RandomState(1)
    d = rng.uniform(-3, 3, (200, 3)).astype(np.float32)
    mdl = (d * 0.9 + rng.normal(0, 0.05, d.shape)).astype(np.float32)
    parts = []
    for sl in (slice(0, 70), slice(70, 200)):
        s = orc.cs_identity()
        for i in range(sl.start, sl.stop):
            one = orc.cs_identity()
            for j, k in enumerate("xyz"):
                one["dataset_mean"][k] = d[i, j]
                one["model_mean"][k] = mdl[i, j]
            one["n_meas"] = 1
            s = orc.cs_merge(s, one)
        parts.append(s)
    s = orc.cs_merge(parts[0], parts[1])
    dm, mm = d.astype(np.float64).mean(0), mdl.astype(np.float64).mean(0)
    C = (mdl - mm).T @ (d - dm) / len(d)   # model x dataset^T, normalised by n
    assert int(s["n_meas"]) == 200
    assert np.allclose(s["covariance"].reshape(3, 3), C, atol=1e-4)
    assert np.allclose([s["dataset_mean"][k] for k in "xyz"], dm, atol=1e-5)


def test_svd3_vs_numpy(orc):
    rng = np.random.RandomState(2)
    mats = [rng.normal(size=(3, 3)) for _ in range(20)]
    mats += [np.outer(rng.normal(size=3), rng.normal(size=3)), np.zeros((3, 3)), np.diag([3.0, 3.0, 1e-12])]
    for A in mats:
        U, w, V = orc.svd3(A)
        assert np.allclose(U @ np.diag(w) @ V.T, A, atol=1e-9 * max(1, np.abs(A).max()))
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-9) and np.allclose(V.T @ V, np.eye(3), atol=1e-9)
        assert np.allclose(w, np.linalg.svd(A, compute_uv=False), atol=1e-8 * max(1, np.abs(A).max()))


def test_umeyama_recovers_transform_g4(orc):
    """G4 (convention-independent): statistics of (d, m = T d) -> umeyama == T, incl. a reflection-prone
    planar set."""
    rng = np.random.RandomState(4)
    for planar in (False, True):
        d = rng.uniform(-5, 5, (500, 3)).astype(np.float32)
        if planar:
            d[:, 2] = 0.3
        Ttrue = orc.transform_from_rpy((0.3, -0.2, 0.1), (0.05, -0.1, 0.2))
        mdl = np.array([orc.tapply(Ttrue, p) for p in d], dtype=np.float32)
        dmean, mmean = d.astype(np.float64).mean(0), mdl.astype(np.float64).mean(0)
        s = orc.cs_identity()
        for j, k in enumerate("xyz"):
            s["dataset_mean"][k], s["model_mean"][k] = dmean[j], mmean[j]
        s["covariance"] = ((mdl - mmean).T @ (d - dmean) / len(d)).reshape(9)
        s["n_meas"] = len(d)
        Tu = orc.umeyama(s)
        q0, q1 = _q(Ttrue), _q(Tu)
        assert np.allclose(q0, q1 * np.sign(np.dot(q0, q1)), atol=1e-5)
        assert np.allclose(_t(Ttrue), _t(Tu), atol=1e-4)
    empty = orc.cs_identity()
    Tu = orc.umeyama(empty)
    assert np.allclose(_q(Tu), [0, 0, 0, 1]) and np.allclose(_t(Tu), 0)


def test_cs_transform_consistency(orc):
    """Transform * CrossStatistics then umeyama == conjugated umeyama (MICPSensor.hpp:182)."""
    rng = np.random.RandomState(6)
    d = rng.uniform(-4, 4, (300, 3))
    Ttrue = orc.transform_from_rpy((0.1, 0.05, -0.07), (0.02, 0.01, -0.04))
    mdl = np.array([orc.tapply(Ttrue, p) for p in d])
    s = orc.cs_identity()
    for j, k in enumerate("xyz"):
        s["dataset_mean"][k], s["model_mean"][k] = d.mean(0)[j], mdl.mean(0)[j]
    s["covariance"] = ((mdl - mdl.mean(0)).T @ (d - d.mean(0)) / len(d)).reshape(9)
    s["n_meas"] = len(d)
    Tsb = orc.transform_from_rpy((0.1, 0.0, 0.3), (0.0, 0.0, 0.17))
    Tb = orc.umeyama(orc.cs_transform(Tsb, s))
    Texp = orc.tmult(orc.tmult(Tsb, Ttrue), orc.tinv(Tsb))
    assert np.allclose(_t(Tb), _t(Texp), atol=1e-4)
    assert np.allclose(np.abs(np.dot(_q(Tb), _q(Texp))), 1, atol=1e-6)


def test_adaptive_max_dist_formula(orc):
    assert orc.adaptive_max_dist(1.0, 0.15, 0.0) == np.float32(1.0)
    assert orc.adaptive_max_dist(1.0, 0.15, 1.0) == np.float32(0.15)
    assert abs(orc.adaptive_max_dist(1.0, 0.15, 0.5) - 0.575) < 1e-7


# ---- particle filter ---------------------------------------------------------------------------------------
def test_gaussian1d_merge_is_running_mean_variance(orc):
    x = np.random.RandomState(8).uniform(0, 1, 50).astype(np.float32)
    g = (0.0, 0.0, 0)
    for xi in x:
        g = orc.gaussian1d_add(g, (float(xi), 0.0, 1))
    assert g[2] == 50 and abs(g[0] - x.mean()) < 1e-5 and abs(g[1] - x.var()) < 1e-5


def test_evaluate_rcc_four_cases(orc, meshes):
    """PCDSensorUpdaterEmbree.cpp:49-83: sim hit/real hit -> |plane distance|; the three fixed errors."""
    from rmcl_amd import types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    params = orc.pf_params(real_hit_sim_miss_error=11.0, real_miss_sim_hit_error=22.0, real_miss_sim_miss_error=0.5)
    poses = np.array([orc.transform()], dtype=orc.TRANSFORM)
    ident = orc.transform()

    def err(direction, rng, scale=1.0):
        beams = np.zeros(1, dtype=orc.RANGE_MEASUREMENT)
        beams["dir"]["x"], beams["dir"]["y"], beams["dir"]["z"] = direction
        beams["range"] = rng
        attrs = np.zeros(1, dtype=orc.PARTICLE_ATTRIBUTES)
        mm = m if scale == 1.0 else orc.Mesh(v * scale, f)
        return float(mm.pf_update(poses, attrs, beams, ident, params, bvh=False, want_errors=True)[0, 0]), attrs

    e, a = err((1, 0, 0), 4.0)            # wall at x = 5: real hit + sim hit
    assert abs(e - 1.0) < 1e-6
    exp_eval = math.exp(-(1.0 / 4.0) / 2) / math.sqrt(2 * 4.0 * math.pi)
    assert abs(float(a["likelihood"]["mean"][0]) - exp_eval) < 1e-7 and int(a["likelihood"]["n_meas"][0]) == 1
    assert err((1, 0, 0), 90.0)[0] == 22.0          # measured range outside [0.05, 80]: real miss, sim hit
    assert err((1, 0, 0), 4.0, scale=100.0)[0] < 500  # still a hit far away (tfar = inf)
    open_v = v[np.abs(v[:, 0] - 5) > 1e-3]           # no mesh to hit -> use a ray parallel case instead
    _ = open_v
    # sim miss: shoot from outside the closed cube away from it
    poses[0]["t"]["x"] = 20.0
    assert err((1, 0, 0), 4.0)[0] == 11.0            # real hit, sim miss
    assert err((1, 0, 0), 90.0)[0] == 0.5            # real miss, sim miss


def test_evaluate_rcc_embree_geometric_normal(orc, meshes):
    """correspondence_type 2 reads Embree's un-normalised Ng (PCDSensorUpdaterEmbree.cpp:56-66): the cube's walls are two
    10 x 10 right triangles each, |Ng| = 2 * area = 100, so the 1 m plane distance becomes 100; penalties are unchanged."""
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    poses = np.array([orc.transform()], dtype=orc.TRANSFORM)
    beams = np.zeros(2, dtype=orc.RANGE_MEASUREMENT)
    beams["dir"]["x"] = 1.0
    beams["range"] = (4.0, 90.0)
    tri = v[f]
    two_area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    assert np.allclose(two_area, two_area[0])
    for bvh in (False, True):
        e = {}
        for ct in (0, 2):
            attrs = np.zeros(1, dtype=orc.PARTICLE_ATTRIBUTES)
            e[ct] = m.pf_update(poses, attrs, beams, orc.transform(), orc.pf_params(correspondence_type=ct), bvh=bvh,
                                want_errors=True)[0]
        assert abs(e[0][0] - 1.0) < 1e-6 and abs(e[2][0] - two_area[0]) < 1e-3 * two_area[0]
        assert e[0][1] == e[2][1] == 100.0


def test_golden_g6_reproduces(orc, meshes):
    from rmcl_amd import types as T
    g = np.load(golden_path("g6_pf_cube.npz"))
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    a = g["attrs_in"].view(T.PARTICLE_ATTRIBUTES).copy()
    for bvh in (False, True):
        a = g["attrs_in"].view(T.PARTICLE_ATTRIBUTES).copy()
        e = m.pf_update(g["poses"].view(T.TRANSFORM), a, g["beams"].view(T.RANGE_MEASUREMENT),
                        g["Tsb"].view(T.TRANSFORM)[0], orc.pf_params(), bvh=bvh, nthreads=2, want_errors=True)
        assert np.array_equal(e, g["errors"])
        assert np.array_equal(a.view(np.uint8), g["attrs_out"])
    assert (a["likelihood"]["n_meas"] == 10000).any()


def test_golden_g5_reproduces(orc, meshes):
    import oracle_micp as om
    from rmcl_amd import synthetic as syn, types as T
    g = np.load(golden_path("g5_micp_sphere20k.npz"))
    v, f = meshes("sphere20k")
    m = orc.Mesh(v, f)
    model = syn.model_vlp16_900(0.0)
    ident = orc.transform()
    meas = m.simulate_spherical(model, ident, ident, bvh=True, nthreads=4)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    Tom = T.transform((0, 0, 0, 1), (0.0, 0.0, 0.2))
    _, _, traj = om.correct_once(m, model, ident, ident, Tom, ds, mask, 10, 1.0, refind=False, nthreads=4)
    assert np.array_equal(np.array(traj, dtype=T.TRANSFORM).view(np.uint8), g["traj_R"])
    z = [float(T.mult(Tom, t)["t"]["z"]) for t in traj]
    assert all(z[i + 1] <= z[i] + 1e-6 for i in range(9)) and z[-1] < 0.2   # moves towards the truth


@pytest.mark.parametrize("name", ["cube", "sphere20k", "room30k"])
def test_bvh4_sse_walk_equals_brute_force_and_bvh2(orc, meshes, name):
    """use_bvh = 2 -- the 4-wide collapse of the BVH2 walked with SSE slab tests, the path bench.py times as `cpu_baseline` -- returns
    what the exhaustive loop and the scalar BVH2 walk return, bit for bit, for every model (spherical with hoisted trig tables, O1Dn
    with NaN directions, pinhole, OnDn), tiny and ragged scans, several poses and threads."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes(name)
    m = orc.Mesh(v, f)
    model = syn.model_c1()
    poses = [syn.pose_c2_truth(), T.transform_from_rpy((-2.1, 1.3, 0.7), (0.3, -0.2, 2.5))]
    Tsb = syn.tsb_offset()
    for Tbm in poses:
        ref = m.simulate_spherical(model, Tsb, Tbm, bvh=False, nthreads=4)
        for mode, nt in ((True, 1), (2, 1), (2, 4)):
            out = m.simulate_spherical(model, Tsb, Tbm, bvh=mode, nthreads=nt)
            for k in ("hits", "ranges", "points", "normals", "face_ids"):
                assert np.array_equal(out[k], ref[k], equal_nan=True), (name, mode, k)
    dirs = syn.model_directions(model).copy()
    dirs[3::17] = np.nan
    H, W = model.phi.size, model.theta.size
    a = m.simulate_o1dn(W, H, 0.1, 100.0, (0.01, -0.02, 0.03), dirs, Tsb, poses[1], bvh=False, nthreads=4)
    b = m.simulate_o1dn(W, H, 0.1, 100.0, (0.01, -0.02, 0.03), dirs, Tsb, poses[1], bvh=2, nthreads=4)
    c = m.simulate_pinhole(40, 30, 0.1, 100.0, (35.0, 35.0), (19.5, 14.5), Tsb, poses[0], bvh=False, nthreads=4)
    d = m.simulate_pinhole(40, 30, 0.1, 100.0, (35.0, 35.0), (19.5, 14.5), Tsb, poses[0], bvh=2, nthreads=4)
    for x, y in ((a, b), (c, d)):
        for k in ("hits", "ranges", "face_ids"):
            assert np.array_equal(x[k], y[k], equal_nan=True), k
    one = orc.Mesh(v[f[0]].reshape(3, 3), np.array([[0, 1, 2]], np.uint32))      # a map of ONE triangle: the BVH2 root is a leaf
    assert np.array_equal(one.simulate_spherical(model, T.identity(), poses[0], bvh=2)["face_ids"],
                          one.simulate_spherical(model, T.identity(), poses[0], bvh=False)["face_ids"])


def test_one_pass_threaded_reduction_equals_the_two_pass_value(orc):
    """orc_statistics_p2l_fast (bench.py's cpu_baseline form: one pass of raw double sums on a worker pool) against the two-pass f64
    authority, with a pre-transform, masks, a gate that rejects a third of the pairs, points 40 m from the origin (where the raw-sum form
    loses digits: it must stay within 1e-9 relative of the spread) and 1 / 3 / 8 threads; an empty selection gives Identity."""
    rng = np.random.RandomState(3)
    n = 50001
    d = (rng.normal(size=(n, 3)) * 3 + np.array([40.0, -25.0, 5.0])).astype(np.float32)
    nrm = rng.normal(size=(n, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(np.float32)
    mp = (d + nrm * rng.normal(scale=0.4, size=(n, 1))).astype(np.float32)
    dm = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    mm = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    Tpre = orc.transform((0.0005, -0.0003, 0.0008, 1.0), (0.05, -0.02, 0.01))     # (a few cm at 40 m: most pairs stay inside the gate)
    q = np.array([Tpre["R"][k] for k in "xyzw"], np.float64)
    for i, k in enumerate("xyzw"):
        Tpre["R"][k] = q[i] / np.linalg.norm(q)
    ref = orc.statistics_p2l_f64(Tpre, d, dm, mp, nrm, mm, 0.5)
    assert 0.3 * n < ref["n_meas"] < 0.8 * n
    for nt in (1, 3, 8):
        s = orc.statistics_p2l_fast(Tpre, d, dm, mp, nrm, mm, 0.5, nthreads=nt)
        assert int(s["n_meas"]) == ref["n_meas"]
        assert np.allclose([s["dataset_mean"][k] for k in "xyz"], ref["dataset_mean"], rtol=1e-6, atol=1e-6)
        assert np.allclose([s["model_mean"][k] for k in "xyz"], ref["model_mean"], rtol=1e-6, atol=1e-6)
        assert np.allclose(s["covariance"].reshape(3, 3), ref["covariance"], rtol=1e-5, atol=1e-5)
    e = orc.statistics_p2l_fast(Tpre, d, np.zeros(n, np.uint8), mp, nrm, mm, 0.5, nthreads=4)
    assert int(e["n_meas"]) == 0
