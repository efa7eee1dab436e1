"""Property tests for the rmagine conventions the reference does NOT restate in its own tree (SURVEY.md App. A,
marked "recollection of upstream").  Parity is unpinned for these (rmagine / Embree are not
buildable here), so each test states what the MICP / particle-filter RESULT does when the convention is the other
way round: either "invariant" (a wrong recollection cannot change the drop-in's output) or a documented variant
(the exact transformation of the result a maintainer would see).

CPU only: the oracle (the spec the HIP kernels are tested against bit-for-bit) and the host algebra of the C ABI.
The GPU twins live in tests/test_gpu_conventions.py.
"""
import numpy as np


def _rand_stats(T, rng, n=500):
    s = np.zeros((), T.CROSS_STATISTICS)
    for k in "xyz":
        s["dataset_mean"][k], s["model_mean"][k] = rng.uniform(-3, 3, 2)
    A = rng.normal(size=(3, 3))
    s["covariance"] = (A + 2.0 * np.eye(3)).astype(np.float32).reshape(-1)
    s["n_meas"] = n
    return s


def _q(Tm):
    return np.array([Tm["R"][k] for k in "xyzw"], dtype=np.float64)


def _t(Tm):
    return np.array([Tm["t"][k] for k in "xyz"], dtype=np.float64)


def _same_rotation(qa, qb, tol=1e-6):
    return min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)) < tol


def test_covariance_normalisation_does_not_change_the_solve(ra, orc):
    """App. A: "covariance is normalised by n".  If rmagine kept raw sums instead, umeyama_transform of ONE
    statistics would return the same pose delta: the optimal rotation is invariant to a positive scale of the
    cross-covariance, and t = m_mean - R d_mean does not read it (micp_localization.cpp:952-953)."""
    T = ra.types
    rng = np.random.RandomState(1)
    for _ in range(20):
        s = _rand_stats(T, rng)
        ref = orc.umeyama(s)
        for k in (1.0 / 500.0, 500.0, 7.3):
            s2 = s.copy()
            s2["covariance"] = (s["covariance"].astype(np.float64) * k).astype(np.float32)
            for solver in (orc.umeyama, T.umeyama_transform):
                got = solver(s2)
                assert _same_rotation(_q(got), _q(ref), 5e-6)
                assert np.allclose(_t(got), _t(ref), atol=2e-5)


def test_merge_is_self_consistent_under_either_normalisation(ra, orc):
    """operator+= (micp_localization.cpp:918-937) with normalised covariances == statistics of the concatenated
    sample; the same identity written for RAW sums gives the same umeyama result, i.e. a consistent library of
    either kind yields the same merged pose delta."""
    T = ra.types
    rng = np.random.RandomState(2)
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    R *= np.sign(np.linalg.det(R))
    d1, d2 = rng.normal(size=(300, 3)), rng.normal(size=(200, 3)) + 1.0
    m1, m2 = d1 @ R.T + 0.3, d2 @ R.T + 0.3

    def stats(d, m):
        s = np.zeros((), T.CROSS_STATISTICS)
        dm, mm = d.mean(0), m.mean(0)
        for i, k in enumerate("xyz"):
            s["dataset_mean"][k], s["model_mean"][k] = dm[i], mm[i]
        s["covariance"] = ((m - mm).T @ (d - dm) / len(d)).astype(np.float32).reshape(-1)
        s["n_meas"] = len(d)
        return s

    merged = orc.cs_merge(stats(d1, m1), stats(d2, m2))
    direct = stats(np.vstack([d1, d2]), np.vstack([m1, m2]))
    assert np.allclose(merged["covariance"], direct["covariance"], atol=1e-5)
    # raw-sum flavour of the same merge: S = n * C; S12 = S1 + S2 + n1 (m1-m)(d1-d)^T + n2 (m2-m)(d2-d)^T
    raw = direct.copy()
    raw["covariance"] = (direct["covariance"].astype(np.float64) * 500).astype(np.float32)
    assert _same_rotation(_q(orc.umeyama(raw)), _q(orc.umeyama(merged)), 5e-6)
    assert np.allclose(_t(orc.umeyama(raw)), _t(orc.umeyama(merged)), atol=2e-5)


def test_covariance_orientation_variant_is_the_inverse_rotation(ra, orc):
    """App. A: covariance oriented model x dataset^T.  DOCUMENTED VARIANT: if rmagine stored the transpose
    (dataset x model^T) and this build consumed it unchanged, umeyama_transform would return the INVERSE rotation
    (R -> R^T) -- not a silent small error: the G5 sphere scenario then diverges instead of converging, so the
    convention is pinned end to end by test_umeyama_recovers_transform_g4 / the G5 trajectory."""
    T = ra.types
    rng = np.random.RandomState(3)
    for _ in range(10):
        s = _rand_stats(T, rng)
        st = s.copy()
        st["covariance"] = s["covariance"].reshape(3, 3).T.reshape(-1)
        for solver in (orc.umeyama, T.umeyama_transform):
            q, qt = _q(solver(s)), _q(solver(st))
            q_inv = q * np.array([-1, -1, -1, 1])
            assert _same_rotation(qt, q_inv, 5e-6)


def test_miss_sentinel_and_miss_payload_never_reach_the_statistics(orc, meshes):
    """App. A: miss => hits 0, range = range.max + 1, NaN point / normal.  Whatever rmagine writes for a miss, the
    reduction reads model entries only where model.mask > 0 (MICPSensorCPU.cpp:73): statistics are bit-identical
    when the miss payload is replaced by arbitrary finite garbage."""
    from rmcl_amd import synthetic as syn, types as T
    import oracle_micp as om
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    model = syn.model_c1()
    truth = T.transform_from_rpy((1.0, 2.0, 1.5), (0.0, 0.1, -0.3))
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, T.identity(), truth, bvh=True)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    sim = m.simulate_spherical(model, T.identity(), est, bvh=True)
    miss = sim["hits"] == 0
    assert miss.any()
    a = orc.statistics_p2l_exact(T.identity(), ds, mask, sim["points"], sim["normals"], sim["hits"], 1.0)
    pts, nrm = sim["points"].copy(), sim["normals"].copy()
    pts[miss] = (123.0, -45.0, 6.0)
    nrm[miss] = (0.0, 0.0, 1.0)
    b = orc.statistics_p2l_exact(T.identity(), ds, mask, pts, nrm, sim["hits"], 1.0)
    assert a.tobytes() == b.tobytes()
    # the dataset side: a measured "no return" (sentinel range.max + 1, scan_operations.cpp:36) is masked out by the
    # range test of unpackMessage, so the sentinel VALUE is irrelevant as long as it is outside [min, max]
    r2 = meas["ranges"].copy()
    out = (r2 > model.range.max) | (r2 < model.range.min)
    r2[out] = np.float32(1.0e6)
    ds2, mask2 = om.dataset_from_ranges(model, r2)
    c = orc.statistics_p2l_exact(T.identity(), ds2, mask2, sim["points"], sim["normals"], sim["hits"], 1.0)
    assert np.array_equal(mask, mask2) and a.tobytes() == c.tobytes()


def test_normal_flip_convention_does_not_change_the_statistics(orc, meshes):
    """App. A: simulated normals are flipped towards the ray.  d = (I - D).N enters as |d| < max_dist and as
    M = D + N d: both are even in N, so un-flipped normals give the same CrossStatistics (to f32 rounding: the sign
    moves through two multiplications exactly)."""
    from rmcl_amd import synthetic as syn, types as T
    import oracle_micp as om
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    model = syn.model_c1()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, T.identity(), truth, bvh=False)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    sim = m.simulate_spherical(model, T.identity(), est, bvh=False)
    a = orc.statistics_p2l_exact(T.identity(), ds, mask, sim["points"], sim["normals"], sim["hits"], 1.0)
    rng = np.random.RandomState(0)
    flip = np.where(rng.rand(len(sim["normals"])) < 0.5, -1.0, 1.0).astype(np.float32)
    b = orc.statistics_p2l_exact(T.identity(), ds, mask, sim["points"], sim["normals"] * flip[:, None], sim["hits"], 1.0)
    assert a.tobytes() == b.tobytes()


def test_particle_weight_is_independent_of_the_gaussian1d_sigma_formula(orc):
    """App. A: Gaussian1D += is "the 1-D count-weighted merge" (parity unpinned).  The particle WEIGHT is
    likelihood.mean, and for the reference's use (+= Gaussian1D{eval, 0, 1} per beam, PCDSensorUpdaterEmbree.cpp:237)
    the mean is the running average of the evals under ANY count-weighted merge -- only `sigma` depends on the
    recollected variance formula, and nothing on the hot path reads sigma (resampling.cu:108-199 uses mean and
    n_meas)."""
    rng = np.random.RandomState(4)
    evals = rng.uniform(0.0, 0.2, 300).astype(np.float32)
    L = (1.0, 0.0, 0)     # (mean, sigma, n_meas): rmcl_localization.cpp:253-254 initial attrs
    run = []
    for e in evals:
        L = orc.gaussian1d_add(L, (float(e), 0.0, 1))
        run.append(float(L[0]))
    expect = np.cumsum(evals.astype(np.float64)) / np.arange(1, len(evals) + 1)
    assert np.allclose(run, expect, rtol=2e-5)
    assert int(L[2]) == len(evals)


def test_rot_dist_l2norm_recollection_only_moves_n_meas(orc):
    """resampling.cu:180 takes `diff.R.l2norm()`; recollected as the 4-vector norm (~1), which makes the rotational
    forget rate the constant likelihood_forget_per_radian.  DOCUMENTED VARIANT: were it the rotation ANGLE instead,
    only the surviving particle's n_meas (the remember_rate multiplier) would differ -- winners, poses and
    likelihood.mean are computed before and independently of rot_dist."""
    from rmcl_amd import synthetic as syn
    poses, attrs = syn.uniform_particles(2000, seed=9)
    attrs["likelihood"]["mean"] = np.random.RandomState(5).uniform(0, 1, len(attrs)).astype(np.float32)
    attrs["likelihood"]["n_meas"] = 4000
    a = orc.gladiator_resample(poses, attrs, orc.gladiator_config(likelihood_forget_per_radian=0.0), 1234, 0)
    b = orc.gladiator_resample(poses, attrs, orc.gladiator_config(likelihood_forget_per_radian=0.9), 1234, 0)
    assert a[0].tobytes() == b[0].tobytes()                                   # poses
    assert np.array_equal(a[1]["likelihood"]["mean"], b[1]["likelihood"]["mean"])
    assert not np.array_equal(a[1]["likelihood"]["n_meas"], b[1]["likelihood"]["n_meas"])
