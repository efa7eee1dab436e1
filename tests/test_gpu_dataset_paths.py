"""GPU tests of the dataset hand-over paths of Correspondences_::dataset: device-resident source buffers
(the CUDA sensors keep the scan in VRAM, MICPSphericalSensorCUDA.cpp:231-232), no mask, and O1Dn ranges."""
import numpy as np
import pytest

import oracle_micp as om

pytestmark = pytest.mark.gpu


def test_device_resident_dataset_and_no_mask(ra, orc, ctx, meshes):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, T.identity(), truth, bvh=False)
    ds, mask = om.dataset_from_ranges(model, meas["ranges"])
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
    rcc.find(est)
    rcc.set_dataset(ds, mask)
    a = rcc.computeCrossStatistics(T.identity())
    d_pts = ra.DeviceArray.from_host(ctx, ds.reshape(-1))
    d_mask = ra.DeviceArray.from_host(ctx, mask)
    rcc.set_dataset(d_pts, d_mask, device=True)
    b = rcc.computeCrossStatistics(T.identity())
    assert a.tobytes() == b.tobytes()
    rcc.set_dataset(ds, None)              # mask omitted == all valid (rm::statistics_p2l with an empty mask)
    c = rcc.computeCrossStatistics(T.identity())
    sim = m.simulate_spherical(model, T.identity(), est, bvh=False)
    r = orc.statistics_p2l_f64(T.identity(), ds, None, sim["points"], sim["normals"], sim["hits"], 1.0)
    assert int(c["n_meas"]) == r["n_meas"]
    assert np.allclose(c["covariance"].reshape(3, 3), r["covariance"], rtol=1e-5, atol=1e-6)


def test_o1dn_dataset_from_ranges_adds_origin(ra, orc, ctx, meshes):
    """MICPO1DnSensorCPU::unpackMessage (MICPO1DnSensorCPU.cpp:211-213): point = dir * range + getOrigin."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    sm = syn.model_pf16()
    dirs = syn.model_directions(sm)
    orig = (0.03, -0.02, 0.07)
    truth = syn.pose_c2_truth()
    est = T.mult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_o1dn(16, 16, 0.05, 80.0, orig, dirs, T.identity(), truth, bvh=False)
    rcc = ra.RCCHipO1Dn(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(16, 16, 0.05, 80.0, orig, dirs)
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
    nv = rcc.set_dataset_from_ranges(meas["ranges"])
    assert nv == 256
    rcc.find(est)
    s = rcc.computeCrossStatistics(T.identity())
    ds = (dirs * meas["ranges"][:, None] + np.asarray(orig, np.float32)).astype(np.float32)
    sim = m.simulate_o1dn(16, 16, 0.05, 80.0, orig, dirs, T.identity(), est, bvh=False)
    r = orc.statistics_p2l_f64(T.identity(), ds, np.ones(256, np.uint8), sim["points"], sim["normals"], sim["hits"], 1.0)
    assert int(s["n_meas"]) == r["n_meas"] > 200
    assert np.allclose(s["covariance"].reshape(3, 3), r["covariance"], rtol=1e-5, atol=1e-6)
    assert np.allclose([s["dataset_mean"][k] for k in "xyz"], r["dataset_mean"], atol=1e-5)
