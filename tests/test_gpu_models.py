"""GPU parity of the two remaining sensor models of RCCEmbree.{hpp,cpp}: RCCEmbreePinhole::find (:39-68) and
RCCEmbreeOnDn::find (:102-130) -- the same traversal with a different ray generator -- incl. the dataset
construction from ranges (unpackMessage adds getOrigin for every model except the spherical one)."""
import math

import numpy as np
import pytest

from test_gpu_find import _compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [23, 0, 2, 24, 32])
def test_pinhole_depth_camera(ra, orc, ctx, meshes, variant):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    W, H, fx, fy, cx, cy = 160, 120, 131.25, 131.25, 79.5, 59.5
    Tsb = syn.tsb_offset()
    Tbm = T.transform_from_rpy((-2.0, 1.5, 1.4), (0.01, 0.1, 0.9))
    rcc = ra.RCCHipPinhole(hm)
    rcc.set_traversal(variant)
    rcc.setTsb(Tsb)
    rcc.setModel(W, H, 0.3, 12.0, fx, fy, cx, cy)
    rcc.find(Tbm)
    gpu = rcc.modelView()
    ref = m.simulate_pinhole(W, H, 0.3, 12.0, (fx, fy), (cx, cy), Tsb, Tbm, bvh=True, nthreads=4)
    _compare(gpu, ref, "pinhole")
    assert (gpu["hits"] == 0).any() and (gpu["hits"] == 1).any()     # range limit 12 m clips far walls
    # dataset from ranges == dir * range, statistics agree with the oracle-built dataset
    dirs = orc.pinhole_directions(W, H, (fx, fy), (cx, cy))
    truth = T.transform_from_rpy((-2.1, 1.45, 1.42), (0.0, 0.09, 0.93))
    meas = m.simulate_pinhole(W, H, 0.3, 12.0, (fx, fy), (cx, cy), Tsb, truth, bvh=True, nthreads=4)
    nv = rcc.set_dataset_from_ranges(meas["ranges"])
    ds = (dirs * meas["ranges"][:, None]).astype(np.float32)
    mask = ((meas["ranges"] >= np.float32(0.3)) & (meas["ranges"] <= np.float32(12.0))).astype(np.uint8)
    assert nv == int(mask.sum())
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 0.5
    s = rcc.computeCrossStatistics(T.identity())
    r64 = orc.statistics_p2l_f64(T.identity(), ds, mask, ref["points"], ref["normals"], ref["hits"], 0.5)
    assert int(s["n_meas"]) == r64["n_meas"] > 100
    assert np.allclose(s["covariance"].reshape(3, 3), r64["covariance"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("variant", [23, 0, 2, 24, 32])
def test_ondn_multi_origin(ra, orc, ctx, meshes, variant):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    W, H = 48, 10
    rng = np.random.RandomState(3)
    sm = T.spherical_model(np.float32(-0.3), np.float32(0.6 / (H - 1)), H, np.float32(-math.pi), np.float32(2 * math.pi / W), W,
                           np.float32(0.1), np.float32(25.0))
    dirs = syn.model_directions(sm).copy()
    origs = rng.uniform(-0.2, 0.2, size=dirs.shape).astype(np.float32)   # e.g. a multi-emitter sensor rig
    dirs[7] = np.nan
    Tsb = syn.tsb_offset()
    Tbm = T.transform_from_rpy((3.0, -2.5, 1.0), (0.0, 0.0, -2.2))
    rcc = ra.RCCHipOnDn(hm)
    rcc.set_traversal(variant)
    rcc.setTsb(Tsb)
    rcc.setModel(W, H, 0.1, 25.0, origs, dirs)
    rcc.find(Tbm)
    ref = m.simulate_ondn(W, H, 0.1, 25.0, origs, dirs, Tsb, Tbm, bvh=False)
    _compare(rcc.modelView(), ref, "ondn")
    # batch of poses through the same model
    poses = np.array([Tbm, T.mult(Tbm, syn.pose_c2_perturbation())], dtype=T.TRANSFORM)
    rcc.find_batch(poses)
    refb = m.simulate_ondn(W, H, 0.1, 25.0, origs, dirs, Tsb, poses, bvh=True)
    _compare(rcc.modelView(), refb, "ondn batch")
