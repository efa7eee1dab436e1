"""Wire-format adapters.  CPU: the oracle's PointCloud2 unpack against a hand-computed case,
the field mappings of rmcl_amd.wire.  GPU: RCCHipO1Dn.setInputPointCloud2 == set_model_o1dn + set_dataset fed
with the oracle's unpack, bit for bit."""
import numpy as np
import pytest

REC = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])  # 22 B, unaligned
REC64 = np.dtype([("pad", "<u4"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8")])                                      # 28 B


def _cloud(rec, H, W, seed, truth_points=None):
    rng = np.random.RandomState(seed)
    a = np.zeros((H, W), rec)
    if truth_points is None:
        a["x"], a["y"], a["z"] = rng.uniform(-6, 6, (H, W)), rng.uniform(-6, 6, (H, W)), rng.uniform(-1, 2, (H, W))
    else:
        a["x"], a["y"], a["z"] = [truth_points[:, k].reshape(H, W) for k in range(3)]
    if "intensity" in rec.names:
        a["intensity"] = rng.uniform(0, 255, (H, W))
        a["ring"] = np.arange(H)[:, None]
    return a


def test_oracle_unpack_hand_case(orc):
    a = np.zeros((2, 3), REC)
    a["x"], a["y"], a["z"] = [[3, 0, np.nan], [0, 0, 1]], [[4, 0, 0], [0, 2, 2]], [[0, 0, 0], [5, 0, 2]]
    r = orc.pointcloud2_unpack(a.tobytes(), 3, 2, REC.itemsize, 3 * REC.itemsize, 0, 4, 8, 7, range_min=0.5, range_max=4.0)
    assert (r["width"], r["height"]) == (3, 2)
    assert np.array_equal(r["ranges"], np.float32([5, 0, 0, 5, 2, 3]))
    assert np.array_equal(r["mask"], [0, 0, 0, 0, 1, 1]) and r["n_valid"] == 2      # 5 > max, 0 < min, NaN -> 0
    assert np.allclose(r["dirs"][0], [0.6, 0.8, 0]) and np.array_equal(r["dirs"][2], [0, 0, 0])
    assert np.allclose(r["points"][5], [1, 2, 2], atol=1e-6)
    # sub-sampling like FilterOptions2D: rows 1.., every second column
    r2 = orc.pointcloud2_unpack(a.tobytes(), 3, 2, REC.itemsize, 3 * REC.itemsize, 0, 4, 8, 7, filter_h=(1, 0, 1), filter_w=(0, 0, 2))
    assert (r2["width"], r2["height"]) == (1, 1) and r2["ranges"][0] == 5
    with pytest.raises(ValueError):
        orc.pointcloud2_unpack(a.tobytes(), 3, 2, REC.itemsize, 3 * REC.itemsize, 0, 4, 8, 2)   # INT16 x field


def test_wire_field_mappings(ra):
    W = ra.wire
    m = W.spherical_from_scan_info(dict(phi_min=-0.2, phi_inc=0.1, phi_n=5, theta_min=-3.0, theta_inc=0.01, theta_n=600,
                                        range_min=0.3, range_max=80.0))
    assert (m.phi.size, m.theta.size) == (5, 600) and abs(m.range.max - 80.0) < 1e-6
    kw = W.pinhole_from_camera_info(640, 480, [525, 0, 319.5, 0, 520, 239.5, 0, 0, 1], 0.3, 8.0)
    assert (kw["fx"], kw["fy"], kw["cx"], kw["cy"]) == (525, 520, 319.5, 239.5)
    kw2 = W.pinhole_from_depth_info(dict(width=64, height=48, range_min=0.1, range_max=5, fx=50, fy=51, cx=32, cy=24))
    assert kw2["fy"] == 51 and kw2["width"] == 64
    for rec in (REC, REC64):
        a = _cloud(rec, 4, 9, 1)
        a["x"][2, 2] = np.nan     # only NaN is retried (x == x && ..., PCDSensorUpdaterEmbree.cpp:303); +-inf passes there too
        dt = W.FLOAT32 if rec is REC else W.FLOAT64
        xyz = W.xyz_from_pointcloud2(a.tobytes(), 36, rec.itemsize, rec.fields["x"][1], rec.fields["y"][1], rec.fields["z"][1], dt)
        assert np.array_equal(xyz[:, 0], a["x"].reshape(-1).astype(np.float32), equal_nan=True)
        beams = W.sample_beams_pointcloud2(a.tobytes(), 36, rec.itemsize, rec.fields["x"][1], rec.fields["y"][1], rec.fields["z"][1],
                                           samples=20, seed=3, datatype=dt)
        assert len(beams) == 20 and np.isfinite(beams["range"]).all()
        ref = ra.sample_beams(xyz, 20, seed=3)
        assert beams.tobytes() == ref.tobytes()
        # ... and the C-ABI sampler (librmclhip, std::mt19937) draws exactly what the oracle's own MT19937 restatement draws
        import oracle as orc
        oref = orc.sample_beams_pointcloud2(a.tobytes(), 36, 1, rec.itemsize, 36 * rec.itemsize, rec.fields["x"][1],
                                            rec.fields["y"][1], rec.fields["z"][1], 20, 3, datatype=dt)
        assert beams.tobytes() == oref.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("rec,fh,fw", [(REC, None, None), (REC64, (1, 2, 2), (3, 1, 3))])
def test_pointcloud2_input_equals_model_plus_dataset(ra, orc, ctx, meshes, rec, fh, fw):
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    Tsb = syn.tsb_offset()
    truth = T.transform_from_rpy((0.5, -0.3, 1.2), (0.02, -0.03, 0.4))
    est = T.mult(truth, syn.pose_c2_perturbation())
    pts = m.simulate_spherical(model, Tsb, truth, bvh=True)["points"]            # NaN where the scan missed
    H, Wd = int(model.phi.size), int(model.theta.size)
    a = _cloud(rec, H, Wd, 2, truth_points=pts)
    ox, oy, oz = (rec.fields[k][1] for k in "xyz")
    dt = 7 if rec is REC else 8
    u = orc.pointcloud2_unpack(a.tobytes(), Wd, H, rec.itemsize, rec.itemsize * Wd, ox, oy, oz, dt,
                               filter_h=fh or (0, 0, 1), filter_w=fw or (0, 0, 1), range_min=0.3, range_max=9.0)
    assert 0 < u["n_valid"] < u["width"] * u["height"]
    a_rcc = ra.RCCHipO1Dn(hm)
    a_rcc.setTsb(Tsb)
    got = a_rcc.setInputPointCloud2(a.tobytes(), Wd, H, rec.itemsize, rec.itemsize * Wd, ox, oy, oz, dt, 0.3, 9.0, fh, fw)
    assert got == (u["width"], u["height"], u["n_valid"])
    b_rcc = ra.RCCHipO1Dn(hm)
    b_rcc.setTsb(Tsb)
    b_rcc.setModel(u["width"], u["height"], 0.3, 9.0, (0, 0, 0), u["dirs"])
    b_rcc.set_dataset(u["points"], u["mask"])
    outs = []
    for rcc in (a_rcc, b_rcc):
        rcc.params.max_dist = 0.5
        rcc.find(est)
        mv = rcc.modelView()
        s = rcc.computeCrossStatistics(T.identity())
        outs.append((mv, s))
    for k in ("hits", "face_ids", "ranges", "points", "normals"):
        assert outs[0][0][k].tobytes() == outs[1][0][k].tobytes(), k
    assert outs[0][1].tobytes() == outs[1][1].tobytes() and int(outs[0][1]["n_meas"]) > 0
    # the cloud may already live on the device (e.g. written by a driver): same result
    d_raw = ra.DeviceArray.from_host(ctx, np.frombuffer(a.tobytes(), np.uint8))
    got2 = a_rcc.setInputPointCloud2(d_raw, Wd, H, rec.itemsize, rec.itemsize * Wd, ox, oy, oz, dt, 0.3, 9.0, fh, fw,
                                     device=True, nbytes=a.nbytes)
    assert got2 == got
    a_rcc.find(est)
    assert a_rcc.modelView()["face_ids"].tobytes() == outs[1][0]["face_ids"].tobytes()
    with pytest.raises(ra.RmclHipError):
        a_rcc.setInputPointCloud2(a.tobytes()[:100], Wd, H, rec.itemsize, rec.itemsize * Wd, ox, oy, oz, dt)


def test_beam_sampler_stream_and_edge_cases(ra):
    """the pinned random stream of the C-ABI beam sampler: MT19937 known answers (the 10000th output of the default-seeded
    engine is 4123659995, ISO C++ [rand.predef]), index = draw % n_points, retry on NaN only, early return when a sample
    stays invalid, organised clouds with row padding."""
    import oracle as orc
    assert orc.mt19937_draw(5489, 9999) == 4123659995
    assert orc.mt19937_draw(5489, 0) == 3499211612
    W = ra.wire
    # organised 3 x 5 cloud with 8 B of row padding and an intensity field before xyz
    rec = np.dtype([("i", "<f4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    rows = []
    rng = np.random.RandomState(0)
    a = np.zeros((3, 5), rec)
    for k in "xyz":
        a[k] = rng.uniform(-4, 4, (3, 5))
    a["x"][1, 2] = np.inf                       # +-inf is NOT retried by the reference (x == x): it comes through
    raw = b"".join(a[r].tobytes() + b"\0" * 8 for r in range(3))
    row_step = 5 * rec.itemsize + 8
    beams = ra.pf.sample_beams_pointcloud2(raw, 5, 3, rec.itemsize, row_step, 4, 8, 12, samples=200, seed=77)
    oref = orc.sample_beams_pointcloud2(raw, 5, 3, rec.itemsize, row_step, 4, 8, 12, 200, 77)
    assert len(beams) == 200 and beams.tobytes() == oref.tobytes()
    ids = [orc.mt19937_draw(77, k) % 15 for k in range(200)]
    pts = np.stack([a[k].reshape(-1) for k in "xyz"], 1)[ids]
    exp = ra.beams_from_points(pts)
    fin = np.isfinite(pts).all(1)
    assert np.array_equal(beams["range"][fin], exp["range"][fin]) and np.isinf(beams["range"][~fin]).all() and (~fin).any()
    # a cloud of NaNs: every retry fails, the sampler stops like the reference ("Point invalid", :306-311)
    nan_cloud = np.full((10, 3), np.nan, np.float32)
    assert len(ra.sample_beams(nan_cloud, 5, seed=1)) == 0
    # short buffers are an error, not a read past the end
    with pytest.raises(ra._capi.RmclHipError):
        ra.pf.sample_beams_pointcloud2(raw[:40], 5, 3, rec.itemsize, row_step, 4, 8, 12, samples=3, seed=1)
