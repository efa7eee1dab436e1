"""CPU pins for the pinhole / OnDn ray generators of the oracle (rmagine PinholeModel / OnDnModel semantics)."""
import numpy as np


def test_pinhole_direction_convention(orc):
    """optical axis -> +x (forward); image right -> -y; image down -> -z; unit length; centre pixel exact."""
    W, H, f, c = 64, 48, (50.0, 60.0), (31.5, 23.5)
    d = orc.pinhole_directions(W, H, f, c).reshape(H, W, 3)
    assert np.allclose(np.linalg.norm(d, axis=2), 1.0, atol=1e-6)
    assert np.all(d[..., 0] > 0)
    assert np.all(d[:, 40:, 1] < 0) and np.all(d[:, :24, 1] > 0)      # right half of the image looks to -y (right)
    assert np.all(d[30:, :, 2] < 0) and np.all(d[:20, :, 2] > 0)      # lower half looks down
    vid, hid = 10, 50
    opt = np.array([(hid - c[0]) / f[0], (vid - c[1]) / f[1], 1.0])
    opt /= np.linalg.norm(opt)
    assert np.allclose(d[vid, hid], [opt[2], -opt[0], -opt[1]], atol=1e-6)
    c2 = (32.0, 24.0)
    assert np.array_equal(orc.pinhole_directions(W, H, f, c2).reshape(H, W, 3)[24, 32], [1.0, -0.0, -0.0])


def test_pinhole_and_ondn_simulation_consistency(orc, meshes):
    """a pinhole model and the OnDn model built from its directions + zero origins give identical results;
    an OnDn origin offset shifts the ray start (point = dir * range + orig)."""
    from rmcl_amd import synthetic as syn, types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    W, H, fx, fy, cx, cy = 40, 30, 35.0, 35.0, 19.5, 14.5
    Tsb, Tbm = syn.tsb_offset(), syn.pose_c2_truth()
    a = m.simulate_pinhole(W, H, 0.1, 50.0, (fx, fy), (cx, cy), Tsb, Tbm, bvh=False)
    dirs = orc.pinhole_directions(W, H, (fx, fy), (cx, cy))
    b = m.simulate_ondn(W, H, 0.1, 50.0, np.zeros_like(dirs), dirs, Tsb, Tbm, bvh=True)
    for k in ("hits", "face_ids", "ranges", "normals"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.allclose(a["points"], b["points"], atol=1e-6)
    origs = np.tile(np.array([[0.1, -0.2, 0.05]], np.float32), (len(dirs), 1))
    c = m.simulate_ondn(W, H, 0.1, 50.0, origs, dirs, Tsb, Tbm, bvh=False)
    assert np.allclose(c["points"], dirs * c["ranges"][:, None] + origs, atol=1e-5)
    o1 = m.simulate_o1dn(W, H, 0.1, 50.0, origs[0], dirs, Tsb, Tbm, bvh=False)
    for k in ("hits", "face_ids", "ranges"):
        assert np.array_equal(c[k], o1[k]), k
