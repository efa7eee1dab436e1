"""CPU tests of the oracle's closest-point query (CPCEmbree::find restated): analytic cases on the cube room,
BVH-culled search == brute force, tie-break, NaN handling."""
import numpy as np


def test_closest_point_analytic_cube(orc, meshes):
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    lo, hi = v.min(0), v.max(0)
    rng = np.random.RandomState(3)
    for _ in range(200):
        P = rng.uniform(lo + 0.05, hi - 0.05).astype(np.float32)
        d, cp, face = m.closest_point(P, bvh=False)
        gaps = np.concatenate([P - lo, hi - P])
        assert abs(d - gaps.min()) < 1e-5
        k = int(np.argmin(gaps))
        expect = P.copy()
        expect[k % 3] = lo[k % 3] if k < 3 else hi[k % 3]
        assert np.allclose(cp, expect, atol=1e-5)
        assert 0 <= face < len(f)


def test_closest_point_bvh_equals_brute(orc, meshes):
    v, f = meshes("room30k")
    m = orc.Mesh(v, f)
    rng = np.random.RandomState(4)
    for _ in range(200):
        P = rng.uniform(-11, 11, 3).astype(np.float32)
        a, b = m.closest_point(P, bvh=True), m.closest_point(P, bvh=False)
        assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1], b[1])


def test_cpc_find_semantics(orc, meshes):
    from rmcl_amd import types as T
    v, f = meshes("cube")
    m = orc.Mesh(v, f)
    I = T.identity()
    pts = np.array([[0, 0, -4.9], [4.0, 0, 2.0], [np.nan, 0, 0], v[0]], np.float32)
    r = m.cpc_find(I, I, pts, 0.5, bvh=False)
    assert list(r["hits"]) == [1, 0, 0, 1]            # 10 cm: hit; 1 m from the wall: beyond max_dist; NaN; on a vertex
    assert np.isnan(r["points"][2]).all() and r["face_ids"][2] == 0xFFFFFFFF
    assert r["ranges"][3] == 0.0
    # vertex shared by several faces: smallest face id wins
    shared = [i for i, tri in enumerate(f) if 0 in tri]
    assert r["face_ids"][3] == min(shared)
    # normals are the unflipped face normals rotated into the sensor frame
    Tsb = T.transform_from_rpy((0.1, 0.2, 0.3), (0.3, 0.2, 0.1))
    r2 = m.cpc_find(Tsb, I, pts[:1], 0.5, bvh=False)
    n_map = m.face_normals()[r2["face_ids"][0]]
    Tms = T.inv(Tsb)
    rot_only = T.transform([Tms["R"][k] for k in "xyzw"], (0, 0, 0))
    assert np.allclose(r2["normals"][0], orc.tapply(rot_only, n_map), atol=1e-6)
