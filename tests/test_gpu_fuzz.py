"""GPU parity on adversarial meshes: random triangle soups (intersecting, sliver, zero-area triangles), exact duplicates
of faces under different ids (every hit is a tie -> the (min t, min face id) rule decides), coplanar overlapping
triangles and coordinates far from the origin.  Every traversal of find (wave packet, one lane per ray, quantised
nodes, four lanes per ray), the closest-point query and the particle filter must agree with the BRUTE-FORCE oracle:
no BVH on the checker's side, so builder, node twins and all traversal orders are covered end to end."""
import math

import numpy as np
import pytest

from conftest import assert_close_rel

pytestmark = pytest.mark.gpu


def _soup(seed, n_tri, offset=(0.0, 0.0, 0.0), scale=8.0):
    rng = np.random.RandomState(seed)
    c = rng.uniform(-scale, scale, (n_tri, 1, 3))
    tri = c + rng.normal(size=(n_tri, 3, 3)) * rng.uniform(0.05, 0.7, (n_tri, 1, 1))
    tri[::17, 2] = tri[::17, 1]                                     # zero-area triangles (two equal vertices)
    tri[::23] = tri[::23] * (1.0, 1.0, 0.0) + (0.0, 0.0, 0.5)       # a stack of coplanar, overlapping triangles in z = 0.5
    v = (tri.reshape(-1, 3) + np.asarray(offset)).astype(np.float32)
    f = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
    return v, f


def _duplicate_faces(v, f, seed):
    """every face twice, ids shuffled: all hits tie exactly"""
    rng = np.random.RandomState(seed)
    ff = np.concatenate([f, f])
    return v, ff[rng.permutation(len(ff))]


@pytest.mark.parametrize("kinds", [(0, 2, 23, 24, 32), pytest.param((1, 4, 5, 7, 9, 11, 12, 13, 14, 16, 17, 19, 20, 21, 22), marks=pytest.mark.lab)],
                         ids=["product", "experiments"])
@pytest.mark.parametrize("case", ["soup", "duplicates", "far_from_origin"])
def test_find_all_traversals_vs_brute_force(ra, orc, ctx, case, kinds):
    from rmcl_amd import synthetic as syn, types as T
    if case == "soup":
        v, f = _soup(1, 3000)
    elif case == "duplicates":
        v, f = _duplicate_faces(*_soup(2, 1500), seed=3)
    else:
        v, f = _soup(4, 2000, offset=(5000.0, -3000.0, 800.0))
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    model = syn.model_c1()
    origin = (0.3, -0.2, 0.1) if case != "far_from_origin" else (5000.3, -3000.2, 800.1)
    pose = T.transform_from_rpy(origin, (0.13, -0.27, 0.9))
    Tsb = syn.tsb_offset()
    ref = m.simulate_spherical(model, Tsb, pose, bvh=False)          # brute force over all triangles
    assert 100 < int(ref["hits"].sum()) < 1024, int(ref["hits"].sum())   # hits and misses
    for variant in kinds:
        rcc = ra.RCCHipSpherical(hm)
        rcc.set_traversal(variant)
        rcc.setTsb(Tsb)
        rcc.setModel(model)
        rcc.find(pose)
        g = rcc.modelView()
        what = "%s variant %d" % (case, variant)
        assert np.array_equal(g["hits"].reshape(-1), ref["hits"]), what
        assert np.array_equal(g["face_ids"].reshape(-1), ref["face_ids"]), what
        assert_close_rel(g["ranges"].reshape(-1), ref["ranges"], 1e-5, 0, what + " ranges")
        rcc.close()
    if case == "duplicates":                                        # ties are everywhere: the smaller id of each pair must win
        hit_ids = ref["face_ids"][ref["hits"] > 0]
        assert len(np.unique(hit_ids)) > 50


def test_cpc_and_pf_vs_brute_force_on_a_soup(ra, orc, ctx):
    from rmcl_amd import synthetic as syn, types as T
    v, f = _duplicate_faces(*_soup(7, 1200), seed=8)
    m = orc.Mesh(v, f)
    hm = ra.import_hip_map(ctx, v, f)
    rng = np.random.RandomState(9)
    pts = rng.uniform(-9, 9, (700, 3)).astype(np.float32)
    I = T.identity()
    for variant in (1, 2):
        cpc = ra.CPCHip(hm)
        cpc.set_variant(variant)
        cpc.setTsb(I)
        cpc.params.max_dist = 0.7
        cpc.set_dataset(pts, None)
        cpc.find(I)
        g = cpc.modelView()
        ref = m.cpc_find(I, I, pts, 0.7, bvh=False)
        assert np.array_equal(g["face_ids"].reshape(-1), ref["face_ids"]) and np.array_equal(g["hits"].reshape(-1), ref["hits"])
        assert_close_rel(g["ranges"].reshape(-1), ref["ranges"], 1e-5, 1e-7, "soup cpc distances")
        cpc.close()
    poses, attrs = syn.uniform_particles(300, seed=10, bb_min=(-7, -7, -7, -0.3, -0.3, -math.pi), bb_max=(7, 7, 7, 0.3, 0.3, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::5] * np.float32(2.5))
    a_ref = attrs.copy()
    e_ref = m.pf_update(poses, a_ref, beams, I, orc.pf_params(), bvh=False, want_errors=True)
    for variant in (64, 48, 64 | 512, 64 | 1024):   # the product's kernel and its knobs (experiments: tests/test_gpu_pf.py, `lab`)
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.set_variant(variant)
        upd.setInput(beams, I)
        d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
        d_e = ra.DeviceArray(ctx, np.float32, len(poses) * len(beams))
        upd.set_error_output(d_e)
        upd.update(d_p, d_a)
        assert_close_rel(d_e.download().reshape(e_ref.shape), e_ref, 1e-5, 1e-6, "soup pf errors variant %d" % variant)
        a = d_a.download()
        assert np.array_equal(a["likelihood"]["n_meas"], a_ref["likelihood"]["n_meas"])
        assert_close_rel(a["likelihood"]["mean"], a_ref["likelihood"]["mean"], 1e-5, 1e-12, "soup pf mean")
        upd.close()


def test_frontier_start_randomised_against_the_packet_traversal(ra, ctx):
    """The frontier start (kinds 23 / 24) culls subtrees for a whole wave at once: a box dropped wrongly loses hits silently.  160
    random scans -- soups, a room, a tiny and a far-away mesh; spherical models from 1 x 7 to 64 x 512 rays with fields of view from
    2 to 360 degrees; O1Dn models with random (partly NaN, partly repeated) directions; sensors inside, outside, far outside and ON
    the map's bounding box; random mounts -- must equal the wave-packet traversal (kind 0: no table, no culling) bit for bit."""
    from rmcl_amd import synthetic as syn, types as T
    rng = np.random.RandomState(77)
    maps = [("soup", _soup(21, 2500)), ("room", syn.noisy_room(20000)), ("tiny", syn.cube_room(side=0.2)),
            ("far", _soup(22, 1500, offset=(4000.0, 2500.0, -700.0), scale=30.0))]
    n_scans, n_hits, n_rays = 0, 0, 0
    for name, (v, f) in maps:
        hm = ra.import_hip_map(ctx, v, f)
        vv = np.asarray(v, np.float32).reshape(-1, 3)
        lo, hi = vv.min(0), vv.max(0)
        centre, ext = 0.5 * (lo + hi), (hi - lo)
        for case in range(40):
            where = case % 4                      # inside / just outside / far outside / on a face of the bounding box
            if where == 0:
                pos = centre + rng.uniform(-0.45, 0.45, 3) * ext
            elif where == 1:
                pos = centre + rng.choice([-1.0, 1.0], 3) * rng.uniform(0.55, 0.9, 3) * ext
            elif where == 2:
                pos = centre + rng.normal(size=3) * 40.0 * np.linalg.norm(ext)
            else:
                pos = centre + rng.uniform(-0.5, 0.5, 3) * ext
                ax = rng.randint(3)
                pos[ax] = lo[ax] if rng.rand() < 0.5 else hi[ax]
            pose = T.transform_from_rpy(tuple(float(x) for x in pos), tuple(float(x) for x in rng.uniform(-math.pi, math.pi, 3)))
            Tsb = T.transform_from_rpy(tuple(rng.uniform(-0.3, 0.3, 3)), tuple(rng.uniform(-0.5, 0.5, 3)))
            far = float(10.0 ** rng.uniform(-0.5, 4.0))
            if case % 3 == 2:
                W, H = int(rng.choice([1, 9, 64, 333])), int(rng.choice([1, 8, 17]))
                d = rng.normal(size=(W * H, 3)).astype(np.float32)
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                d[rng.rand(W * H) < 0.1] = np.nan
                d[rng.rand(W * H) < 0.1] = d[0]
                op = ra.RCCHipO1Dn(hm)
                op.setModel(W, H, 0.0, far, tuple(float(x) for x in rng.uniform(-0.2, 0.2, 3)), d)
            else:
                H, W = int(rng.choice([1, 3, 16, 64])), int(rng.choice([7, 64, 512]))
                fov_v, fov_h = float(rng.uniform(0.03, math.pi)), float(rng.uniform(0.03, 2 * math.pi))
                f32 = np.float32
                model = T.spherical_model(f32(-fov_v / 2), f32(fov_v / max(H - 1, 1)), H, f32(-fov_h / 2), f32(fov_h / W), W, f32(0.0), f32(far))
                op = ra.RCCHipSpherical(hm)
                op.setModel(model)
            op.setTsb(Tsb)
            out = {}
            for k in (0, 23, 24):
                op.set_traversal(k)
                op.find(pose)
                out[k] = op.modelView()
            for k in (23, 24):
                for key in ("hits", "ranges", "points", "normals", "face_ids"):
                    assert np.array_equal(out[k][key], out[0][key], equal_nan=True), (name, case, k, key)
            n_scans += 1
            n_hits += int(out[0]["hits"].sum())
            n_rays += out[0]["hits"].size
            op.close()
        hm.release()
    assert n_scans == 160 and 0.05 * n_rays < n_hits < 0.9 * n_rays, (n_hits, n_rays)     # hits and misses, plenty of both
