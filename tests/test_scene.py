"""Scenes of several (instanced) meshes: rmclhip_map_create_scene / rmclhip_scene_flatten_host.

The reference hands whole assimp scenes to rm::import_embree_map / import_optix_map (micp_localization.cpp:187-195).  The
flattening is restated here in numpy, operation by operation, and compared bit for bit; on the GPU a scene map must answer
exactly like a map made from the flattened soup."""
import numpy as np
import pytest

import rmcl_amd as ra
from rmcl_amd import synthetic as syn


def _affine(rng, scale=True):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    if scale:
        R = R @ np.diag(rng.uniform(0.5, 2.0, 3))
    A = np.zeros((3, 4), np.float32)
    A[:, :3] = R
    A[:, 3] = rng.uniform(-3, 3, 3)
    return A


def _flatten_numpy(meshes, instances):
    vs, fs, first = [], [], [0]
    vo = 0
    for m, A in instances:
        v, f = meshes[m]
        v = np.asarray(v, np.float32).reshape(-1, 3)
        if A is None:
            w = v.copy()
        else:
            A = np.asarray(A, np.float32)
            w = np.empty_like(v)
            for r in range(3):
                acc = A[r, 0] * v[:, 0]
                acc = acc + A[r, 1] * v[:, 1]
                acc = acc + A[r, 2] * v[:, 2]
                w[:, r] = acc + A[r, 3]
        vs.append(w)
        fs.append(np.asarray(f, np.uint32).reshape(-1, 3) + np.uint32(vo))
        vo += len(v)
        first.append(first[-1] + len(fs[-1]))
    return np.concatenate(vs), np.concatenate(fs), np.array(first, np.uint32)


def _scene(seed=3):
    rng = np.random.RandomState(seed)
    meshes = [syn.uv_sphere(300, radius=0.8), syn.cube_room(), syn.uv_sphere(60, radius=0.5)]
    instances = [(0, _affine(rng)), (1, np.eye(4, dtype=np.float32)), (2, _affine(rng)), (0, _affine(rng)), (2, _affine(rng, False))]
    return meshes, instances


def test_flatten_matches_numpy_bit_for_bit():
    meshes, instances = _scene()
    v, f, first = ra.flatten_scene_host(meshes, instances)
    ev, ef, efirst = _flatten_numpy(meshes, instances)
    assert v.shape == ev.shape and f.shape == ef.shape
    assert np.array_equal(v.view(np.uint32), ev.view(np.uint32))
    assert np.array_equal(f, ef) and np.array_equal(first, efirst)


def test_flatten_without_instances_is_concatenation():
    meshes, _ = _scene()
    v, f, first = ra.flatten_scene_host(meshes, None)
    ev, ef, efirst = _flatten_numpy(meshes, [(k, None) for k in range(len(meshes))])
    assert np.array_equal(v.view(np.uint32), ev.view(np.uint32)) and np.array_equal(f, ef) and np.array_equal(first, efirst)


def test_flatten_single_mesh_identity_is_the_mesh():
    v0, f0 = syn.uv_sphere(100)
    v, f, first = ra.flatten_scene_host([(v0, f0)], [(0, np.eye(3, 4, dtype=np.float32))])
    assert np.array_equal(v, np.asarray(v0, np.float32).reshape(-1, 3)) and np.array_equal(f, np.asarray(f0, np.uint32).reshape(-1, 3))
    assert list(first) == [0, len(f)]


def test_flatten_rejects_bad_input():
    v0, f0 = syn.uv_sphere(100)
    with pytest.raises(RuntimeError, match="mesh index 1 of 1"):
        ra.flatten_scene_host([(v0, f0)], [(1, np.eye(3, 4))])
    bad = np.array(f0, np.uint32).reshape(-1, 3).copy()
    bad[5, 1] = len(np.asarray(v0).reshape(-1, 3))
    with pytest.raises(RuntimeError, match="face 5 references vertex"):
        ra.flatten_scene_host([(v0, bad)], None)
    with pytest.raises(ValueError, match="3x4 or 4x4"):
        ra.flatten_scene_host([(v0, f0)], [(0, np.eye(3))])
    with pytest.raises(RuntimeError, match="no meshes"):
        ra.flatten_scene_host([], None)


def test_flatten_empty_instance_list_and_empty_mesh():
    v0, f0 = syn.uv_sphere(100)
    v, f, first = ra.flatten_scene_host([(v0, f0)], [])
    assert v.shape == (0, 3) and f.shape == (0, 3) and list(first) == [0]
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    v, f, first = ra.flatten_scene_host([empty, (v0, f0)], None)
    assert list(first) == [0, 0, len(f)]


@pytest.mark.gpu
def test_scene_map_answers_like_the_flattened_map(ra, orc, ctx):
    from rmcl_amd import types as T
    meshes, instances = _scene()
    v, f, first = ra.flatten_scene_host(meshes, instances)
    scene_map, flat_map = ra.import_hip_scene(ctx, meshes, instances), ra.import_hip_map(ctx, v, f)
    assert np.array_equal(scene_map.scene_instances(), first)
    assert list(flat_map.scene_instances()) == [0, len(f)]
    assert scene_map.info()["n_faces"] == len(f)
    model = syn.model_c2()
    Tbm = T.transform_from_rpy((0.4, -0.3, 0.2), (0.05, -0.1, 0.7))
    outs = []
    for mp in (scene_map, flat_map):
        rcc = ra.RCCHipSpherical(mp)
        rcc.setTsb(syn.tsb_offset())
        rcc.setModel(model)
        rcc.find(Tbm)
        outs.append(rcc.modelView())
        rcc.close()
    for k in ("hits", "ranges", "points", "normals", "face_ids"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    # and like the oracle on the flattened soup (hits / face ids bit-exact)
    ref = orc.Mesh(v, f).simulate_spherical(model, syn.tsb_offset(), Tbm)
    assert np.array_equal(outs[0]["hits"], ref["hits"]) and np.array_equal(outs[0]["face_ids"], ref["face_ids"])
    hit = outs[0]["hits"] > 0
    assert hit.all()               # the room encloses the sensor
    seen = set()
    for fid in outs[0]["face_ids"][hit][::37]:
        inst, local = scene_map.scene_locate(fid)
        assert first[inst] <= fid < first[inst + 1] and local == fid - first[inst]
        assert local < len(np.asarray(meshes[instances[inst][0]][1]).reshape(-1, 3))
        seen.add(inst)
    assert len(seen) >= 3          # the scan sees several instances
    with pytest.raises(RuntimeError, match="out of range"):
        scene_map.scene_locate(len(f))
