#!/usr/bin/env python
"""Generates the committed golden fixtures from the CPU oracle.

The reference ships no tests and cannot be built here (parity unpinned), so these vectors freeze the
oracle's restatement; the GPU path is compared against them on the GPU box, where neither the
reference nor this script's outputs can be regenerated from anything but the repo itself.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle as orc  # noqa: E402
import oracle_micp as om  # noqa: E402
from rmcl_amd import synthetic as syn  # noqa: E402  (input generators only)
from rmcl_amd import types as T  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def g2_cube():
    v, f = syn.cube_room()
    m = orc.Mesh(v, f)
    model = syn.model_c1()
    Tsb = syn.tsb_offset()
    poses = np.array([syn.pose_c2_truth(), T.transform_from_rpy((-2.1, 1.3, -0.7), (0.3, -0.2, 2.5)),
                      T.transform_from_rpy((3.9, -3.3, 2.2), (-0.1, 0.25, -1.2))], dtype=T.TRANSFORM)
    out = m.simulate_spherical(model, Tsb, poses, bvh=False)
    np.savez_compressed(os.path.join(HERE, "g2_cube_32x32.npz"), Tsb=np.array([Tsb]).view(np.uint8),
                        Tbm=poses.view(np.uint8), **out)
    return m, model, Tsb, poses, out


def g3_stats(m, model, Tsb, poses):
    truth = poses[0]
    est = orc.tmult(truth, syn.pose_c2_perturbation())
    meas = m.simulate_spherical(model, Tsb, truth, bvh=False)
    ds_pts, ds_mask = om.dataset_from_ranges(model, meas["ranges"])
    ds_mask[::17] = 0  # some invalid measurements
    sim = m.simulate_spherical(model, Tsb, est, bvh=False)
    Tpre = T.transform_from_rpy((0.01, -0.02, 0.015), (0.001, 0.002, -0.003))
    s32 = orc.statistics_p2l(Tpre, ds_pts, ds_mask, sim["points"], sim["normals"], sim["hits"], 0.8)
    s64 = orc.statistics_p2l_f64(Tpre, ds_pts, ds_mask, sim["points"], sim["normals"], sim["hits"], 0.8)
    Tu = orc.umeyama(s32)
    np.savez_compressed(os.path.join(HERE, "g3_stats_cube.npz"), truth=np.array([truth]).view(np.uint8),
                        est=np.array([est]).view(np.uint8), Tpre=np.array([Tpre]).view(np.uint8),
                        ds_mask=ds_mask, max_dist=np.float32(0.8), stats_f32=np.array([s32]).view(np.uint8),
                        f64_dataset_mean=s64["dataset_mean"], f64_model_mean=s64["model_mean"],
                        f64_covariance=s64["covariance"], f64_n=np.uint32(s64["n_meas"]),
                        umeyama=np.array([Tu]).view(np.uint8))


def g5_micp():
    v, f = syn.uv_sphere(20000)
    m = orc.Mesh(v, f)
    model = syn.model_vlp16_900(0.0)
    ident = orc.transform()
    meas = m.simulate_spherical(model, ident, ident, bvh=True, nthreads=8)
    ds_pts, ds_mask = om.dataset_from_ranges(model, meas["ranges"])
    Tom = T.transform((0, 0, 0, 1), (0.0, 0.0, 0.2))  # lidar_corrector_embree_benchmark.cpp:109-114
    res = {}
    for name, refind in (("R", False), ("B", True)):
        Tfin, merged, traj = om.correct_once(m, model, ident, ident, Tom, ds_pts, ds_mask, 10, 1.0, refind=refind, nthreads=8)
        res["traj_" + name] = np.array(traj, dtype=T.TRANSFORM).view(np.uint8)
        res["stats_" + name] = np.array([merged]).view(np.uint8)
    # same scene with a sensor offset and a base->odom transform
    Tsb, Tbo = syn.tsb_offset(), T.transform_from_rpy((0.3, -0.1, 0.0), (0.0, 0.0, 0.2))
    Tom2 = T.transform_from_rpy((0.1, -0.15, 0.2), (0.01, -0.02, 0.03))
    truth_bm = orc.tmult(T.identity(), Tbo)
    meas2 = m.simulate_spherical(model, Tsb, truth_bm, bvh=True, nthreads=8)
    ds2, mask2 = om.dataset_from_ranges(model, meas2["ranges"])
    Tfin, merged, traj = om.correct_once(m, model, Tsb, Tbo, Tom2, ds2, mask2, 10, 1.0, adaptive_min=0.15,
                                         convergence_progress=0.3, nthreads=8)
    res["traj_frames"] = np.array(traj, dtype=T.TRANSFORM).view(np.uint8)
    res["Tom2"] = np.array([Tom2]).view(np.uint8)
    res["Tbo"] = np.array([Tbo]).view(np.uint8)
    res["Tsb"] = np.array([Tsb]).view(np.uint8)
    np.savez_compressed(os.path.join(HERE, "g5_micp_sphere20k.npz"), **res)


def g6_pf():
    v, f = syn.cube_room()
    m = orc.Mesh(v, f)
    poses, attrs = syn.uniform_particles(64, seed=7, bb_min=(-4, -4, -2, 0, 0, -math.pi), bb_max=(4, 4, 2, 0, 0, math.pi))
    attrs["likelihood"]["n_meas"][::5] = 9995  # exercise the MAX_N_MEAS clamp
    attrs["likelihood"]["sigma"][::3] = 0.01
    dirs = syn.model_directions(syn.model_pf16())[::16][:16]
    from rmcl_amd.pf import beams_from_points
    beams = beams_from_points(dirs * np.linspace(0.02, 95.0, 16, dtype=np.float32)[:, None])  # some out of sensor range
    Tsb = syn.tsb_offset()
    params = orc.pf_params()
    a = attrs.copy()
    err = m.pf_update(poses, a, beams, Tsb, params, bvh=False, want_errors=True)
    np.savez_compressed(os.path.join(HERE, "g6_pf_cube.npz"), poses=poses.view(np.uint8), attrs_in=attrs.view(np.uint8),
                        beams=beams.view(np.uint8), Tsb=np.array([Tsb]).view(np.uint8), errors=err,
                        attrs_out=a.view(np.uint8))


def g7_digests():
    out = {}
    v, f = syn.uv_sphere(100000)
    m = orc.Mesh(v, f)
    model = syn.model_c2()
    r = m.simulate_spherical(model, T.identity(), syn.pose_c2_truth(), bvh=True, nthreads=8)
    # the BVH path is itself checked against brute force on a sample before its digest is frozen
    idx = np.random.RandomState(0).randint(0, len(r["face_ids"]), 1024)
    dirs = om.directions(model)
    Tbm = syn.pose_c2_truth()
    for i in idx:
        Dm = np.array(orc.tapply(orc.transform((Tbm["R"]["x"], Tbm["R"]["y"], Tbm["R"]["z"], Tbm["R"]["w"])), dirs[i]))
        O = (Tbm["t"]["x"], Tbm["t"]["y"], Tbm["t"]["z"])
        hit, t, face = m.intersect(O, Dm, 0.0, float(model.range.max), bvh=False)
        assert hit and face == r["face_ids"][i] and np.float32(t) == r["ranges"][i], i
    out["c2_sphere100k_face_ids_sha256"] = sha(r["face_ids"])
    out["c2_sphere100k_ranges_sha256"] = sha(r["ranges"])
    v, f = syn.noisy_room(100000)
    m = orc.Mesh(v, f)
    Tr = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    r = m.simulate_spherical(model, T.identity(), Tr, bvh=True, nthreads=8)
    out["c2_room100k_face_ids_sha256"] = sha(r["face_ids"])
    out["c2_room100k_hits_sha256"] = sha(r["hits"])
    out["c2_room100k_n_hits"] = int(r["hits"].sum())
    with open(os.path.join(HERE, "g7_digests.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    m, model, Tsb, poses, _ = g2_cube()
    g3_stats(m, model, Tsb, poses)
    g5_micp()
    g6_pf()
    g7_digests()
    print("golden fixtures written to", HERE)
