#!/usr/bin/env python
"""Pose batches in world order (rmclhip_rcc_set_batch_order, VERDICT r5 #7b) against the pose-major launch: the reference's v1 benchmark
shape (1000 poses x 16x900 VLP-16, lidar_corrector_embree_benchmark.cpp:117-135) on UV spheres and the room; time per batch INCLUDING
the keys and the sort, identical outputs.  `--pmc on|off`: run only that form a few times (for tools/pmc_sets.sh FETCH_SIZE / WRITE_SIZE).
usage (GPU box): python tools/batch_order_ab.py [faces...] [--room] [--pmc on|off]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T, _capi

args = sys.argv[1:]
pmc = None
if "--pmc" in args:
    i = args.index("--pmc")
    pmc = args[i + 1]
    args = args[:i] + args[i + 2:]
room = "--room" in args
sizes = [int(a) for a in args if a != "--room"] or [100000, 1000000]
ctx = ra.Context(0)
rng = np.random.RandomState(1)
poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-1.0, 1.0, 3) * (1, 1, 0.3)), (0, 0, rng.uniform(-3, 3))) for _ in range(1000)], dtype=T.TRANSFORM)
cases = [("sphere-%d" % nf, lambda nf=nf: syn.uv_sphere(nf), poses) for nf in sizes]
if room:
    rp = np.array([T.transform_from_rpy((rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.5, 2.5)), (0, 0, rng.uniform(-3, 3))) for _ in range(1000)], dtype=T.TRANSFORM)
    cases.append(("room-100000", lambda: syn.noisy_room(100000), rp))
for name, mesh, P in cases:
    v, f = mesh()
    hm = ra.import_hip_map(ctx, v, f)
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_vlp16_900())
    out = {}
    line = []
    for on in ((0, 1, 16, 256) if pmc is None else ((1,) if pmc == "on" else (0,))):
        _capi.check(_capi.lib().rmclhip_rcc_set_batch_order(rcc._h, on))
        rcc.find_batch(P)
        mv = rcc.modelView()
        out[on] = {k: np.array(mv[k]) for k in ("hits", "ranges", "face_ids", "normals")}
        ms = sorted(rcc.time_find_batch(P, iters=5) for _ in range(5 if pmc is None else 2))
        line.append("%s %.4f ms (kind %d)" % ("world order, %d workgroups per XCD turn" % (64 if on == 1 else on) if on else "pose-major", ms[len(ms) // 2], rcc.find_variant(len(P))))
    if pmc is None:
        same = all(np.array_equal(out[0][k], out[o][k], equal_nan=True) for k in out[0] for o in out)
        line.append("outputs identical" if same else "OUTPUTS DIFFER")
    print("%-16s 1000 x 16x900: %s" % (name, "\n                                ".join(line)), flush=True)
    rcc.close()
    hm.release()
