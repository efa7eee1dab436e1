#!/usr/bin/env python
"""Run-to-run determinism of find() and of the synchronous statistics: the same call N times, every output compared bit for
bit with the first run (a parity test that passes once says nothing about a 1-in-300 race).
usage: python tools/determinism.py [reps]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = ra.Context(0)


def digest(mv):
    h = hashlib.sha256()
    for k in ("hits", "ranges", "points", "normals", "face_ids"):
        h.update(np.ascontiguousarray(mv[k]).tobytes())
    return h.hexdigest()


cases = [("sphere20k vlp16 z+0.2", syn.uv_sphere(20000), syn.model_vlp16_900(0.0), T.transform((0, 0, 0, 1), (0.0, 0.0, 0.2)), (15, 2, 1)),
         ("sphere100k C2", syn.uv_sphere(100000), syn.model_c2(), syn.pose_c2_truth(), (15, 2, 19, 21, 22)),
         ("room100k C2", syn.noisy_room(100000), syn.model_c2(), T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4)), (15, 2))]
for name, (v, f), model, pose, kinds in cases:
    hm = ra.import_hip_map(ctx, v, f)
    for kind in kinds:
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        if kind == 15:
            rcc.set_variant(15)
        else:
            rcc.set_traversal(kind)
        rcc.find(pose)
        mv0 = rcc.modelView()
        d0 = digest(mv0)
        rcc.set_dataset(mv0["points"].reshape(-1, 3), mv0["hits"].reshape(-1))
        s0 = rcc.computeCrossStatistics(T.identity()).tobytes()
        bad_find = bad_stats = 0
        for i in range(reps):
            rcc.find(pose)
            mv = rcc.modelView()
            if digest(mv) != d0:
                bad_find += 1
                if bad_find <= 2:
                    diff = np.nonzero(mv["face_ids"] != mv0["face_ids"])[0]
                    dr = np.nonzero(mv["ranges"].view(np.uint32) != mv0["ranges"].view(np.uint32))[0]
                    print("   find differs: %d face ids, %d ranges; first ray %s" % (len(diff), len(dr), (dr[:4], mv["ranges"][dr[:4]], mv0["ranges"][dr[:4]])))
            if rcc.computeCrossStatistics(T.identity()).tobytes() != s0:
                bad_stats += 1
        print("%-24s kind %2d (launched %2d): %d runs, find differs %d x, statistics differ %d x" % (name, kind, rcc.find_variant(1), reps, bad_find, bad_stats), flush=True)
        rcc.close()
    hm.release()
