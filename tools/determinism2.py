#!/usr/bin/env python
"""find_async immediately followed by the synchronous statistics (schedule (B)'s iteration), N times: statistics compared bit
for bit; then the same with a stream sync in between.   usage: determinism2.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ctx = ra.Context(0)
v, f = syn.uv_sphere(20000)
hm = ra.import_hip_map(ctx, v, f)
model = syn.model_vlp16_900(0.0)
pose = T.transform((0, 0, 0, 1), (0.0, 0.0, 0.2))
pose2 = T.transform((0, 0, 0, 1), (0.0, 0.0, 0.15))
for kind in (15, 1):
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    if kind != 15:
        rcc.set_traversal(kind)
    rcc.find(T.identity())
    mv0 = rcc.modelView()
    rcc.set_dataset(mv0["points"].reshape(-1, 3), mv0["hits"].reshape(-1))
    rcc.params.max_dist = rcc.adaptive_max_dist_min = 1.0
    ref = {}
    for P, name in ((pose, "a"), (pose2, "b")):
        rcc.find(P)
        ref[name] = rcc.computeCrossStatistics(T.identity()).tobytes()
    for mode in ("back-to-back", "with sync"):
        bad = 0
        for i in range(reps):
            P, name = ((pose, "a"), (pose2, "b"))[i & 1]     # alternate poses so that a stale model buffer shows
            rcc.find_async(P)
            if mode == "with sync":
                rcc.sync()
            s = rcc.computeCrossStatistics(T.identity()).tobytes()
            if s != ref[name]:
                bad += 1
                if bad <= 3:
                    a = np.frombuffer(s, T.CROSS_STATISTICS)[0]; b = np.frombuffer(ref[name], T.CROSS_STATISTICS)[0]
                    print("   differs at", i, "n_meas", a["n_meas"], b["n_meas"], "dm", a["dataset_mean"], b["dataset_mean"], "== other pose:", s == ref["ab"[1 - (i & 1)]])
        print("kind %d %s: %d runs, %d differ" % (rcc.find_variant(1), mode, reps, bad), flush=True)
    rcc.close()
