#!/usr/bin/env python
"""closest-point correspondences on the C2 dataset (131 072 points, sphere-100k), 100 finds (for rocprofv3).  usage: cpc_trace.py [variant]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 15
tracking = (sys.argv[2] != "cold") if len(sys.argv) > 2 else True
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
truth = syn.pose_c2_truth()
est = T.mult(truth, syn.pose_c2_perturbation())
rcc = ra.RCCHipSpherical(hm)
rcc.setTsb(T.identity())
rcc.setModel(syn.model_c2())
rcc.find(truth)
mv = rcc.modelView()
cpc = ra.CPCHip(hm)
cpc.set_variant(variant)
cpc.set_tracking(tracking)
cpc.setTsb(T.identity())
cpc.params.max_dist = 1.0
cpc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
import time
import numpy as np
# a drifting estimate: the pose moves by a few millimetres / 0.02 degrees per call, like successive ICP corrections
steps = [T.mult(est, T.transform_from_rpy((0.002 * k, -0.001 * k, 0.0005 * k), (0.0, 0.0, 0.0003 * k))) for k in range(100)]
cpc.find(steps[0])
t0 = time.perf_counter()
for P in steps:
    cpc.find(P)
print("variant %d tracking %s: %.1f us per synchronous find (131 072 points, drifting pose)" % (variant, tracking, (time.perf_counter() - t0) / 100 * 1e6))
