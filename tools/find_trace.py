#!/usr/bin/env python
"""one traversal kind, C2 scan, 300 back-to-back launches (for rocprofv3 --kernel-trace --stats).  usage: find_trace.py [sphere|room] kind"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()
from rmcl_amd import synthetic as syn, types as T

mesh, kind = sys.argv[1], int(sys.argv[2])
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
pose = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
rcc = ra.RCCHipSpherical(hm)
rcc.setTsb(T.identity())
rcc.setModel(syn.model_c2())
rcc.set_traversal(kind)
print("events: %.2f us" % (rcc.time_find(pose, 300) * 1e3))
