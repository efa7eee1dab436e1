#!/usr/bin/env python
"""schedule sweep of the round-3 particle-filter kernel (C4): refill threshold x tail lanes.  usage: pf_sweep.py [sphere|room]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
n = 100000
if mesh == "sphere":
    poses, attrs = syn.uniform_particles(n, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
else:
    poses, attrs = syn.uniform_particles(n, seed=42, bb_min=(-9, -9, 0.3, 0, 0, -math.pi), bb_max=(9, 9, 3, 0, 0, math.pi))
beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
d_poses, d_attrs = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
upd = ra.PCDSensorUpdaterHip(hm)
upd.init()
upd.setInput(beams, T.identity())
for refill in (40, 48, 52, 56, 60):
    row = []
    for tail in (2, 4, 8, 12, 16):
        upd.set_schedule(refill, tail)
        row.append(min(upd.time_update(d_poses, d_attrs, n, iters=4) for _ in range(3)))
    print(mesh, "refill>=%2d  tail<= 2/4/8/12/16: " % refill + "  ".join("%.3f" % x for x in row), flush=True)
