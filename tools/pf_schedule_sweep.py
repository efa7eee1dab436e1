#!/usr/bin/env python
"""profiles/r04_pf_schedule_sweep.txt: config C4's sensor update over the persistent-lane schedule (rmclhip_pf_set_schedule: refill when
>= R lanes of a wave are idle, 0 = the default 48; leave the node phase when <= T lanes still descend and a lane holds a leaf) at the
occupancy of round 4's kernel (7 workgroups per CU).   usage: python tools/pf_schedule_sweep.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
ctx = ra.Context(0)
beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
n = 100000
for mesh, bb in (("sphere100k", ((-5, -5, -1), (5, 5, 1))), ("room100k", ((-9, -9, 0.3), (9, 9, 3)))):
    v, f = syn.uv_sphere(100000) if mesh.startswith("sphere") else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    poses, attrs = syn.uniform_particles(n, seed=42, bb_min=bb[0] + (0, 0, -math.pi), bb_max=bb[1] + (0, 0, math.pi))
    d_p = ra.DeviceArray.from_host(ctx, poses); d_a = ra.DeviceArray.from_host(ctx, attrs)
    for refill in (0, 24, 32, 40, 48, 56):
        row = []
        for tail in (0, 4, 8, 12, 16, 24):
            upd = ra.PCDSensorUpdaterHip(hm); upd.init(); upd.setInput(beams, T.identity())
            upd.set_schedule(refill, tail)
            upd.time_update(d_p, d_a, n, iters=1)
            ms = sorted(upd.time_update(d_p, d_a, n, iters=3) for _ in range(3))[1]
            row.append("%.3f" % ms); upd.close()
        print(mesh, "refill %2d:" % refill, " ".join(row), " (tail 0 4 8 12 16 24)", flush=True)
