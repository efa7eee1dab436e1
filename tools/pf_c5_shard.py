#!/usr/bin/env python
"""config C5's per-GPU term alone -- 125 000 particles x 256 beams on the 1 M-triangle sphere, the fused sensor update -- for the PMC
passes of tools/pmc_traffic.sh (profiles/traffic.json: k_pf_update_v3_c5_shard_sphere1m) and for --valu / --l2 counter passes.
usage (GPU box): python tools/pf_c5_shard.py [launches=5]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = ra.Context(0)
v, f = syn.uv_sphere(1000000)
hm = ra.import_hip_map(ctx, v, f)
poses, attrs = syn.uniform_particles(125000, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
upd = ra.PCDSensorUpdaterHip(hm)
upd.init()
upd.setInput(beams, T.identity())
d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
ms = upd.time_update(d_p, d_a, 125000, iters=n_launch)
print("C5 shard (125000 x 256, sphere-1M): %.4f ms per launch" % ms)
