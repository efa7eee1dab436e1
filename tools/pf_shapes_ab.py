#!/usr/bin/env python
"""The two forms of the sensor update over cloud SHAPES (particles x beams): the order-independent accumulation keeps 384 B of accumulators per
particle of a workgroup in LDS, and a workgroup takes 2048 / n_beams particles -- few beams per particle mean many particles per workgroup.
   usage: python tools/pf_shapes_ab.py"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
dirs = syn.model_directions(syn.model_pf16())
for n, nb in ((100000, 16), (100000, 32), (100000, 64), (100000, 128), (100000, 256), (10000, 64), (10000, 256), (3000, 256), (1000, 256), (100000, 100), (20000, 1024)):
    poses, attrs = syn.uniform_particles(n, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    sel = np.linspace(0, len(dirs) - 1, nb).astype(int) if nb <= len(dirs) else np.arange(nb) % len(dirs)
    beams = ra.beams_from_points(dirs[sel] * np.float32(6.0))
    row = []
    for name, var in (("accumulate", 64), ("stored", 64 | 4096)):
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.set_variant(var)
        upd.setInput(beams, T.identity())
        d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
        upd.time_update(d_p, d_a, n, iters=1)
        ms = sorted(upd.time_update(d_p, d_a, n, iters=5) for _ in range(5))[2]
        row.append("%s %.4f ms" % (name, ms))
        upd.close()
    print("%7d x %4d: %s" % (n, nb, "   ".join(row)), flush=True)
