#!/usr/bin/env python
"""config C4's sensor update (100 000 particles x 256 beams), sphere-100k and room-100k, uniform and converged clouds: the four figures
the bench reports, alone (A/B of kernel edits).   usage: python tools/pf_c4_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

ctx = ra.Context(0)
for mesh, bb, centre in (("sphere100k", ((-5, -5, -1), (5, 5, 1)), T.transform_from_rpy((0.4, -0.3, 0.1), (0, 0, 0.4))),
                         ("room100k", ((-9, -9, 0.3), (9, 9, 3)), T.transform_from_rpy((1.5, -2.0, 1.6), (0, 0, 0.4)))):
    v, f = syn.uv_sphere(100000) if mesh.startswith("sphere") else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    ms, _ = bench._pf_c4(ra, syn, T, np, ctx, hm, 100000, 256, iters=3, bb=bb)
    mc, _ = bench._pf_c4(ra, syn, T, np, ctx, hm, 100000, 256, iters=3, converged_at=centre)
    print("%-10s uniform %.4f ms   converged %.4f ms" % (mesh, ms, mc), flush=True)
    hm.release()
