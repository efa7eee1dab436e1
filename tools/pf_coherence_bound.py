#!/usr/bin/env python
"""profiles/r04_pf_coherence_bound.txt: what the sensor update can gain AT MOST from coherent waves.  64 beams per particle, so that a
block of 64 slots holds what config C4's block of 16 slots x 256 beams holds (same LDS, same occupancy); clouds from uniform down to
sigma = 0 (every particle the same pose: with particle-minor dealing all 64 lanes of a wave then walk the SAME ray -- no divergence of
any kind, one cache line per fetch).   usage: python tools/pf_coherence_bound.py [n_particles]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
ctx = ra.Context(0)
dirs = syn.model_directions(syn.model_pf16()).reshape(16, 16, 3)[::2, ::2].reshape(-1, 3)
beams = ra.beams_from_points(np.ascontiguousarray(dirs) * np.float32(6.0))
for mesh, centre in (("sphere100k", T.transform_from_rpy((0.4, -0.3, 0.1), (0, 0, 0.4))), ("room100k", T.transform_from_rpy((1.5, -2.0, 1.6), (0, 0, 0.4)))):
    v, f = syn.uv_sphere(100000) if mesh.startswith("sphere") else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    clouds = {"uniform": syn.uniform_particles(n, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi)),
              "0.25 m / 5 deg": syn.converged_particles(n, centre, 0.25, 5.0, seed=42),
              "0.05 m / 1 deg": syn.converged_particles(n, centre, 0.05, 1.0, seed=42),
              "sigma = 0": syn.converged_particles(n, centre, 0.0, 0.0, seed=42)}
    for cname, (poses, attrs) in clouds.items():
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.setInput(beams, T.identity())
        d_p = ra.DeviceArray.from_host(ctx, poses)
        d_order = ra.DeviceArray.from_host(ctx, syn.morton_order_xy_yaw(poses))
        ref = None
        for label, mapping, ppb, od in (("beam-minor, 64 slots", 0, 64, None), ("particle-minor 64", 1, 64, None), ("particle-minor 64 + Morton", 1, 64, d_order)):
            upd.set_mapping(mapping, ppb, od)
            d_a = ra.DeviceArray.from_host(ctx, attrs)
            upd.update(d_p, d_a)
            out = d_a.download()
            if ref is None:
                ref = out
            same = out.tobytes() == ref.tobytes()
            ms = min(upd.time_update(d_p, d_a, n, iters=5) for _ in range(3))
            print("%-10s %-15s %-28s %7.3f ms  %6.2f G beam evaluations/s  %s" % (mesh, cname, label, ms, n * len(beams) / ms / 1e6,
                                                                                  "== first row" if same else "DIFFERS"), flush=True)
        upd.close()
    hm.release()
