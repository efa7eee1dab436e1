#!/usr/bin/env python
"""profiles/r04_pf_converged_mapping.txt: the sensor update on a CONVERGED cloud (100 000 particles ~ N(pose, 0.25 m, 5 deg yaw) x 256
beams) against the uniform cloud of config C4, beam-minor (mapping 0) vs particle-minor dealing (mapping 1: a wave's lanes hold the
same beam of consecutive slots), with and without the Morton order of (x, y, yaw), blocks of 16 / 32 / 64 slots.  Every configuration
must leave the SAME attributes as mapping 0 (checked).   usage: python tools/pf_converged.py [n_particles]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = ra.Context(0)
beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
for mesh, centre in (("sphere100k", T.transform_from_rpy((0.4, -0.3, 0.1), (0, 0, 0.4))), ("room100k", T.transform_from_rpy((1.5, -2.0, 1.6), (0, 0, 0.4)))):
    v, f = syn.uv_sphere(100000) if mesh.startswith("sphere") else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    clouds = {"uniform": syn.uniform_particles(n, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi)),
              "converged": syn.converged_particles(n, centre, 0.25, 5.0, seed=42),
              "converged-wide": syn.converged_particles(n, centre, 1.0, 20.0, seed=43)}
    for cname, (poses, attrs) in clouds.items():
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.setInput(beams, T.identity())
        d_p = ra.DeviceArray.from_host(ctx, poses)
        order = syn.morton_order_xy_yaw(poses)
        d_order = ra.DeviceArray.from_host(ctx, order)
        ref = None
        for label, mapping, ppb, od in (("beam-minor (round 3)", 0, 0, None), ("particle-minor 16", 1, 16, None), ("particle-minor 32", 1, 32, None),
                                        ("particle-minor 64", 1, 64, None), ("particle-minor 16 + Morton", 1, 16, d_order),
                                        ("particle-minor 32 + Morton", 1, 32, d_order), ("particle-minor 64 + Morton", 1, 64, d_order)):
            upd.set_mapping(mapping, ppb, od)
            d_a = ra.DeviceArray.from_host(ctx, attrs)
            upd.update(d_p, d_a)
            out = d_a.download()
            if ref is None:
                ref = out
            same = out.tobytes() == ref.tobytes()
            ms = min(upd.time_update(d_p, d_a, n, iters=5) for _ in range(3))
            print("%-10s %-15s %-28s %7.3f ms  %6.2f G beam evaluations/s  %s" % (mesh, cname, label, ms, n * len(beams) / ms / 1e6,
                                                                                  "== mapping 0" if same else "DIFFERS"), flush=True)
        upd.close()
    hm.release()
