#!/usr/bin/env python
"""the particle filter's closest-point mode (correspondence_type 1, PCDSensorUpdaterEmbree.cpp:88-95): unseeded queries (rounds 1-3)
vs queries seeded from the map's FULL near grid (round 4); identical attributes (checked).  Figures quoted in DESIGN.md 4.5.
usage: python tools/pf_cpc_time.py [n_particles]"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = ra.Context(0)
for mesh in ("sphere", "room"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    bb = ((-5, -5, -1), (5, 5, 1)) if mesh == "sphere" else ((-9, -9, 0.3), (9, 9, 3))
    poses, attrs = syn.uniform_particles(n, seed=42, bb_min=bb[0] + (0, 0, -math.pi), bb_max=bb[1] + (0, 0, math.pi))
    beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
    ref = None
    for label, ct, flag in (("ray casting (correspondence_type 0)", 0, 0), ("closest point, near-grid seed (default)", 1, 0), ("closest point, unseeded (rounds 1-3)", 1, 256)):
        if n > 20000 and flag == 256 and mesh == "sphere":
            sub = 20000     # (the unseeded hollow-sphere case takes ~0.8 s per 100 000 particles)
        else:
            sub = n
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.config = T.pf_params(correspondence_type=ct)
        upd.init()
        upd.setInput(beams, T.identity())
        upd.set_mapping(flag, 0, None)
        d_p, d_a = ra.DeviceArray.from_host(ctx, poses[:sub]), ra.DeviceArray.from_host(ctx, attrs[:sub])
        t0 = time.perf_counter()
        upd.update(d_p, d_a)
        first_ms = (time.perf_counter() - t0) * 1e3
        out = d_a.download()
        same = ""
        if ct == 1:
            if ref is None:
                ref = out
            else:
                same = "== seeded" if out.tobytes() == ref[:sub].tobytes() else "DIFFERS"
        ms = min(upd.time_update(d_p, d_a, sub, iters=2) for _ in range(2))
        print("%-7s %-42s %7d particles %9.3f ms  %6.2f G evals/s  (first call %.1f ms) %s" % (mesh, label, sub, ms, sub * 256 / ms / 1e6, first_ms, same), flush=True)
        upd.close()
    hm.release()
