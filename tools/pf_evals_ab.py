#!/usr/bin/env python
"""profiles/r04_pf_occupancy.txt: config C4's sensor update with a workgroup's beam errors in LDS (rounds 3: rmclhip_pf_set_mapping
bit 9) and in global scratch (round 4 default: 16 KB less LDS, 7 instead of 4 workgroups per CU); results must be identical.
usage: python tools/pf_evals_ab.py [lds|global]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

ctx = ra.Context(0)
beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
n = 100000
for mesh, bb, centre in (("sphere100k", ((-5, -5, -1), (5, 5, 1)), T.transform_from_rpy((0.4, -0.3, 0.1), (0, 0, 0.4))),
                         ("room100k", ((-9, -9, 0.3), (9, 9, 3)), T.transform_from_rpy((1.5, -2.0, 1.6), (0, 0, 0.4)))):
    v, f = syn.uv_sphere(100000) if mesh.startswith("sphere") else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    clouds = {"uniform": syn.uniform_particles(n, seed=42, bb_min=bb[0] + (0, 0, -math.pi), bb_max=bb[1] + (0, 0, math.pi)),
              "converged": syn.converged_particles(n, centre, 0.25, 5.0, seed=42)}
    for cname, (poses, attrs) in clouds.items():
        d_p = ra.DeviceArray.from_host(ctx, poses)
        ref = None
        forms = (("errors in LDS (round 3)", 1 << 9), ("errors in global scratch", 0))
        if len(sys.argv) > 1:      # `lds` / `global`: one form only (PMC passes: both forms are the same kernel name)
            forms = tuple(x for x in forms if (x[1] != 0) == (sys.argv[1] == "lds"))
        for label, bits in forms:
            upd = ra.PCDSensorUpdaterHip(hm)
            upd.init()
            upd.setInput(beams, T.identity())
            upd.set_mapping(bits, 0, None)
            d_a = ra.DeviceArray.from_host(ctx, attrs)
            upd.update(d_p, d_a)
            out = d_a.download()
            if ref is None:
                ref = out
            ms = sorted(upd.time_update(d_p, d_a, n, iters=3) for _ in range(5))[2]
            print("%-10s %-10s %-26s %7.4f ms  %6.2f G beam evaluations/s  %s" % (mesh, cname, label, ms, n * len(beams) / ms / 1e6,
                                                                                  "identical" if out.tobytes() == ref.tobytes() else "DIFFERS"), flush=True)
            upd.close()
    hm.release()
