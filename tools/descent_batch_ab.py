#!/usr/bin/env python
"""pose batches: find kind 31 (cooperative descent, caps 12 / 32 / 64) against the rule's kind 24 and kind 23 -- the reference's v1 benchmark shape
(1000 poses x 16x900, lidar_corrector_optix_benchmark.cpp:86-133) and 64 x 128x1024, on spheres of 100 k / 1 M / 10 M faces.
Kernel time: HIP events around back-to-back launches (rmclhip_rcc_time_find_batch), median of 5.
usage (GPU box): python tools/descent_batch_ab.py [--big]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T, _capi

ctx = ra.Context(0)
rng = np.random.RandomState(0)
faces = [100000, 1000000] + ([10000000] if "--big" in sys.argv else [])
for nf in faces:
    v, f = syn.uv_sphere(nf)
    hm = ra.import_hip_map(ctx, v, f)
    for name, model, npos in (("1000 x 16x900", syn.model_vlp16_900(0.0), 1000), ("64 x 128x1024", syn.model_c2(), 64)):
        poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-3.0, 3.0, 3)), tuple(rng.uniform(-0.3, 0.3, 2)) + (rng.uniform(-3.1, 3.1),)) for _ in range(npos)], dtype=T.TRANSFORM)
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        row = []
        for kind, cap in ((24, 0), (23, 0), (31, 12), (31, 32), (31, 64)):
            rcc.set_traversal(kind)
            if cap:
                _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, cap, 24))
            ts = sorted(rcc.time_find_batch(poses, iters=3) for _ in range(5))
            row.append("kind %d%s %8.3f ms" % (kind, ("/%d" % cap) if cap else "", ts[2]))
        print("sphere %8d faces, %-14s: %s" % (nf, name, " | ".join(row)), flush=True)
        rcc.close()
    hm.release()
