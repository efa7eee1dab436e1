#!/usr/bin/env python
"""pose batches (64 poses x 128x1024 rays; 1000 poses x 16x900) and the single C2 scan with the quantised-node kinds, by traversal
kind; kernel time by HIP events.  usage: python tools/batch_variants.py [kinds...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()
from rmcl_amd import synthetic as syn, types as T

kinds = [int(a) for a in sys.argv[1:]] or [22, 24, 19, 23, 4]
ctx = ra.Context(0)
for mesh in ("sphere", "room"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    base = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    rng = np.random.RandomState(0)
    p64 = np.array([T.mult(base, T.transform_from_rpy(tuple(rng.uniform(-0.5, 0.5, 3)), (0, 0, rng.uniform(-3, 3)))) for _ in range(64)], dtype=T.TRANSFORM)
    p1000 = np.array([T.mult(base, T.transform_from_rpy(tuple(rng.uniform(-1.0, 1.0, 3) * (1, 1, 0.3)), (0, 0, rng.uniform(-3, 3)))) for _ in range(1000)], dtype=T.TRANSFORM)
    for kind in kinds:
        row = []
        for model, poses in ((syn.model_c2(), p64), (syn.model_vlp16_900(), p1000), (syn.model_c2(), p64[:8])):
            rcc = ra.RCCHipSpherical(hm)
            rcc.setTsb(T.identity())
            rcc.setModel(model)
            rcc.set_traversal(kind)
            row.append(min(rcc.time_find_batch(poses, iters=5) for _ in range(3)))
            rcc.close()
        print("%s kind %2d: 64 x 128x1024 %8.4f ms | 1000 x 16x900 %8.4f ms | 8 x 128x1024 %8.4f ms" % (mesh, kind, row[0], row[1], row[2]), flush=True)
    hm.release()
