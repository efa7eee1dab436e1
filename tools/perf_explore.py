#!/usr/bin/env python
"""GPU perf exploration of the find kernel: traversal variant x tile shape x batch size (rays in flight).
Prints one line per configuration; used to choose defaults (results quoted in DESIGN.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
print("map", hm.info())
model = syn.model_c2()
n = model.phi.size * model.theta.size
base = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
rng = np.random.RandomState(0)
poses = np.array([T.mult(base, T.transform_from_rpy(tuple(rng.uniform(-0.5, 0.5, 3)), (0, 0, rng.uniform(-3, 3))))
                  for _ in range(64)], dtype=T.TRANSFORM)
for kind in (1, 4):
    for tile in (0,):
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        rcc.set_variant(kind | (tile << 4))
        line = "kind=%d tile=%s" % (kind, "auto" if tile == 0 else "%dx%d" % (1 << (tile - 1), 64 >> (tile - 1)))
        for nposes in (1, 8, 64):
            ms = rcc.time_find(base, 30) if nposes == 1 else rcc.time_find_batch(poses[:nposes], 5)
            line += " | %2d poses: %8.4f ms %7.2f Grays/s" % (nposes, ms, nposes * n / ms / 1e6)
        print(line, flush=True)
        rcc.close()
