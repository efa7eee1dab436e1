#!/usr/bin/env python
"""per-wave clocks of one find() of the C2 scan (clocked instantiation of the experiments library) as an .npz: tile id, whole-wave and
traversal cycles.  usage (GPU box): python tools/wave_dump.py [sphere|room] kind out.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()
from rmcl_amd import synthetic as syn, types as T

mesh, kind, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
pose = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
rcc = ra.RCCHipSpherical(hm)
rcc.setTsb(T.identity())
rcc.setModel(syn.model_c2())
rcc.set_traversal(kind)
rcc.time_find(pose, 20)
w = rcc.debug_wave_clock(pose)
w = w[w[:, 1] != 0].astype(np.int64)
np.savez(out, tile=w[:, 3] & 0xFFFFFF, whole=(w[:, 1] - w[:, 0]) & 0xFFFFFFFF, trav=(w[:, 5] - w[:, 4]) & 0xFFFFFFFF, maxvis=w[:, 7] & 63, tail_rays=(w[:, 7] >> 6) & 31, slow_steps=(w[:, 7] >> 11) & 511, tail_cycles=((w[:, 7] >> 20) & 4095) * 64)
print(mesh, kind, len(w), "waves; slowest", ((w[:, 1] - w[:, 0]) & 0xFFFFFFFF).max())
