#!/usr/bin/env python
"""Launch timeline of one find(): entry, traversal begin / end, stores issued, exit of every wave (s_memtime) -- is the launch
bound by a few slow waves, by what happens before / after the traversal, or by all of them?
usage: python tools/wave_timeline.py [sphere|room] [kinds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
kinds = [int(a) for a in sys.argv[2:]] or [1, 5, 2]
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
pose = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
q = lambda x: "mean %6.0f p50 %6.0f p95 %6.0f max %6.0f" % (x.mean(), np.median(x), np.percentile(x, 95), x.max())
for kind in kinds:
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_c2())
    rcc.set_variant((kind & 15) | ((kind >> 4) << 13))
    ms = rcc.time_find(pose, 50)
    w = rcc.debug_wave_clock(pose)
    w = w[w[:, 1] != 0].astype(np.int64)
    d = lambda a, b: (w[:, a] - w[:, b]) & 0xFFFFFFFF
    rt = w[:, 2] - w[:, 2].min()
    print("== %s kind %d: kernel %.2f us (events, back-to-back launches); %d waves" % (mesh, kind, ms * 1e3, len(w)))
    print("   whole wave        cycles: %s" % q(d(1, 0)))
    print("   entry -> traversal       : %s" % q(d(4, 0)))
    print("   traversal                : %s" % q(d(5, 4)))
    print("   traversal -> stores issued %s" % q(d(6, 5)))
    print("   stores issued -> completed %s" % q(d(1, 6)))
    print("   wave entry, us after the first wave (100 MHz realtime clock): p50 %.2f p95 %.2f max %.2f" %
          (np.median(rt) / 100.0, np.percentile(rt, 95) / 100.0, rt.max() / 100.0))
    rcc.close()
