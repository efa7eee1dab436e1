#!/usr/bin/env python
"""How much of a single-scan find is dependent-load latency and how much is issue contention: the same scene with
fewer and fewer rays in flight (rows of the C2 model).  If the time does not drop with the ray count, the kernel is
bound by the slowest ray's chain of dependent node fetches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
base = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
for kind in (1, 5, 2):
    for H, W in ((128, 1024), (64, 1024), (64, 2048), (16, 900)):
        m = syn.model_c2()
        m.phi.inc = m.phi.inc * 128.0 / H
        m.phi.size = H
        m.theta.inc = m.theta.inc * 1024.0 / W
        m.theta.size = W
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(m)
        rcc.set_variant(kind)
        ms = rcc.time_find(base, 50)
        print("kind=%d %4dx%-5d rays %7d waves %5d : %7.2f us" % (kind, H, W, H * W, H * W // 64, ms * 1e3), flush=True)
        rcc.close()
