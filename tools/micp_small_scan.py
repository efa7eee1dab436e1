#!/usr/bin/env python
"""C3-style correction (1 find + 10 iterations) for a 16x900 scan (VLP-16, the reference benchmark's sensor) and a 32x1024 one: host time
at the C ABI + the moment form's outcome.  usage (GPU box): python tools/micp_small_scan.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
truth = syn.pose_c2_truth()
est = T.mult(truth, syn.pose_c2_perturbation())
m32 = syn.model_c2()
m32.phi.inc = m32.phi.inc * 4.0
m32.phi.size = 32
for name, model in (("16x900", syn.model_vlp16_900()), ("32x1024", m32)):
    for mode in (1, 0):
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(model)
        rcc.find(truth)
        rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
        rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
        rcc.set_micp_fast(mode)
        for _ in range(3):
            rcc.correct_once(est, T.identity(), 10, 0.0, False)
        ms = sorted(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=50) for _ in range(5))[2]
        kms = rcc.time_find(est, 50)
        info = rcc.micp_fast_info()
        print("%-8s moment form %d: %6.1f us per correction | find kernel %5.2f us (kind %d) | done %d/%d" % (
            name, mode, ms * 1e3, kms * 1e3, rcc.find_variant(1), info["done"], info["attempts"]), flush=True)
        rcc.close()
