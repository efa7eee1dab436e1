#!/usr/bin/env python
"""closest-point correspondences: one lane per point vs four lanes per point, by dataset size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
truth = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
est = T.mult(truth, syn.pose_c2_perturbation())
for H, W in ((128, 1024), (64, 1024), (16, 900)):
    m = syn.model_c2()
    m.phi.inc, m.phi.size = m.phi.inc * 128.0 / H, H
    m.theta.inc, m.theta.size = m.theta.inc * 1024.0 / W, W
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(m)
    rcc.find(truth)
    mv = rcc.modelView()
    for variant in (1, 2):
        cpc = ra.CPCHip(hm)
        cpc.set_variant(variant)
        cpc.setTsb(T.identity())
        cpc.params.max_dist = 1.0
        cpc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
        cpc.find(est)
        t0 = time.perf_counter()
        for _ in range(20):
            cpc.find(est)
        dt = (time.perf_counter() - t0) / 20
        print("%s %4dx%-5d points %7d variant %d: %8.2f us  %7.1f M points/s" % (mesh, H, W, H * W, variant, dt * 1e6, H * W / dt / 1e6), flush=True)
        cpc.close()
    rcc.close()
