#!/usr/bin/env python
"""config C3's two figures alone (A/B of edits to the moment epilogue / the publish launch): a 10-iteration correction through
rmclhip_rcc_correct_once and the reference's unchanged caller loop through find + computeCrossStatistics, timed in C.
usage: python tools/c3_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

ctx = ra.Context(0)
for mesh in ("sphere100k", "room100k"):
    v, f = syn.uv_sphere(100000) if mesh.startswith("sphere") else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    truth = syn.pose_c2_truth() if mesh.startswith("sphere") else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_c2())
    rcc.find(truth)
    rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
    rcc.params.max_dist = 1.0
    rcc.adaptive_max_dist_min = 0.15
    est = T.mult(truth, syn.pose_c2_perturbation())
    rcc.correct_once(est, T.identity(), 10, 0.0, False)
    a = sorted(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=100) for _ in range(5))
    rcc.find(est)
    b = sorted(rcc.time_caller_loop(est, T.identity(), 10, 0.0, iters=100)[0] for _ in range(5))
    c = sorted(rcc.time_find_sync(est, iters=100) for _ in range(5))
    print("%-10s correct_once %.2f us (min %.2f)   unchanged caller loop %.2f us (min %.2f)   plain find %.2f us" %
          (mesh, a[2] * 1e3, a[0] * 1e3, b[2] * 1e3, b[0] * 1e3, c[2] * 1e3), flush=True)
    rcc.close()
    hm.release()
