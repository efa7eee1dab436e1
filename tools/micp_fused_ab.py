#!/usr/bin/env python
"""C3 schedule (R) with the moments formed in the find's epilogue (rmclhip_rcc_set_micp_fast 1, default) against the moments in a pass of
their own (mode 3, round 3's three-kernel form) and the per-iteration form (mode 0).  usage (GPU box): python tools/micp_fused_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

ctx = ra.Context(0)
for mesh in ("sphere", "room"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    truth = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    pert = syn.pose_c2_perturbation() if mesh == "sphere" else T.transform_from_rpy((0.02, -0.01, 0.01), (0.002, -0.003, 0.005))
    for rep in range(2):
        for mode, name in ((1, "moments in the find's epilogue"), (3, "moments in a pass of their own"), (0, "per-iteration form")):
            rcc = ra.RCCHipSpherical(hm)
            rcc.setTsb(T.identity())
            rcc.setModel(syn.model_c2())
            rcc.find(truth)
            rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
            rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
            rcc.set_micp_fast(mode)
            est = T.mult(truth, pert)
            rcc.correct_once(est, T.identity(), 10, 0.0, False)
            ms = sorted(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=50) for _ in range(5))[2]
            Tr, st = rcc.correct_once(est, T.identity(), 10, 0.0, False)
            info = rcc.micp_fast_info()
            print("%-6s %-34s %7.1f us | n_meas %d t %.7f %.7f %.7f | done %d/%d uncertain %d" %
                  (mesh, name, ms * 1e3, int(st["n_meas"]), Tr["t"]["x"], Tr["t"]["y"], Tr["t"]["z"], info["done"], info["attempts"],
                   info["last_uncertain"]), flush=True)
            rcc.close()
    hm.release()
