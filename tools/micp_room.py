#!/usr/bin/env python
"""C3 schedule (R) on the occluded room-100k map, moment form vs one streaming launch per iteration, for a tracking-size and a
large correction (the large one leaves > 4096 correspondences undecided: the moment form reports it and holds off).
usage (GPU box): python tools/micp_room.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
ctx = ra.Context(0)
v, f = syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
model = syn.model_c2()
truth = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
for name, pert in (("2 cm / 0.3 deg", T.transform_from_rpy((0.02, -0.01, 0.005), (0.0, 0.0, 0.005))), ("C2 perturbation 0.2 m / 2 deg", syn.pose_c2_perturbation())):
    for mode in (1, 0):
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity()); rcc.setModel(model)
        rcc.find(truth)
        rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
        rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
        rcc.set_micp_fast(mode)
        est = T.mult(truth, pert)
        rcc.correct_once(est, T.identity(), 10, 0.0, False)
        ms = sorted(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=30) for _ in range(5))[2]
        i = rcc.micp_fast_info()
        print("room-100k C3(R), %-30s %s: %6.1f us  (attempts %d done %d cap exits %d overflows %d, undecided %d)" %
              (name, "moment form" if mode else "per-iteration", ms * 1e3, i["attempts"], i["done"], i["cap_exits"], i["overflows"], i["last_uncertain"]), flush=True)
        rcc.close()
