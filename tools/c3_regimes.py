#!/usr/bin/env python
"""profiles/r04_c3_regimes.txt: a 10-iteration MICP correction of a 128x1024 scan (rmclhip_rcc_correct_once, and the reference's
unchanged caller loop) over the size of the initial error, on the sphere and on the occluded room: how many correspondences are
undecided for the learnt caps, which form served the call (host iterations / device loop / per-iteration), what it cost.
usage: python tools/c3_regimes.py"""
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
    for scale in (0.02, 0.05, 0.1, 0.25, 0.5, 1.0):
        rcc = ra.RCCHipSpherical(hm)
        rcc.setTsb(T.identity())
        rcc.setModel(syn.model_c2())
        rcc.find(truth)
        rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
        rcc.params.max_dist = 1.0
        rcc.adaptive_max_dist_min = 0.15
        est = T.mult(truth, T.transform_from_rpy((0.2 * scale, 0.0, 0.0), (0.0, 0.0, 0.0349 * scale)))   # scale 1 = (0.2 m, 2 deg)
        for _ in range(3):
            rcc.correct_once(est, T.identity(), 10, 0.0, False)
        i0 = rcc.micp_fast_info()
        a = sorted(rcc.time_correct_once(est, T.identity(), 10, 0.0, False, iters=50) for _ in range(3))[1]
        i1 = rcc.micp_fast_info()
        rcc.find(est)
        b = sorted(rcc.time_caller_loop(est, T.identity(), 10, 0.0, iters=50)[0] for _ in range(3))[1]
        c = rcc.ccs_info()
        print("%-10s error %.3f m / %.2f deg: undecided %5d  correct_once %7.2f us (host loops %d of %d)   caller loop %7.2f us (from moments %d of %d calls)" %
              (mesh, 0.2 * scale, 2.0 * scale, i1["last_uncertain"], a * 1e3, i1["host_loops"] - i0["host_loops"], 150, b * 1e3, c["from_moments"], c["calls"]), flush=True)
        rcc.close()
    hm.release()
