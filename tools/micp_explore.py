#!/usr/bin/env python
"""C3 (schedule R: 1 find + 10 x (reduce + solve)) with every form of the MICP loop; host clock inside the library
(rmclhip_rcc_time_correct_once).  usage (GPU box): python tools/micp_explore.py [n_iter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
model = syn.model_c2()
forms = [("moment form (default, automatic)", -1), ("moment form replayed from a hipGraph (round 2)", -2), ("one launch per iteration", 0), ("reduce + solve launches", 1 << 10), ("persistent, 32 blocks", 3 << 10),
         ("persistent, 64 blocks", 4 << 10), ("persistent, 32 blocks on ONE XCD", 7 << 10), ("default without hipGraph", 1 << 9)]
for name, bits in forms:
    rcc = ra.RCCHipSpherical(hm)
    rcc.set_variant(15 | max(bits, 0))
    rcc.set_micp_fast({-1: 1, -2: 2}.get(bits, 0))
    rcc.setTsb(T.identity())
    rcc.setModel(model)
    rcc.find(syn.pose_c2_truth())
    rcc.set_dataset_from_ranges(rcc.modelView()["ranges"])
    rcc.params.max_dist, rcc.adaptive_max_dist_min = 1.0, 0.15
    est = T.mult(syn.pose_c2_truth(), syn.pose_c2_perturbation())
    ms = sorted(rcc.time_correct_once(est, T.identity(), n_iter, 0.0, False, iters=30) for _ in range(5))[2]
    ms0 = sorted(rcc.time_correct_once(est, T.identity(), 0, 0.0, False, iters=30) for _ in range(5))[2] if bits in (-1, -2, 0, 1 << 9) else float("nan")
    Tr, st = rcc.correct_once(est, T.identity(), n_iter, 0.0, False)
    print("%-40s %7.1f us per correction (%d iterations)  | 0 iterations %6.1f us | n_meas %d t %.6f %.6f %.6f" %
          (name, ms * 1e3, n_iter, ms0 * 1e3, int(st["n_meas"]), Tr["t"]["x"], Tr["t"]["y"], Tr["t"]["z"]), flush=True)
    if bits < 0:
        print("    moment form:", rcc.micp_fast_info(), flush=True)
    rcc.close()
