#!/usr/bin/env python
"""two sensors (128x1024 + 16x900) and six sensors through rmclhip_micp_correct_once: host time per correction at the C ABI (median of 25
calls after warm-up).  usage (GPU box): python tools/micp_multi_time.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import _capi as _c, synthetic as syn, types as T

ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
est = T.mult(syn.pose_c2_truth(), syn.pose_c2_perturbation())
vp = lambda a: a.ctypes.data_as(C.c_void_p)
for models in ((syn.model_c2(), syn.model_vlp16_900()), (syn.model_c2(), syn.model_c2()),
               (syn.model_c2(), syn.model_vlp16_900(), syn.model_vlp16_900(), syn.model_c2(), syn.model_vlp16_900(), syn.model_vlp16_900())):
    ops = []
    for mdl in models:
        rc = ra.RCCHipSpherical(hm)
        rc.setTsb(T.identity())
        rc.setModel(mdl)
        rc.find(syn.pose_c2_truth())
        rc.set_dataset_from_ranges(rc.modelView()["ranges"])
        rc.params.max_dist, rc.adaptive_max_dist_min = 1.0, 0.15
        rc._push_params()
        ops.append(rc)
    n = len(ops)
    hnd = (C.c_void_p * n)(*[o._h for o in ops])
    Tbo, w = np.array([T.identity()] * n, dtype=T.TRANSFORM), np.ones(n, np.float64)
    Tin, Tout, mrg = np.ascontiguousarray(est, dtype=T.TRANSFORM).reshape(1), np.zeros(1, T.TRANSFORM), np.zeros(1, T.CROSS_STATISTICS)
    call = lambda: _c.check(_c.lib().rmclhip_micp_correct_once(hnd, n, vp(Tin), vp(Tbo), vp(w), 10, 0.0, vp(Tout), vp(mrg)))
    for _ in range(10):
        call()
    ts = []
    for _ in range(25):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    print("%d sensors (%s): %.1f us per correction of 10 iterations; t = %.6f %.6f %.6f, n_meas %d" % (
        n, " + ".join("%dx%d" % (m.phi.size, m.theta.size) for m in models), sorted(ts)[12] * 1e6,
        Tout["t"]["x"][0], Tout["t"]["y"][0], Tout["t"]["z"][0], int(mrg["n_meas"][0])), flush=True)
    for o in ops:
        o.close()
