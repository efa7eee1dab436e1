#!/usr/bin/env python
"""two sensors (128x1024 + 16x900) through rmclhip_micp_correct_once, 200 corrections back to back (for rocprofv3 --kernel-trace)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import _capi as _c, synthetic as syn, types as T

ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
ops = []
for mdl in (syn.model_c2(), syn.model_vlp16_900()):
    rc = ra.RCCHipSpherical(hm)
    rc.setTsb(T.identity())
    rc.setModel(mdl)
    rc.find(syn.pose_c2_truth())
    rc.set_dataset_from_ranges(rc.modelView()["ranges"])
    rc.params.max_dist, rc.adaptive_max_dist_min = 1.0, 0.15
    rc._push_params()
    ops.append(rc)
est = T.mult(syn.pose_c2_truth(), syn.pose_c2_perturbation())
hnd = (C.c_void_p * 2)(ops[0]._h, ops[1]._h)
Tbo2, w2 = np.array([T.identity(), T.identity()], dtype=T.TRANSFORM), np.ones(2, np.float64)
Tin, Tout, mrg = np.ascontiguousarray(est, dtype=T.TRANSFORM).reshape(1), np.zeros(1, T.TRANSFORM), np.zeros(1, T.CROSS_STATISTICS)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
for _ in range(200):
    _c.check(_c.lib().rmclhip_micp_correct_once(hnd, 2, vp(Tin), vp(Tbo2), vp(w2), 10, 0.0, vp(Tout), vp(mrg)))
print(Tout)
