#!/usr/bin/env python
"""A/B of the cooperative descent's knobs (find kind 31, rmclhip_rcc_set_descent) against kind 23 on sphere-100k and room-100k:
kernel time = HIP events around back-to-back launches (rmclhip_rcc_time_find), median of 7 batches of 30.
usage (GPU box): python tools/descent_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T, _capi

ctx = ra.Context(0)
for mesh in ("sphere", "room"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    base = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_c2())

    def t(kind, cap=64, lev=24):
        rcc.set_traversal(kind)
        _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, cap, lev))
        ts = sorted(rcc.time_find(base, 30) for _ in range(7))
        return ts[3] * 1e3
    print("%s-100k 128x1024: kind 23 %.2f us" % (mesh, t(23)), flush=True)
    for lev in (0, 1, 2, 3, 4, 24):
        print("  levels %2d: " % lev + "  ".join("cap %2d %6.2f" % (cap, t(31, cap, lev)) for cap in (8, 12, 16, 24, 32, 48, 64)), flush=True)
    rcc.close()
    hm.release()
