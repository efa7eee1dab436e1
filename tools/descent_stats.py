#!/usr/bin/env python
"""What the cooperative descent of find kind 31 does per wave (clocked instantiation, librmclhip_lab.so): survivors of the frontier pass,
levels descended, final list length, inner nodes left unexpanded -- against the wave's cycles, and by tile row (elevation).
usage (GPU box): python tools/descent_stats.py [sphere|room] [final_cap] [levels]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()
from rmcl_amd import synthetic as syn, types as T, _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lev = int(sys.argv[3]) if len(sys.argv) > 3 else 24
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
hm = ra.import_hip_map(ctx, v, f)
pose = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
for kind in (23, 31):
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_c2())
    rcc.set_traversal(kind)
    _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, cap, lev))
    ms = rcc.time_find(pose, 50)
    w = rcc.debug_wave_clock(pose)
    w = w[w[:, 1] != 0].astype(np.int64)
    cyc = (w[:, 1] - w[:, 0]) & 0xFFFFFFFF
    tile = w[:, 3] & 0xFFFFFF
    row = tile // 64          # 1024 / 16 tiles per row of tiles
    print("== %s kind %d cap %d levels %d: kernel %.2f us; wave cycles mean %.0f p95 %.0f max %.0f" %
          (mesh, kind, cap, lev, ms * 1e3, cyc.mean(), np.percentile(cyc, 95), cyc.max()))
    print("   by tile row (32 rows of 4 scan lines, top = +22.5 deg): mean kcycles  " + " ".join("%2.0f" % (cyc[row == r].mean() / 1e3) for r in range(32)))
    print("                                                          max  kcycles  " + " ".join("%2.0f" % (cyc[row == r].max() / 1e3) for r in range(32)))
    st, lf, nv = (w[:, 2] & 0xFFFF) * 16, (w[:, 2] >> 16) & 255, (w[:, 2] >> 24) & 255
    trav = (w[:, 5] - w[:, 4]) & 0xFFFFFFFF
    print("   start (frontier cull / descent + per-ray filter) cycles mean %.0f p95 %.0f max %.0f | per-lane traversal mean %.0f p95 %.0f max %.0f" %
          (st.mean(), np.percentile(st, 95), st.max(), (trav - st).mean(), np.percentile(trav - st, 95), (trav - st).max()))
    print("   most leaf visits of a lane in the wave (before the quad tail): mean %.1f p95 %d max %d | most node visits: mean %.1f p95 %d max %d" %
          (lf.mean(), np.percentile(lf, 95), lf.max(), nv.mean(), np.percentile(nv, 95), nv.max()))
    if kind == 31:
        d = w[:, 7]
        ns, l1, l2, l3, nf = d & 63, (d >> 6) & 63, (d >> 12) & 63, (d >> 18) & 63, (d >> 24) & 127
        for name, x in (("frontier survivors", ns), ("entries after level 1", l1), ("after level 2", l2), ("after level 3", l3), ("final entries", nf)):
            print("   %-22s mean %5.1f p50 %3d p95 %3d max %3d   by row: %s" % (name, x.mean(), np.median(x), np.percentile(x, 95), x.max(),
                  " ".join("%2.0f" % x[row == r].mean() for r in range(32))))
        slow = cyc > np.percentile(cyc, 95)
        print("   the slowest 5 %% of the waves: survivors %.1f after level 1 %.1f level 2 %.1f level 3 %.1f final %.1f" %
              (ns[slow].mean(), l1[slow].mean(), l2[slow].mean(), l3[slow].mean(), nf[slow].mean()))
        fast = cyc < np.percentile(cyc, 50)
        print("   the faster half               : survivors %.1f after level 1 %.1f level 2 %.1f level 3 %.1f final %.1f" %
              (ns[fast].mean(), l1[fast].mean(), l2[fast].mean(), l3[fast].mean(), nf[fast].mean()))
    rcc.close()
