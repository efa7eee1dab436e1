#!/usr/bin/env python
"""Which workgroup computes which tile of a single scan (FindParams::xcd_mapping; rmclhip_rcc_set_descent bits 29..30 force one, A/B):
as the hardware deals them (workgroup b on XCD b % 8: every XCD sees all of the image; the default since round 6), an eighth of the
image per XCD (rounds 1-5: neighbouring tiles share an L2), or the two workgroups of a CU from the two halves of the image.
Kernel time per kind, map and scan size; identical outputs.  (profiles/r06_tile_mapping_ab.txt also holds the first run, with a
fourth mapping -- tile rows round-robin over the XCDs -- that was never the best.)
usage (GPU box): python tools/band_pairing_ab.py [--big]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T, _capi

ctx = ra.Context(0)
NAMES = ("as dealt (default)", "an eighth of the image per XCD (rounds 1-5)", "a CU's two workgroups from the image's halves")
maps = [("sphere-100k", lambda: syn.uv_sphere(100000)), ("room-100k", lambda: syn.noisy_room(100000)), ("sphere-1M", lambda: syn.uv_sphere(1000000))]
if "--big" in sys.argv:
    maps.append(("sphere-10M", lambda: syn.uv_sphere(10000000)))
rng = np.random.RandomState(7)
rot = [T.transform_from_rpy(tuple(rng.uniform(-3.0, 3.0, 3)), tuple(rng.uniform(-0.4, 0.4, 2)) + (rng.uniform(-3.1, 3.1),)) for _ in range(16)]
for name, gen in maps:
    v, f = gen()
    hm = ra.import_hip_map(ctx, v, f)
    pose = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4)) if name.startswith("room") else syn.pose_c2_truth()
    for H, W in ((128, 1024), (64, 1024), (128, 2048), (32, 2048)):
        m = syn.model_c2()
        m.phi.inc = m.phi.inc * 128.0 / H; m.phi.size = H
        m.theta.inc = m.theta.inc * 1024.0 / W; m.theta.size = W
        rcc = ra.RCCHipSpherical(hm); rcc.setTsb(T.identity()); rcc.setModel(m)
        for kind in (23, 32, 24, 2):
            rcc.set_traversal(kind)
            row, ref = [], None
            for mode in (0, 1, 2):
                _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, 64, 24 | (24 << 8) | ((mode + 1) << 29)))
                ms = sorted(rcc.time_find(pose, 30) for _ in range(7))[3]
                extra = ""
                if (H, W) == (128, 1024) and name.startswith("sphere") and mode in (0, 1):
                    extra = " (16 poses in turn %.2f)" % (1e3 * float(np.median([np.mean([rcc.time_find(p, 1) for p in rot]) for _ in range(5)])))
                rcc.find(pose); mv = rcc.modelView()
                out = {k: np.array(mv[k]) for k in ("hits", "ranges", "face_ids")}
                ref = ref or out
                same = all(np.array_equal(ref[k], out[k], equal_nan=True) for k in ref)
                row.append("%s %.2f%s%s" % (NAMES[mode], ms * 1e3, extra, "" if same else " DIFFERENT"))
            print("%-12s %4dx%-4d kind %2d: %s us" % (name, H, W, kind, " | ".join(row)), flush=True)
        rcc.close()
    hm.release()
