#!/usr/bin/env python
"""Find kind 32 (cooperative descent + one bit per final entry and ray instead of the sorted hand-over) against kinds 23 and 31: kernel
time and bit-identical outputs, per map.  Kind 32 runs once per leaves-per-ray bound (a wave one of whose rays enters more final leaves
starts at the root).  Needs librmclhip_lab.so for rmclhip_rcc_set_descent.
usage (GPU box): python tools/coop_leaves_ab.py [final caps...] [--leaf b1,b2,...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
try:
    ra.load_lab()
except Exception:
    pass
from rmcl_amd import synthetic as syn, types as T, _capi

args = sys.argv[1:]
leafs = [12]
if "--leaf" in args:
    i = args.index("--leaf")
    leafs = [int(x) for x in args[i + 1].split(",")]
    args = args[:i] + args[i + 2:]
caps = [int(a) for a in args] or [64]
ctx = ra.Context(0)
room_pose = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
cases = [("sphere-100k", syn.uv_sphere(100000), syn.pose_c2_truth()), ("room-100k", syn.noisy_room(100000), room_pose),
         ("sphere-1M", syn.uv_sphere(1000000), syn.pose_c2_truth())]
for name, (v, f), pose in cases:
    hm = ra.import_hip_map(ctx, v, f)
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_c2())
    ref = None
    line = []
    for kind, cap, leaf in [(23, 64, 12)] + [(31, c, 12) for c in caps] + [(32, c, l) for c in caps for l in leafs]:
        rcc.set_traversal(kind)
        _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, cap, 24 | (leaf << 8)))
        ms = sorted(rcc.time_find(pose, 30) for _ in range(7))[3]
        rcc.find(pose)
        mv = rcc.modelView()
        out = {k: np.array(mv[k]) for k in ("hits", "ranges", "points", "normals", "face_ids")}
        if ref is None:
            ref = out
            same = "reference"
        else:
            same = "identical" if all(np.array_equal(ref[k], out[k], equal_nan=True) for k in ref) else "DIFFERENT (%s)" % ", ".join(
                "%s: %d" % (k, int((~((ref[k] == out[k]) | ((ref[k] != ref[k]) & (out[k] != out[k])))).sum())) for k in ref)
        line.append("kind %d%s%s %.2f us %s" % (kind, "" if kind == 23 else " cap %d" % cap, " leaves <= %d" % leaf if kind == 32 else "", ms * 1e3, same))
    print("%-12s %s" % (name, "\n             ".join(line)))
    rcc.close()
    hm.release()
