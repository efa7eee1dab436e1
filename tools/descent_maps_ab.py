#!/usr/bin/env python
"""find kind 32 (cooperative descent, several final caps) against kind 23 beyond the two benchmark maps: spheres of 1 M / 10 M faces (the same
pose, and 16 poses in turn: nothing of the previous launch's lines in the MALL helps), other scan sizes, the mixed-scale and sliver maps.
Kernel time: HIP events around back-to-back launches, median of 5 batches.
usage (GPU box): python tools/descent_maps_ab.py [--big]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T, _capi

ctx = ra.Context(0)
CAPS = (8, 12, 16, 64)


def model(H, W):
    m = syn.model_c2()
    m.phi.inc = m.phi.inc * 128.0 / H
    m.phi.size = H
    m.theta.inc = m.theta.inc * 1024.0 / W
    m.theta.size = W
    return m


def row(name, hm, m, poses, iters=30):
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(m)

    def t(kind, cap):
        rcc.set_traversal(kind)
        _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, cap, 24))
        ts = []
        for _ in range(5):
            if len(poses) == 1:
                ts.append(rcc.time_find(poses[0], iters))
            else:
                ts.append(float(np.mean([rcc.time_find(p, 1) for p in poses])))
        return sorted(ts)[2] * 1e3
    base = t(23, 64)
    print("%-44s kind 23 %8.2f us | kind 32: %s" % (name, base, "  ".join("cap %2d %8.2f (%+5.1f %%)" % (c, x, 100.0 * (x / base - 1.0)) for c, x in ((c, t(32, c)) for c in CAPS))), flush=True)
    rcc.close()


rng = np.random.RandomState(7)
rot = [syn.pose_c2_truth()] + [T.transform_from_rpy(tuple(rng.uniform(-3.0, 3.0, 3)), tuple(rng.uniform(-0.4, 0.4, 2)) + (rng.uniform(-3.1, 3.1),)) for _ in range(15)]
room_pose = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
maps = [("sphere-100k", lambda: syn.uv_sphere(100000), syn.pose_c2_truth()), ("room-100k", lambda: syn.noisy_room(100000), room_pose),
        ("room-30k", lambda: syn.noisy_room(30000), room_pose), ("cadmix-20k", lambda: syn.cad_mix(20000), T.transform_from_rpy((0.5, 0.3, 1.0), (0, 0, 0.3)))]
if "--big" in sys.argv:
    maps += [("sphere-1M", lambda: syn.uv_sphere(1000000), syn.pose_c2_truth()), ("sphere-10M", lambda: syn.uv_sphere(10000000), syn.pose_c2_truth())]
for name, gen, pose in maps:
    v, f = gen()
    hm = ra.import_hip_map(ctx, v, f)
    for H, W in ((128, 1024), (64, 1024), (128, 2048), (32, 2048), (64, 512)):
        row("%s %dx%d" % (name, H, W), hm, model(H, W), [pose])
    if name.startswith("sphere"):
        row("%s 128x1024, 16 poses in turn" % name, hm, model(128, 1024), rot)
    hm.release()
