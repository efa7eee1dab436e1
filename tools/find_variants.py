#!/usr/bin/env python
"""A/B table of the find traversals (set_variant kinds) on sphere-100k and room-100k at several scan sizes; kernel time =
HIP events around back-to-back launches on the handle's stream (rmclhip_rcc_time_find), median of 7 batches of 30.
usage (GPU box): python tools/find_variants.py [kinds...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

kinds = [int(a) for a in sys.argv[1:]] or [19, 23, 22, 24, 2]
ctx = ra.Context(0)
sizes = ((128, 1024), (64, 1024), (128, 2048), (256, 2048), (16, 900))
for mesh in ("sphere", "room"):
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    base = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    print("%s-100k   %s" % (mesh, "  ".join("%4dx%-4d" % s for s in sizes)), flush=True)
    for kind in kinds:
        row = []
        for H, W in sizes:
            m = syn.model_c2()
            m.phi.inc = m.phi.inc * 128.0 / H
            m.phi.size = H
            m.theta.inc = m.theta.inc * 1024.0 / W
            m.theta.size = W
            rcc = ra.RCCHipSpherical(hm)
            rcc.setTsb(T.identity())
            rcc.setModel(m)
            rcc.set_variant((kind & 15) | ((kind >> 4) << 13))
            ts = sorted(rcc.time_find(base, 30) for _ in range(7))
            row.append(ts[3] * 1e3)
            rcc.close()
        print("  kind %2d  %s   us" % (kind, "  ".join("%9.2f" % t for t in row)), flush=True)
    hm.release()
