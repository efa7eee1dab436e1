#!/usr/bin/env python
"""What could intra-CU ray stealing win for one find()?  Block-level model (tools/wavesim.c: orc_blocksim): the 8 tiles a CU
receives at C2 run as 8 waves with their own clocks; a finished wave takes half of the walking rays of the busiest wave.
Prints the slowest block with and without stealing.   usage: python tools/blocksim.py [sphere|room] [adjacent|strided]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import rmcl_amd as ra  # noqa: E402
import wavesim as ws  # noqa: E402
from rmcl_amd import synthetic as syn  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
layout = sys.argv[2] if len(sys.argv) > 2 else "adjacent"
v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
info, nodes, tris = ra.build_bvh_host(v, f)
model, O, dm = ws.rays_c2(mesh)
H, W = model.phi.size, model.theta.size
img = dm.reshape(H, W, 3)
tiles = [np.ascontiguousarray(img[ty:ty + 8, tx:tx + 8].reshape(-1, 3)) for ty in range(0, H, 8) for tx in range(0, W, 8)]
nt = len(tiles)
order = np.arange(nt) if layout == "adjacent" else np.arange(nt).reshape(8, nt // 8).T.reshape(-1)
L = ws.simlib()
L.orc_blocksim.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p]
# clocks measured on the device: node iteration 730 (lane) / 600 (quad tail), leaf round 1400 / 800
for thief, victim, minv in ((1500.0, 300.0, 8), (3000.0, 600.0, 8), (1500.0, 300.0, 24)):
    costs = np.array([730.0, 600.0, 1400.0, 800.0, thief, victim], np.float64)
    res = []
    for b in range(0, nt, 8):
        d = np.ascontiguousarray(np.concatenate([tiles[i] for i in order[b:b + 8]]))
        o = np.ascontiguousarray(np.tile(O, (len(d), 1)))
        out = np.zeros(3, np.float64)
        L.orc_blocksim(nodes.ctypes.data, tris.ctypes.data, o.ctypes.data, d.ctypes.data, 8, float(model.range.max), costs.ctypes.data, minv, out.ctypes.data)
        res.append(out.copy())
    r = np.array(res)
    print("%s %s thief %4.0f victim %3.0f min_victim %2d: slowest block %6.0f -> %6.0f clocks (%.0f %%), mean block %6.0f -> %6.0f, steals/block %.1f" %
          (mesh, layout, thief, victim, minv, r[:, 0].max(), r[:, 1].max(), 100 * (1 - r[:, 1].max() / r[:, 0].max()), r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean()))
