#!/usr/bin/env python
"""GPU perf exploration of the particle-filter update (config C4): kernel generations and their knobs.
usage: python tools/pf_explore.py [sphere|room|sphere1m] [variant ...]   (sphere1m: the 1 M-triangle map of config C5, 125 k particles)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T

LEGACY, BIG, MAPTREE = 256, 512, 1024
NAMES = {64 | LEGACY | MAPTREE: "round-2 kernel, map tree (leaves<=4)", 64 | LEGACY: "round-2 kernel, filter tree (leaves<=2)",
         64 | MAPTREE: "round-3 kernel, map tree", 64: "round-3 kernel (default)", 48: "round-3, refill at 32 idle",
         32: "round-3, refill at 16 idle", 64 | BIG: "round-3, 4096 rays per workgroup"}

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
variants = [int(a) for a in sys.argv[2:]] or list(NAMES)
ctx = ra.Context(0)
v, f = syn.uv_sphere(1000000) if mesh == "sphere1m" else (syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000))
hm = ra.import_hip_map(ctx, v, f)
print("map", hm.info())
dirs = syn.model_directions(syn.model_pf16())
for n_particles, n_beams in (((125000, 256),) if mesh == "sphere1m" else ((100000, 256), (100000, 100))):
    if mesh in ("sphere", "sphere1m"):
        poses, attrs = syn.uniform_particles(n_particles, seed=42, bb_min=(-5, -5, -1, 0, 0, -math.pi), bb_max=(5, 5, 1, 0, 0, math.pi))
    else:
        poses, attrs = syn.uniform_particles(n_particles, seed=42, bb_min=(-9, -9, 0.3, 0, 0, -math.pi), bb_max=(9, 9, 3, 0, 0, math.pi))
    sel = np.linspace(0, len(dirs) - 1, n_beams).astype(int)
    beams = ra.beams_from_points(dirs[sel] * np.float32(6.0))
    d_poses, d_attrs = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
    for variant in variants:
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.set_variant(variant)
        upd.setInput(beams, T.identity())
        ms = min(upd.time_update(d_poses, d_attrs, n_particles, iters=4) for _ in range(3))
        rays = n_particles * n_beams
        print("%s particles=%7d beams=%4d variant=%4d %-42s %8.3f ms  %7.3f Grays/s" %
              (mesh, n_particles, n_beams, variant, NAMES.get(variant, ""), ms, rays / ms / 1e6), flush=True)
        upd.close()
