#!/usr/bin/env python
"""Does the ORDER of the particles matter on maps that do not fit the L2s?  (round 5)  Config C5's shard (125 000 particles x 256 beams,
sphere-1M) and C4 on sphere-100k / room-100k with the cloud as drawn (random order), sorted by a Morton key of (x, y, yaw) before the
upload (default dealing: a workgroup = 8 consecutive particles), and with the particle-minor dealing on that order (rmclhip_pf_set_mapping 1).
   usage: python tools/pf_sorted_cloud.py"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rmcl_amd as ra  # noqa: E402
from rmcl_amd import synthetic as syn, types as T  # noqa: E402

ctx = ra.Context(0)
beams = ra.beams_from_points(syn.model_directions(syn.model_pf16()) * np.float32(6.0))
for mesh, n, bb in (("sphere1m", 125000, ((-5, -5, -1), (5, 5, 1))), ("sphere100k", 100000, ((-5, -5, -1), (5, 5, 1))), ("room100k", 100000, ((-9, -9, 0.3), (9, 9, 3)))
                    ):
    v, f = syn.noisy_room(100000) if mesh.startswith("room") else syn.uv_sphere({"sphere1m": 1000000, "sphere100k": 100000, "sphere10m": 10000000}[mesh])
    hm = ra.import_hip_map(ctx, v, f)
    poses, attrs = syn.uniform_particles(n, seed=42, bb_min=bb[0] + (0, 0, -math.pi), bb_max=bb[1] + (0, 0, math.pi))
    order = syn.morton_order_xy_yaw(poses)
    rng = np.random.RandomState(1)

    def coarse(bits):
        """cells of `bits` bits per dimension in Morton order, RANDOM order inside a cell (what a counting sort with atomics leaves)"""
        o = syn.morton_order_xy_yaw(poses, bits=bits)
        # morton_order sorts stably by the coarse key; shuffle inside equal-key runs
        x, y = poses["t"]["x"].astype(np.float64), poses["t"]["y"].astype(np.float64)
        yaw = 2.0 * np.arctan2(poses["R"]["z"].astype(np.float64), poses["R"]["w"].astype(np.float64))
        q = lambda a: np.clip((a - a.min()) / (a.max() - a.min()) * (2 ** bits - 1), 0, 2 ** bits - 1).astype(np.int64)
        cell = (q(yaw) * (1 << bits) + q(x)) * (1 << bits) + q(y)
        return o[np.lexsort((rng.rand(len(o)), np.searchsorted(np.unique(cell), cell[o])))]   # cells in (yaw, x, y) order

    row = []
    for name, P, mapping in (("as drawn", poses, None), ("as drawn + pm16", poses, (1, 16)), ("Morton-sorted", poses[order], None), ("sorted + pm16", poses[order], (1, 16)),
                             ("sorted + pm32", poses[order], (1, 32)), ("sorted + pm8", poses[order], (1, 8)),
                             ("cells 4 bits + pm16", poses[coarse(4)], (1, 16)), ("cells 5 bits + pm16", poses[coarse(5)], (1, 16))):
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.setInput(beams, T.identity())
        if mapping:
            upd.set_mapping(mapping[0], mapping[1], None)
        d_p, d_a = ra.DeviceArray.from_host(ctx, P), ra.DeviceArray.from_host(ctx, attrs)
        upd.time_update(d_p, d_a, n, iters=1)
        ms = sorted(upd.time_update(d_p, d_a, n, iters=3) for _ in range(5))[2]
        row.append("%s %.4f ms" % (name, ms))
        upd.close()
    print("%-10s %d particles: " % (mesh, n) + "   ".join(row), flush=True)
    hm.release()
