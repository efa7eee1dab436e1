#!/usr/bin/env python
"""Where a wave of find kind 31 spends its cycles: shader-clock stamps at the phase boundaries of the cooperative descent (clocked lab
instantiation; every stamp waits for what is in flight, so "arrived" phases are the round trips and the others are issue + LDS time).
usage (GPU box): python tools/descent_phases.py [sphere|room|sphere1m] [final_cap] [kind = 31 | 32]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()
from rmcl_amd import synthetic as syn, types as T, _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 31
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000) if mesh == "sphere" else (syn.noisy_room(100000) if mesh == "room" else syn.uv_sphere(1000000))
hm = ra.import_hip_map(ctx, v, f)
pose = T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4)) if mesh == "room" else syn.pose_c2_truth()
rcc = ra.RCCHipSpherical(hm)
rcc.setTsb(T.identity())
rcc.setModel(syn.model_c2())
rcc.set_traversal(kind)
if cap:
    _capi.check(_capi.lib().rmclhip_rcc_set_descent(rcc._h, cap, 24))
ms = sorted(rcc.time_find(pose, 50) for _ in range(5))[2]
w = rcc.debug_wave_clock(pose)
s = rcc._last_descent_stamps
ok = w[:, 1] != 0
w, s = w[ok].astype(np.int64), s[ok].astype(np.int64)
d = lambda a, b: (a - b) & 0xFFFFFFFF
total = d(w[:, 1], w[:, 0])
print("== %s kind %d%s: kernel %.2f us (un-instrumented); clocked waves: %d, cycles entry -> stores done mean %.0f p95 %.0f max %.0f" %
      (mesh, kind, " cap %d" % cap if cap else "", ms * 1e3, len(w), total.mean(), np.percentile(total, 95), total.max()))
descended = s[:, 10] != 0
print("   waves that descended: %d of %d; levels mean %.2f (1: %d, 2: %d, 3+: %d)" % (
    descended.sum(), len(w), s[descended, 12].mean(), (s[descended, 12] == 1).sum(), (s[descended, 12] == 2).sum(), (s[descended, 12] >= 3).sum()))
rows = []
def phase(name, a, b, mask=None):
    m = descended if mask is None else (descended & mask)
    x = d(a[m], b[m])
    rows.append((name, m.sum(), x.mean(), np.percentile(x, 50), np.percentile(x, 95)))
phase("kernel entry -> rays set up (pose, model, directions)", w[:, 4], w[:, 0])
phase("  -> descent entered", s[:, 0], w[:, 4])
phase("frontier table + tile planes arrive", s[:, 1], s[:, 0])
phase("cull 4 entries / lane, ballots", s[:, 2], s[:, 1])
phase("survivors to the LDS lists", s[:, 3], s[:, 2])
for L in range(3):
    have = s[:, 4 + 2 * L] != 0
    done = s[:, 5 + 2 * L] != 0
    prev = s[:, 3] if L == 0 else s[:, 3 + 2 * L]
    phase("level %d: predict, read refs from LDS, nodes arrive" % (L + 1), s[:, 4 + 2 * L], prev, have)
    phase("level %d: test, compact, LDS lists" % (L + 1), s[:, 5 + 2 * L], s[:, 4 + 2 * L], done)
last = np.where(s[:, 9] != 0, s[:, 9], np.where(s[:, 7] != 0, s[:, 7], np.where(s[:, 5] != 0, s[:, 5], s[:, 3])))
phase("leftover inner nodes appended, final list lane-resident", s[:, 10], last)
if kind == 32:
    phase("every ray: a bit per final leaf it enters, then those leaves", s[:, 11], s[:, 10])
    phase("per-ray traversal of the unexpanded inner nodes", w[:, 5], s[:, 11])
else:
    phase("every ray against every final entry (slab test, LDS pushes)", s[:, 11], s[:, 10])
    phase("per-ray leaf tests (trace_lane_bf_tail without the tail)", w[:, 5], s[:, 11])
phase("epilogue: record tail fetched, stores issued", w[:, 6], w[:, 5])
phase("stores complete", w[:, 1], w[:, 6])
print("   %-62s %6s %8s %8s %8s" % ("phase", "waves", "mean", "p50", "p95"))
for r in rows:
    print("   %-62s %6d %8.0f %8.0f %8.0f" % r)
print("   sum of the means %.0f" % sum(r[2] for r in rows))
