#!/usr/bin/env python
"""Where do the cycles of a single-scan find go?  Runs the instrumented traversal (rmclhip_debug_probe_find) on C2 and
prints, per wave and per tree depth, the split of a step into "loads issued -> data arrived" and "arithmetic + stack".
usage (GPU box): python tools/probe_find.py [sphere|room] [mode...]   mode bit0 = one-round-trip leaves, bit1 = LDS top"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmcl_amd as ra
ra.load_lab()   # experiments library: the kinds / kernels this tool compares are not all in the product
from rmcl_amd import synthetic as syn, types as T


def analyse(log, label):
    nt = log.shape[0]
    n = (log[:, 0, 0] & 0xFFFF).astype(int)
    xcc = (log[:, 0, 0] >> 16).astype(int)
    t0 = log[:, 0, 1].astype(np.int64)
    tot, endt, cal = np.zeros(nt, np.int64), np.zeros(nt, np.int64), []
    rows = []          # (wave, step, group, wait, alu, active, uniform)
    for w in range(nt):
        e = log[w, 1:n[w] + 1]
        t, meta = e[:, 0].astype(np.int64), e[:, 1]
        kind, act, uni, step = meta & 0xFF, (meta >> 8) & 0xFF, (meta >> 16) & 1, meta >> 20
        i = 0
        while i < len(e):
            k = kind[i]
            if k == 1 and i + 2 < len(e) and kind[i + 1] == 2 and kind[i + 2] == 3:
                rows.append((w, step[i], 0, t[i + 1] - t[i], t[i + 2] - t[i + 1], act[i], uni[i]))
                i += 3
            elif k == 4 and i + 2 < len(e) and kind[i + 1] == 5 and kind[i + 2] == 6:
                rows.append((w, step[i], 1, t[i + 1] - t[i], t[i + 2] - t[i + 1], act[i], 0))
                i += 3
            elif k == 4 and i + 1 < len(e) and kind[i + 1] == 6:
                rows.append((w, step[i], 2, t[i + 1] - t[i], 0, act[i], 0))
                i += 2
            elif k == 9 and i + 1 < len(e) and kind[i + 1] == 10:
                cal.append(t[i + 1] - t[i])
                i += 2
            elif k == 7:
                tot[w] = t[i]
                endt[w] = t[i + 1] if (i + 1 < len(e) and kind[i + 1] == 8) else t[i]
                i += 2
            else:
                i += 1
    R = np.array(rows, dtype=np.int64)
    c = float(np.median(cal))
    print("== %s: %d waves, entries/wave mean %.1f max %d (cap 255); one stamp costs %.0f cycles (median; subtracted below)" %
          (label, nt, n.mean(), n.max(), c))
    print("traversal cycles per wave incl. stamps: mean %.0f  median %.0f  p95 %.0f  max %.0f   stores %.0f" %
          (tot.mean(), np.median(tot), np.percentile(tot, 95), tot.max(), (endt - tot).mean()))
    node, leaf, leafb = R[R[:, 2] == 0], R[R[:, 2] == 1], R[R[:, 2] == 2]
    per_wave = lambda X, col: np.bincount(X[:, 0], weights=X[:, col], minlength=nt)
    cnt_wave = lambda X: np.bincount(X[:, 0], minlength=nt)
    print("node steps/wave mean %.1f max %d | wait/step %.0f  alu/step %.0f | per wave: wait %.0f alu %.0f" %
          (cnt_wave(node).mean(), cnt_wave(node).max(), node[:, 3].mean() - c, node[:, 4].mean() - c,
           per_wave(node, 3).mean() - c * cnt_wave(node).mean(), per_wave(node, 4).mean() - c * cnt_wave(node).mean()))
    if len(leaf):
        print("leaf tri-iterations/wave mean %.1f max %d | wait/iter %.0f  alu/iter %.0f | per wave: wait %.0f alu %.0f" %
              (cnt_wave(leaf).mean(), cnt_wave(leaf).max(), leaf[:, 3].mean() - c, leaf[:, 4].mean() - c,
               per_wave(leaf, 3).mean() - c * cnt_wave(leaf).mean(), per_wave(leaf, 4).mean() - c * cnt_wave(leaf).mean()))
    if len(leafb):
        print("leaf batches/wave mean %.1f max %d | cycles/batch %.0f | per wave %.0f" %
              (cnt_wave(leafb).mean(), cnt_wave(leafb).max(), leafb[:, 3].mean() - c, per_wave(leafb, 3).mean() - c * cnt_wave(leafb).mean()))
    print("stamps per wave %.1f => %.0f cycles of the mean are instrumentation; un-instrumented estimate %.0f cycles" %
          (n.mean(), n.mean() * c, tot.mean() - n.mean() * c))
    print("node step by index: idx  n  uniform%  active  wait  alu")
    for s in range(0, int(node[:, 1].max()) + 1):
        X = node[node[:, 1] == s]
        if len(X) < nt // 50:
            continue
        print("  %2d %6d  %5.1f  %5.1f  %6.0f %6.0f" % (s, len(X), 100 * X[:, 6].mean(), X[:, 5].mean(), X[:, 3].mean() - c, X[:, 4].mean() - c))
    u, d = node[node[:, 6] == 1], node[node[:, 6] == 0]
    print("uniform node steps: n %d wait %.0f alu %.0f | divergent: n %d wait %.0f alu %.0f" %
          (len(u), u[:, 3].mean() - c if len(u) else 0, u[:, 4].mean() - c if len(u) else 0, len(d), d[:, 3].mean() - c, d[:, 4].mean() - c))
    for lo, hi in ((1, 8), (9, 24), (25, 48), (49, 64)):
        X = d[(d[:, 5] >= lo) & (d[:, 5] <= hi)]
        if len(X):
            print("  divergent, %2d-%2d active lanes: n %6d wait %.0f alu %.0f" % (lo, hi, len(X), X[:, 3].mean() - c, X[:, 4].mean() - c))
    # launch timeline: the shader clock is per XCD, so offsets are taken inside each XCD
    print("timeline per XCD (cycles from the XCD's first wave start): start p50/max | end p50 / p95 / max")
    for x in range(8):
        sel = np.where(xcc == x)[0]
        if len(sel) == 0:
            continue
        st = (t0[sel] - t0[sel].min()) & 0xFFFFFFFF
        en = st + endt[sel]
        print("  xcc %d: %4d waves  start %6.0f / %6.0f | end %6.0f / %6.0f / %6.0f" %
              (x, len(sel), np.median(st), st.max(), np.median(en), np.percentile(en, 95), en.max()))
    w = int(np.argmax(tot))
    print("slowest wave %d (xcc %d): %d cycles, %d node steps, %d leaf rounds" %
          (w, xcc[w], tot[w], cnt_wave(node)[w], cnt_wave(leaf)[w] + cnt_wave(leafb)[w]))


if __name__ == "__main__":
    mesh = sys.argv[1] if len(sys.argv) > 1 else "sphere"
    modes = [int(a) for a in sys.argv[2:]] or [0, 1]
    ctx = ra.Context(0)
    v, f = syn.uv_sphere(100000) if mesh == "sphere" else syn.noisy_room(100000)
    hm = ra.import_hip_map(ctx, v, f)
    pose = syn.pose_c2_truth() if mesh == "sphere" else T.transform_from_rpy((1.5, -2.0, 1.6), (0.02, -0.03, 0.4))
    rcc = ra.RCCHipSpherical(hm)
    rcc.setTsb(T.identity())
    rcc.setModel(syn.model_c2())
    os.makedirs("gpurun_out", exist_ok=True)
    for mode in modes:
        log = rcc.debug_probe_find(pose, mode)
        np.savez_compressed("gpurun_out/probe_%s_mode%d.npz" % (mesh, mode), log=log)
        analyse(log, "%s-100k C2, probe mode %d" % (mesh, mode))
