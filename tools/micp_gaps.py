#!/usr/bin/env python
"""per-correction timeline from a rocprofv3 kernel trace of tools/micp_trace.py: kernel durations and the gaps between the
launches of one correction and between corrections.  usage: micp_gaps.py <kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
short = lambda n: n.replace("rmclhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
seq = [(s, e, short(n)) for s, e, n in rows]
# group: a correction starts with k_find
groups, cur = [], []
for s, e, n in seq:
    if n.startswith("k_find") and cur:
        groups.append(cur); cur = []
    cur.append((s, e, n))
groups.append(cur)
groups = [g for g in groups if len(g) == len(groups[len(groups) // 2])][20:]
names = [n for _, _, n in groups[0]]
print("%d corrections of %d launches: %s" % (len(groups), len(names), " | ".join(names)))
for k, n in enumerate(names):
    d = np.array([g[k][1] - g[k][0] for g in groups]) / 1e3
    print("  %-42s duration  median %6.2f  mean %6.2f us" % (n, np.median(d), d.mean()))
    if k:
        gap = np.array([g[k][0] - g[k - 1][1] for g in groups]) / 1e3
        print("  %-42s gap before median %6.2f  mean %6.2f us" % ("", np.median(gap), gap.mean()))
span = np.array([g[-1][1] - g[0][0] for g in groups]) / 1e3
period = np.diff(np.array([g[0][0] for g in groups])) / 1e3
print("  first start -> last end: median %.2f us; period (start to next start): median %.2f us; host turn-around %.2f us" %
      (np.median(span), np.median(period), np.median(period) - np.median(span)))
