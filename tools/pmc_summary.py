#!/usr/bin/env python
"""text summary of a tools/pmc_cmd.sh output directory: average kernel duration + per-launch counters.  usage: pmc_summary.py <dir> <kernel filter>"""
import collections
import csv
import glob
import sys

d, kf = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(d + "/pass1_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if kf in r["Kernel_Name"]:
            x = dur[r["Kernel_Name"][:90]]
            x[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); x[1] += 1
for k, (s, n) in dur.items():
    print("%-92s %9.2f us average over %d launches" % (k, s / n / 1e3, n))
vals = {}
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if kf in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        vals[k] = s / n
        print("%-24s %14.0f per launch (%d launches)" % (k, s / n, n))
if "SQ_THREAD_CYCLES_VALU" in vals and "SQ_ACTIVE_INST_VALU" in vals:
    print("# lanes active in VALU instructions = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.3f" % (vals["SQ_THREAD_CYCLES_VALU"] / 64.0 / vals["SQ_ACTIVE_INST_VALU"]))
    print("# VALU issue time = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs = %.0f cycles = %.1f us at 2.4 GHz" % (vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024, vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2400.0))
if "SQ_WAIT_ANY" in vals and "SQ_WAVE_CYCLES" in vals:
    print("# SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"]))
