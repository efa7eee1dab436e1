#!/bin/bash
# lane utilisation of the persistent particle-filter kernel per refill threshold (kernel template arguments
# distinguish the variants inside one run of tools/pf_explore.py)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcv
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES -d $OUT -o p --output-format csv -- python tools/pf_explore.py sphere > $OUT/stdout.txt 2>&1
python - <<PY
import csv, glob, collections, re
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        if "k_pf_update" not in r["Kernel_Name"]:
            continue
        m = re.search(r"k_pf_update\w*<[^>]*>", r["Kernel_Name"])
        a = acc[m.group(0) if m else r["Kernel_Name"][:40]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k in sorted(acc):
        d = {c: s / n for c, (s, n) in acc[k].items()}
        print("%-40s lanes active %.3f  VALU insts %.3g  VALU busy qc %.3g" % (k, d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]), d["SQ_INSTS_VALU"], d["SQ_ACTIVE_INST_VALU"]))
PY
