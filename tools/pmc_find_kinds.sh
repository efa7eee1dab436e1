#!/bin/bash
# SQ / GRBM counters of one find traversal kind on the C2 scan (experiments library loaded): tools/pmc_find_kinds.sh <mesh> <kind> [<kind> ...]
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
MESH=$1; shift
for K in "$@"; do
  OUT=gpurun_out/pmcf_${MESH}_$K
  mkdir -p $OUT
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pass$i --output-format csv -- python tools/find_trace.py $MESH $K > $OUT/pass$i.stdout 2>&1 || echo "pass $i failed" >> $OUT/errors.txt
  done
  echo "== $MESH kind $K"
  python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_find" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print("%-24s %14.0f per launch (%d launches)" % (k, s / n, n))
PY
done
