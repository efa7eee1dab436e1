#!/bin/bash
# SQ / GRBM counters (two PMC passes, kernel-trace only) of the kernels matching a name filter while a command runs:
#   tools/pmc_cmd.sh <tag> <kernel name filter> <command...>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; KF=$2; shift; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pass$i --output-format csv -- "$@" > $OUT/pass$i.stdout 2>&1 || echo "pass $i failed" >> $OUT/errors.txt
done
echo "== $TAG ($KF): $*"
python - <<PY
import csv, glob, collections
dur = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$OUT/pass1_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "$KF" in r["Kernel_Name"]:
            d = dur[r["Kernel_Name"][:60]]
            d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
for k, (s, n) in dur.items():
    print("%-62s %9.2f us average over %d launches" % (k, s / n / 1e3, n))
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "$KF" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print("%-24s %14.0f per launch (%d launches)" % (k, s / n, n))
PY
