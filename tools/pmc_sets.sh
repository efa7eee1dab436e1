#!/bin/bash
# PMC passes with caller-chosen counter sets (one pass per set, kernel-trace only) of the kernels matching a name filter:
#   tools/pmc_sets.sh <tag> <kernel name filter> "<set 1>" "<set 2>" ... -- <command...>
# profiles/r04_pmc_pf_bound.txt: which pipe bounds k_pf_update_v3 (VALU classes, VMEM / TA, LDS conflicts)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; KF=$2; shift; shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pass$i --output-format csv -- "$@" > $OUT/pass$i.stdout 2>&1 || echo "pass $i failed: $set" >> $OUT/errors.txt
done
echo "== $TAG ($KF): $*"
[ -f $OUT/errors.txt ] && cat $OUT/errors.txt
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
        print("%-28s %16.0f per launch (%d launches)" % (k, s / n, n))
PY
