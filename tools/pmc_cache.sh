#!/bin/bash
# L1/L2 request counters of one kernel (own PMC passes, --kernel-trace only):
#   tools/pmc_cache.sh <tag> <workload c2|pf> <kernel name filter> [variant]
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcc_$1
mkdir -p $OUT
WL=${2:-pf}; KF=${3:-k_pf_update}
STEPS=30; [ "$WL" = pf ] && STEPS=4
CMD="python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras --variant ${4:-15} --workload $WL"
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pass$i --output-format csv -- $CMD > $OUT/pass$i.stdout 2>&1 || echo "pass $i failed: $set" >> $OUT/errors.txt
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "$KF" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print("%-32s %16.0f per launch (%d launches)" % (k, s / n, n))
PY
[ -f $OUT/errors.txt ] && cat $OUT/errors.txt
