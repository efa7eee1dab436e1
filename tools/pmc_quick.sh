#!/bin/bash
# two PMC passes (instruction mix, issue/wait cycles) of the find kernel for one traversal variant:
#   tools/pmc_quick.sh <tag> <variant> [workload c2|pf] [kernel name filter]
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq_$1
mkdir -p $OUT
WL=${3:-c2}
KF=${4:-k_find}
STEPS=30; [ "$WL" = pf ] && STEPS=4
CMD="python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras --variant ${2:-15} --workload $WL"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS"; do
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
        print("%-24s %14.0f per launch (%d launches)" % (k, s / n, n))
PY
