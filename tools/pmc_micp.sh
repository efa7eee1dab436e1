#!/bin/bash
# PMC passes (one counter per run, kernel-trace only) over 300 MICP corrections of tools/micp_trace.py: HBM traffic and MFMA activity of the
# find with the moment epilogue and of the loop launch.  usage (GPU box, via gpurun): bash tools/pmc_micp.sh <tag> [mode]
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_micp_$1
MODE=${2:-1}
mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES; do
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d $OUT -o $ctr -- python tools/micp_trace.py 10 $MODE > $OUT/$ctr.stdout 2>&1 || echo "$ctr failed" >> $OUT/errors.txt
done
python - <<'PY' > $OUT/summary.txt
import sqlite3, glob, os, sys
out = os.environ.get("OUT_DIR") or sorted(glob.glob("gpurun_out/pmc_micp_*"))[-1]
print("# rocprofv3 --pmc <one counter per pass> --kernel-trace -- python tools/micp_trace.py 10 <mode>; average per dispatch")
for db in sorted(glob.glob(out + "/*_results.db")):
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select k.name, p.counter_name, avg(v), count(*) from (select dispatch_id, counter_name, sum(counter_value) as v from pmc_events group by dispatch_id, counter_name) p join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name").fetchall()
    except Exception as e:
        print(os.path.basename(db), "unreadable:", e); continue
    for name, ctr, v, n in rows:
        if "k_find" in name or "k_micp" in name:
            print("%-28s %-60s avg %14.1f over %d dispatches" % (ctr, name.replace("rmclhip::(anonymous namespace)::", "")[:60], v, n))
PY
cat $OUT/errors.txt 2>/dev/null
