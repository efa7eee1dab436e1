#!/bin/bash
# HBM traffic passes (rocprofv3 PMC, ONE counter per run, kernel-trace only -- never combined with sys / hip / hsa traces) for the
# hot kernels, in the exact configuration bench.py times.  usage (on the GPU box, via gpurun): bash tools/pmc_traffic.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/traffic_$1
mkdir -p $OUT
run() { # name counter cmd...
  name=$1; shift; ctr=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT -o $name -- "$@" > $OUT/$name.stdout 2>&1 || echo "$name failed" >> $OUT/errors.txt
}
# the automatic traversal (15 -> kind 23 for C2) and the quad traversal, each in its own pass pair; `find` as 2nd argument: only these
for v in 15 2; do
  run find_v${v}_fetch FETCH_SIZE python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --variant $v
  run find_v${v}_write WRITE_SIZE python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --variant $v
done
# round 6: variant 15 autotunes (kind 32 on the sphere); the rule's kind 23 in a pass pair of its own
run find_v15rule_fetch FETCH_SIZE python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-autotune
run find_v15rule_write WRITE_SIZE python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-autotune
[ "${2:-all}" = find ] && { ls $OUT; exit 0; }
# round 6: the C5 shard -- 125 000 particles x 256 beams on the 1 M-triangle sphere -- whose 2.7 ms had no traffic record
run pfc5_fetch FETCH_SIZE python tools/pf_c5_shard.py
run pfc5_write WRITE_SIZE python tools/pf_c5_shard.py
run pf_fetch FETCH_SIZE python bench.py --workload pf --steps 3 --warmup 1 --no-extras
run pf_write WRITE_SIZE python bench.py --workload pf --steps 3 --warmup 1 --no-extras
run red_fetch FETCH_SIZE python bench.py --steps 5 --warmup 2 --no-cpu-baseline
run red_write WRITE_SIZE python bench.py --steps 5 --warmup 2 --no-cpu-baseline
ls $OUT
