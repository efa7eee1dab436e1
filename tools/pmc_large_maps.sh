#!/bin/bash
# Round 5: the reference's 1 M / 10 M-face rows under rocprofv3 -- kernel durations (--kernel-trace --stats) and, in passes of their own
# (ONE counter set per run, kernel-trace only), the L2's memory-side traffic of k_find (C2 scan, 16 poses in turn) and of the v1 batch.
# Plus the calibration of FETCH_SIZE on the access patterns of a BVH walk (tools/ubench/gather_calib.hip).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/pmc_large_maps.sh r05'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_large_$1
mkdir -p $O
run() { # name counters cmd...
  name=$1; shift; ctr=$1; shift
  if [ "$ctr" = stats ]; then
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/$name -o t -- "$@" > $O/$name.stdout 2>&1 || echo "$name failed" >> $O/errors.txt
  else
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $O/$name -o t -- "$@" > $O/$name.stdout 2>&1 || echo "$name failed" >> $O/errors.txt
  fi
  python tools/prof_summary.py $(find $O/$name -name "*results.db" | head -1) 2>&1 | grep -v "^$" | cut -c1-200 > $O/$name.txt
  rm -rf $O/$name
}
run calib_fetch FETCH_SIZE tools/ubench/gather_calib
run calib_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" tools/ubench/gather_calib
for nf in 100000 1000000 10000000; do
  for what in find_rot v1; do
    run ${what}_${nf}_stats stats python tools/large_maps.py --faces $nf --only $what --no-parity
    run ${what}_${nf}_fetch FETCH_SIZE python tools/large_maps.py --faces $nf --only $what --no-parity
    run ${what}_${nf}_write WRITE_SIZE python tools/large_maps.py --faces $nf --only $what --no-parity
    run ${what}_${nf}_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" python tools/large_maps.py --faces $nf --only $what --no-parity
  done
done
ls $O
