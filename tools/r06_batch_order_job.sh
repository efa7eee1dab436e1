#!/bin/bash
# VERDICT r5 #7b: pose batches in world order -- A/B incl. the 10 M-face map, the kernels of one ordered batch, HBM traffic before / after.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r06_batch_order_job.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_batch_order; mkdir -p $O
python tools/batch_order_ab.py 100000 1000000 10000000 --room > $O/ab.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python tools/batch_order_ab.py 1000000 --pmc on > /dev/null 2> $O/tr.err
python tools/prof_summary.py $(find $O/tr -name "*results.db" | head -1) 2>&1 | head -24 | cut -c1-170 > $O/kernels_ordered_batch_1m.txt
rm -rf $O/tr
for nf in 1000000 10000000; do
  for form in off on; do
    bash tools/pmc_sets.sh bo_${nf}_$form "k_find<1u, 24" "FETCH_SIZE" "WRITE_SIZE" -- python tools/batch_order_ab.py $nf --pmc $form > $O/pmc_${nf}_$form.txt 2>&1
    rm -rf gpurun_out/pmc_bo_${nf}_$form
  done
done
cat $O/ab.txt; cat $O/kernels_ordered_batch_1m.txt; for f in $O/pmc_*.txt; do echo "== $f"; tail -n 4 $f; done
