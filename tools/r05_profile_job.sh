#!/bin/bash
# Round-4 evidence in one gpurun call: the default bench line, the driver-form line, rocprofv3 kernel stats of the headline loop and of
# the whole bench, and the HBM traffic passes.  Everything lands under gpurun_out/r05/ and is copied into profiles/ by hand.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r05_profile_job.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form_steps20_warmup5.json 2> $O/bench_driver_form.err
rocprofv3 --kernel-trace --stats -d $O/trh -o t -- python bench.py --no-cpu-baseline --no-extras > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err
python tools/prof_summary.py $(find $O/trh -name "*results.db" | head -1) 2>&1 | cut -c1-170 > $O/kernel_stats_headline.txt
rocprofv3 --kernel-trace --stats -d $O/trf -o t -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python tools/prof_summary.py $(find $O/trf -name "*results.db" | head -1) 2>&1 | head -60 | cut -c1-170 > $O/kernel_stats_full_bench.txt
rm -rf $O/trh $O/trf
bash tools/pmc_traffic.sh r05 > $O/traffic_passes.log 2>&1
python tools/traffic_from_pmc.py gpurun_out/traffic_r05 $O/traffic_r05.json > $O/traffic_summary.txt 2>&1
rm -rf gpurun_out/traffic_r05
tail -3 $O/kernel_stats_headline.txt; head -12 $O/kernel_stats_full_bench.txt; cat $O/traffic_summary.txt | tail -12
