#!/bin/bash
# Round-6 evidence in one gpurun call: the default bench line, the driver-form line, rocprofv3 kernel stats of the headline loop (autotuned
# and on the rule), ONE rocprofv3 run per bench section (VERDICT r5 #7c: per-kernel averages attributable to a workload), the HBM traffic
# passes.  Everything lands under gpurun_out/r06/ and is copied into profiles/ by hand.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r06_profile_job.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form_steps20_warmup5.json 2> $O/bench_driver_form.err
stats() { # tag cmd...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/tr_$tag -o t -- "$@" > $O/${tag}_under_rocprof.json 2> $O/${tag}_under_rocprof.err
  python tools/prof_summary.py $(find $O/tr_$tag -name "*results.db" | head -1) 2>&1 | head -40 | cut -c1-170 > $O/kernel_stats_$tag.txt
  rm -rf $O/tr_$tag
}
stats headline python bench.py --no-cpu-baseline --no-extras
stats headline_rule_kind23 python bench.py --no-cpu-baseline --no-extras --no-autotune
stats workload_pf_c4 python bench.py --workload pf --steps 20 --warmup 3 --no-extras --no-cpu-baseline
stats workload_pf_c5_shard python tools/pf_c5_shard.py 20
stats workload_room100k_find python tools/find_variants.py 23
stats full_bench python bench.py --no-cpu-baseline
cp profiles/traffic.json $O/traffic_r06.json
bash tools/pmc_traffic.sh r06 > $O/traffic_passes.log 2>&1
python tools/traffic_from_pmc.py gpurun_out/traffic_r06 $O/traffic_r06.json > $O/traffic_summary.txt 2>&1
rm -rf gpurun_out/traffic_r06
tail -3 $O/kernel_stats_headline.txt; tail -3 $O/kernel_stats_headline_rule_kind23.txt; head -8 $O/kernel_stats_workload_pf_c5_shard.txt; tail -30 $O/traffic_summary.txt
