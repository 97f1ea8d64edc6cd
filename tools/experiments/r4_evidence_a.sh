#!/bin/bash
# round 4 evidence, part a: PMC passes (traffic, VALU count) + bench line + rocprofv3 kernel stats of c2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
bash tools/collect_traffic.sh > $o/traffic.log 2>&1
cp gpurun_out/traffic/summary.json $o/hbm_traffic_pmc.json
cp gpurun_out/traffic/pyr_fast_counters.json $o/pyr_fast_counters.json
mkdir -p profiles && cp $o/pyr_fast_counters.json profiles/pyr_fast_counters_latest.json
for c in c2; do
  timeout -k 10 500 python bench.py --config $c > $o/bench_$c.json 2> $o/bench_$c.err || echo "bench $c failed" >> $o/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/rocprof_$c -- python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-pcie > $o/rocprof_bench_$c.json 2> $o/rocprof_$c.err
  cp $(find $o/rocprof_$c -name "*kernel_stats.csv" | head -1) $o/kernel_stats_$c.csv
  rm -rf $o/rocprof_$c
  echo "== $c"; cat $o/bench_$c.json; head -12 $o/kernel_stats_$c.csv
done
