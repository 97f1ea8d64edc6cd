#!/bin/bash
# Produces the evidence committed under profiles/: GPU tests, the bench line of every workload (c2 with cpu_baseline and
# PCIe-inclusive rates), rocprofv3 kernel stats of the same bench commands, PMC passes (HBM traffic, VALU instruction count).
# Usage on the GPU box: bash tools/profile_round.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
if [ -z "$SKIP_TESTS" ]; then   # (SKIP_TESTS=1: the GPU tests ran in a call of their own)
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; prc=$?; echo "pytest exit=$prc" >> $o/pytest_gpu.log
tail -3 $o/pytest_gpu.log
if [ $prc -ne 0 ]; then echo "!!!! profile_round: the GPU tests FAILED - no evidence is collected from a failing build !!!!"; exit 1; fi
fi
if [ -n "$WITH_STRESS" ]; then   # (WITH_STRESS=1: the randomised sweeps too; a failing sweep stops the round)
bash tools/stress_round.sh $tag || { echo "!!!! profile_round: a stress sweep FAILED !!!!"; exit 1; }
fi
bash tools/collect_traffic.sh > $o/traffic.log 2>&1
cp gpurun_out/traffic/summary.json $o/hbm_traffic_pmc.json
cp gpurun_out/traffic/pyr_fast_counters.json $o/pyr_fast_counters.json
mkdir -p profiles && cp $o/pyr_fast_counters.json profiles/pyr_fast_counters_latest.json   # bench.py reads this one
for c in c2 c3 c4 c5; do
  timeout -k 10 500 python bench.py --config $c > $o/bench_$c.json 2> $o/bench_$c.err || echo "bench $c failed" >> $o/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/rocprof_$c -- python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-pcie > $o/rocprof_bench_$c.json 2> $o/rocprof_$c.err
  cp $(find $o/rocprof_$c -name "*kernel_stats.csv" | head -1) $o/kernel_stats_$c.csv
  rm -rf $o/rocprof_$c
  echo "== $c"; cat $o/bench_$c.json; head -12 $o/kernel_stats_$c.csv
done
