#!/bin/bash
# rocprofv3 kernel stats of the default bench command of every workload (4 distinct batches per step, 1 warm-up + 5 steps = 24 launch sets)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04g; mkdir -p $o
for c in c2 c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/rocprof_$c -- python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-pcie > $o/rocprof_bench_$c.json 2> $o/rocprof_$c.err
  cp $(find $o/rocprof_$c -name "*kernel_stats.csv" | head -1) $o/kernel_stats_$c.csv
  rm -rf $o/rocprof_$c
  head -3 $o/kernel_stats_$c.csv
done
