#!/bin/bash
# round 4 evidence, part b: bench line + rocprofv3 kernel stats of c3, c4, c5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
for c in c3 c4 c5; do
  timeout -k 10 400 python bench.py --config $c > $o/bench_$c.json 2> $o/bench_$c.err || echo "bench $c failed" >> $o/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/rocprof_$c -- python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-pcie > $o/rocprof_bench_$c.json 2> $o/rocprof_$c.err
  cp $(find $o/rocprof_$c -name "*kernel_stats.csv" | head -1) $o/kernel_stats_$c.csv
  rm -rf $o/rocprof_$c
  echo "== $c"; python3 -c "
import json; d=json.load(open('$o/bench_$c.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('metric_8d_frames_per_s'), d.get('pcie_inclusive_frames_per_s'), d['cpu_baseline']['value'], d.get('matcher',{}).get('ms_per_call'), d['roofline']['frac'], d['roofline']['avg_duration_ms'])"
done
