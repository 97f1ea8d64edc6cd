mkdir -p gpurun_out/r02i
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02i/pytest_gpu.log 2>&1; echo "pytest exit=$?" >> gpurun_out/r02i/pytest_gpu.log; tail -3 gpurun_out/r02i/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; o=gpurun_out/r02i
for c in c5 c3; do
  timeout -k 10 500 python bench.py --config $c > $o/bench_$c.json 2> $o/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/rocprof_$c -- python3 bench.py --config $c --batches-per-step 1 --steps 5 --warmup 1 --no-cpu-baseline --no-pcie > $o/rocprof_bench_$c.json 2> $o/rocprof_$c.err
  cp $(find $o/rocprof_$c -name "*kernel_stats.csv" | head -1) $o/kernel_stats_$c.csv; rm -rf $o/rocprof_$c
done
timeout -k 10 300 python tools/time_matcher.py > $o/matcher_knn2.json 2> $o/matcher_knn2.err
tail -c 700 $o/bench_c5.json
