#!/bin/bash
# Produces the evidence committed under profiles/: GPU tests, bench line (with cpu_baseline), rocprofv3 kernel stats
# of the same bench command, PMC HBM traffic.  Usage on the GPU box: bash tools/profile_round.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/pytest_gpu.log 2>&1; echo "pytest exit=$?" >> gpurun_out/$tag/pytest_gpu.log
timeout -k 10 500 python bench.py --pcie > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/rocprof -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/rocprof_bench.json 2> gpurun_out/$tag/rocprof.err
cp $(find gpurun_out/$tag/rocprof -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/kernel_stats.csv
bash tools/collect_traffic.sh > gpurun_out/$tag/traffic.log 2>&1
cp gpurun_out/traffic/summary.json gpurun_out/$tag/hbm_traffic_pmc.json
tail -3 gpurun_out/$tag/pytest_gpu.log; cat gpurun_out/$tag/bench.json; cat gpurun_out/$tag/kernel_stats.csv
