#!/bin/bash
# HBM traffic of every kernel from rocprofv3 PMC counters (MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE in
# separate passes — they do not fit one TCC pass; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950).
# Writes gpurun_out/traffic/{fetch,write}.csv summaries; tools/summarize_traffic.py turns them into profiles/*.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic/$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/traffic/$c.log 2>&1
done
python3 tools/summarize_traffic.py gpurun_out/traffic > gpurun_out/traffic/summary.json
cat gpurun_out/traffic/summary.json
