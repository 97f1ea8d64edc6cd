#!/bin/bash
# Where a 1080p quadtree instance spends its time: the bench line of a workload (CFG=c5 by default) with the kernel cut short (ORBX_QT_STOP: 1 after the gather, 2 after
# the roots, 3 without the sorted phase, 4 without the sort itself).  Timing only - the keypoints are wrong with a stop point.
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/qtstop
for st in 0 1 2 3 4; do
  ORBX_QT_STOP=$st timeout -k 10 200 python bench.py --config ${CFG:-c5} --no-cpu-baseline --no-pcie > gpurun_out/qtstop/bench_${CFG:-c5}_$st.json 2> gpurun_out/qtstop/err_${CFG:-c5}_$st.txt || { tail -3 gpurun_out/qtstop/err_${CFG:-c5}_$st.txt; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/qtstop/bench_${CFG:-c5}_$st.json'))
print('stop=$st', 'ms', d['ms_per_step'], d.get('stages_ms'))"
done
