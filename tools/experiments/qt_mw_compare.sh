#!/bin/bash
# Multi-wave quadtree instances (k_quadtree_mw) against one wave per instance: parity with the choice forced, then single-frame
# latency and the c5 / c3 / c2 bench lines with ORBX_QT_MW = 0 / 1 / default.
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/qtmw
ORBX_QT_MW=1 timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/qtmw/pytest_mw1.log 2>&1 || { tail -30 gpurun_out/qtmw/pytest_mw1.log; exit 1; }
tail -2 gpurun_out/qtmw/pytest_mw1.log
for mw in 0 1; do
  ORBX_QT_MW=$mw timeout -k 10 120 python tools/time_single_frame.py > gpurun_out/qtmw/single_$mw.txt 2>&1 || { tail -5 gpurun_out/qtmw/single_$mw.txt; exit 1; }
  echo "single frame, ORBX_QT_MW=$mw:"; tail -4 gpurun_out/qtmw/single_$mw.txt
done
for c in c5 c3 c2; do
  for mw in 0 1; do
    ORBX_QT_MW=$mw timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-pcie > gpurun_out/qtmw/bench_${c}_$mw.json 2> gpurun_out/qtmw/err_${c}_$mw.txt || { tail -3 gpurun_out/qtmw/err_${c}_$mw.txt; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/qtmw/bench_${c}_$mw.json'))
print('$c mw=$mw', 'value', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
  done
done
