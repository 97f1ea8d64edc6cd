#!/bin/bash
# k_quadtree<KCH> choice on a workload (CFG, default c4): default rule against ORBX_QT_KCH=8 / 32, and the level grouping ORBX_QT_PER_LEVEL
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/qtkch
for v in "" "ORBX_QT_KCH=8" "ORBX_QT_KCH=32" "ORBX_QT_PER_LEVEL=2" "ORBX_QT_PER_LEVEL=4" "ORBX_QT_PER_LEVEL=0"; do
  env $v timeout -k 10 200 python bench.py --config ${CFG:-c4} --no-cpu-baseline --no-pcie > gpurun_out/qtkch/o.json 2> gpurun_out/qtkch/err.txt || { tail -3 gpurun_out/qtkch/err.txt; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/qtkch/o.json'))
print('${CFG:-c4} [$v]', 'fps', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
done
