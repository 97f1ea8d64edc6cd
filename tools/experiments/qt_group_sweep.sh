#!/bin/bash
# levels per quadtree launch (ORBX_QT_PER_LEVEL: 0 = all levels in one launch, k >= 2 = k - 1 levels per launch) over the workloads
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/qtgrp
for cfg in ${CFGS:-c2 c3 c4 c5}; do
  for v in ${VALS:-0 3 4 5}; do
    ORBX_QT_PER_LEVEL=$v timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --no-pcie > gpurun_out/qtgrp/o.json 2> gpurun_out/qtgrp/err.txt || { tail -3 gpurun_out/qtgrp/err.txt; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/qtgrp/o.json'))
print('$cfg per_level=$v', 'fps', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
  done
done
