#!/bin/bash
# quadtree launch groups dealt over ORBX_QT_STREAMS extra streams (0: behind each other on the back stream), per grouping ORBX_QT_PER_LEVEL;
# VALS = list of streams,per_level
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/qtstr
for cfg in ${CFGS:-c2 c3 c4 c5}; do
  for v in ${VALS:-0,3 1,3 2,3 1,4 2,4 1,0}; do
    IFS=, read a b <<< $v; set -- $a $b
    ORBX_QT_STREAMS=$1 ORBX_QT_PER_LEVEL=$2 timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --no-pcie > gpurun_out/qtstr/o.json 2> gpurun_out/qtstr/err.txt || { tail -3 gpurun_out/qtstr/err.txt; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/qtstr/o.json'))
print('$cfg streams=$1 per_level=$2', 'fps', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
  done
done
