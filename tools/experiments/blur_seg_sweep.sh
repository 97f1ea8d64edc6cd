#!/bin/bash
# The blur's rows per workgroup (ORBX_BLUR_SEG = 4 / 8 blocks of 16 rows) and width (ORBX_BLUR_WAVES = 4 / 8 strips of 16 columns) over the workloads
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/blurseg
for cfg in ${CFGS:-c2 c3 c4 c5}; do
  for v in "4 4" "8 4" "4 8" "8 8"; do
    set -- $v
    ORBX_BLUR_SEG=$1 ORBX_BLUR_WAVES=$2 timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --no-pcie > gpurun_out/blurseg/${cfg}_$1_$2.json 2> gpurun_out/blurseg/err.txt || { tail -3 gpurun_out/blurseg/err.txt; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/blurseg/${cfg}_$1_$2.json'))
print('$cfg seg=$1 waves=$2', 'fps', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
  done
done
