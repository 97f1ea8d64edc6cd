#!/bin/bash
# k_pyr_fast chain cut into groups of frames on streams of their own (ORBX_FRONT_PARTS), staggered by one level or not (ORBX_FRONT_STAGGER)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fparts
for cfg in "1 1" "2 1" "2 0" "3 1" "4 1"; do
  set -- $cfg
  export ORBX_FRONT_PARTS=$1 ORBX_FRONT_STAGGER=$2
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-pcie > gpurun_out/fparts/bench_$1_$2.json 2> gpurun_out/fparts/err_$1_$2.txt || { tail -3 gpurun_out/fparts/err_$1_$2.txt; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/fparts/bench_$1_$2.json'))
print('parts $1 stagger $2', 'value', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
done
