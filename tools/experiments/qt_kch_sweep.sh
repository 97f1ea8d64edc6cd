#!/bin/bash
# Quadtree register-chunk sweep: the bench line (stage times) of c2 and c5 with builds of the library that keep 32 / 16 / 8 64-key chunks
# in registers (variants/liborbx_kch*.so, built with -DQT3_KCH=..).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/kch
for v in "" kch16 kch8; do
  for c in c2 c5; do
    if [ -n "$v" ]; then export ORBX_LIB=$GRAFT_REPO_ROOT/wut-cuda-orb-slam3_amd/variants/liborbx_$v.so; else unset ORBX_LIB; fi
    timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-pcie > gpurun_out/kch/bench_${v:-base}_$c.json 2> gpurun_out/kch/err_${v:-base}_$c.txt || { tail -3 gpurun_out/kch/err_${v:-base}_$c.txt; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/kch/bench_${v:-base}_$c.json'))
print('${v:-base}', '$c', 'value', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
  done
done
