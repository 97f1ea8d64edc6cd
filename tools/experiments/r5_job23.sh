#!/bin/bash
# round 5, GPU job 23: k_orient_desc prefetch depth (OD_PF) / moment-pass batch (OD_KPB) against registers = resident waves
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job23; mkdir -p $o
run() { ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$1 python3 bench.py --config ${2:-c2} --batches-per-step 2 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ${2:-c2}', d['stages_ms']['orient+desc'], d['ms_per_step']/2, d['value'])"; }
for i in 1 2; do for l in liborbx.so liborbx_od28.so liborbx_od24.so liborbx_od14.so liborbx_od34.so liborbx_od48.so; do run $l; done; done 2>&1 | tee $o/ab.txt
for l in liborbx.so liborbx_od28.so liborbx_od24.so; do run $l c4; run $l c5; done 2>&1 | tee -a $o/ab.txt
