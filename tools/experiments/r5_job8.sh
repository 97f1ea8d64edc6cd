#!/bin/bash
# round 5, GPU job 8: what a fourth resident workgroup per CU is worth to k_pyr_fast at EQUAL tile width
# (narrow tiles that fit 40 KB, with and without LDS padding that takes the fourth workgroup away; wide tiles 3 -> 2 for the slope)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job8; mkdir -p $o
run() { env $1 python3 bench.py --batches-per-step 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['stages_ms']['pyramid+FAST'], d['ms_per_step'], d['value'])"; }
for i in 1 2; do
run ORBX_PF_LDS_FIT=1
run ORBX_PF_LDS_FIT=40448
run "ORBX_PF_LDS_FIT=40448 ORBX_PF_LDS_PAD=13000"
run "ORBX_PF_LDS_FIT=1 ORBX_PF_LDS_PAD=28000"
run "ORBX_PF_LDS_FIT=40448 ORBX_PF_LDS_PAD=36000"
done 2>&1 | tee $o/ab.txt
