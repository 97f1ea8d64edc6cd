#!/bin/bash
# round 5, GPU job 32: ORBX_QT_EARLY=6 against 0 on every workload, alternating
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job32; mkdir -p $o
run() { env $2 python3 bench.py --config $1 --batches-per-step 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step']/4, d['value'])"; }
for c in c2 c3 c4 c5; do for i in 1 2 3; do run $c ORBX_QT_EARLY=0; run $c ORBX_QT_EARLY=6; done; done 2>&1 | tee $o/ab.txt
run c2 ORBX_QT_EARLY=7 | tee -a $o/ab.txt
