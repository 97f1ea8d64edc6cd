#!/bin/bash
# round 5, GPU job 31: the first quadtree launch group (levels 0 - 1) next to the fused launches of the small levels (ORBX_QT_EARLY=k)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job31; mkdir -p $o
timeout -k 10 300 env ORBX_QT_EARLY=4 python3 tests/tools/gpu_check.py 640 480 1000 700 > $o/check.txt 2>&1; echo "parity exit=$?"; tail -1 $o/check.txt
run() { env $2 python3 bench.py --config $1 --batches-per-step 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['stages_ms'], d['ms_per_step']/4, d['value'])"; }
for i in 1 2; do for k in 0 2 3 4 5 6; do run c2 ORBX_QT_EARLY=$k; done; done 2>&1 | tee $o/ab.txt
for k in 0 4 5; do run c4 ORBX_QT_EARLY=$k; run c5 ORBX_QT_EARLY=$k; done 2>&1 | tee -a $o/ab.txt
