#!/bin/bash
# round 5, GPU job 26: quadtree launch groups / streams re-swept on the final kernels (c2, c4)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job26; mkdir -p $o
run() { env $2 python3 bench.py --config $1 --batches-per-step 2 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['stages_ms']['quadtree'], d['stages_ms']['blur'], d['ms_per_step']/2, d['value'])"; }
for c in c2 c4; do
for e in "ORBX_QT_STREAMS=0 ORBX_QT_PER_LEVEL=3" "ORBX_QT_STREAMS=1 ORBX_QT_PER_LEVEL=3" "ORBX_QT_STREAMS=0 ORBX_QT_PER_LEVEL=4" "ORBX_QT_STREAMS=1 ORBX_QT_PER_LEVEL=4" "ORBX_QT_STREAMS=2 ORBX_QT_PER_LEVEL=4" "ORBX_QT_STREAMS=1 ORBX_QT_PER_LEVEL=5" "ORBX_QT_STREAMS=3 ORBX_QT_PER_LEVEL=3" "ORBX_QT_STREAMS=0 ORBX_QT_PER_LEVEL=0" "ORBX_QT_STREAMS=0 ORBX_QT_PER_LEVEL=2"; do run $c "$e"; done; done 2>&1 | tee $o/ab.txt
