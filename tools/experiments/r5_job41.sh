#!/bin/bash
# round 5, GPU job 41: blur workgroup width re-swept on the final kernel (ORBX_BLUR_WAVES 4 (default) / 8, ORBX_BLUR_SEG 4 / 8)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job41; mkdir -p $o
run() { env $2 python3 bench.py --config $1 --batches-per-step 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['stages_ms']['quadtree'], d['stages_ms']['blur'], d['ms_per_step']/4, d['value'])"; }
for i in 1 2; do run c2 "ORBX_BLUR_WAVES=4"; run c2 "ORBX_BLUR_WAVES=8"; run c2 "ORBX_BLUR_WAVES=8 ORBX_BLUR_SEG=8"; done 2>&1 | tee $o/ab.txt
