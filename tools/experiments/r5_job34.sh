#!/bin/bash
# round 5, GPU job 34: train-row splits of the kNN-2 scan on c5's 128-pair call (2048 workgroups = 2.67 rounds of 768 resident ones: the tail)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job34; mkdir -p $o
run() { env $1 python3 bench.py --config c5 --batches-per-step 2 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('matcher',{}); print('$1', m.get('ms_per_call'), m.get('frac_of_mfma_peak'), d['ms_per_step']/2, d['value'])"; }
for i in 1 2; do for s in 0 2 3 4 6; do if [ $s = 0 ]; then run ORBX_STRICT_ENV=1; else run ORBX_KNN_SPLIT=$s; fi; done; done 2>&1 | tee $o/ab.txt
