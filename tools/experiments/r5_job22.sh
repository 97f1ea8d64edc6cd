#!/bin/bash
# round 5, GPU job 22: the batch sweep with more frames per case (single-wave quadtree instances: > 1024 per launch group)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job22; mkdir -p $o
STRESS_BATCH=${1:-160} timeout -k 10 1000 python3 tests/tools/stress_parity.py ${2:-50} 23 > $o/stress_batch.log 2>&1; echo "sweep exit=$?"; grep -E "MISMATCH|exception" $o/stress_batch.log | cut -c1-300 | head; tail -1 $o/stress_batch.log; grep -c " ok$" $o/stress_batch.log
