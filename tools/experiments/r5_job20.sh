#!/bin/bash
# round 5, GPU job 20: the randomised parity sweep on BATCH handles (360-px tiles, margin bands, capped group queue)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job20; mkdir -p $o
STRESS_BATCH=20 timeout -k 10 1000 python3 tests/tools/stress_parity.py ${1:-80} 21 > $o/stress_batch.log 2>&1; echo "exit=$?"; grep -c " ok$" $o/stress_batch.log; grep -E "MISMATCH|exception|Traceback" $o/stress_batch.log | head; tail -2 $o/stress_batch.log
