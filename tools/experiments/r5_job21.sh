#!/bin/bash
# round 5, GPU job 21: the large-quota batch fix: new test, the batch sweep again (other seed too), quadtree tests
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job21; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "large_level_quota or quadtree or level_quota_beyond" > $o/pytest.txt 2>&1; echo "pytest exit=$?"; tail -3 $o/pytest.txt
STRESS_BATCH=20 timeout -k 10 800 python3 tests/tools/stress_parity.py 80 21 > $o/stress_batch_21.log 2>&1; echo "sweep 21 exit=$?"; grep -E "MISMATCH|exception" $o/stress_batch_21.log | cut -c1-200 | head; tail -1 $o/stress_batch_21.log
STRESS_BATCH=40 timeout -k 10 800 python3 tests/tools/stress_parity.py 60 22 > $o/stress_batch_22.log 2>&1; echo "sweep 22 exit=$?"; grep -E "MISMATCH|exception" $o/stress_batch_22.log | cut -c1-200 | head; tail -1 $o/stress_batch_22.log
