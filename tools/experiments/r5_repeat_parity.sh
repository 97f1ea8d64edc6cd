#!/bin/bash
# the parity file N times in fresh processes (a host-side crash seen once in test_two_threads_recapturing_all_the_time on a discarded build)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$1; mkdir -p $o
for i in $(seq 1 ${2:-3}); do
  timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $o/run_$i.txt 2>&1; echo "run $i exit=$? $(tail -1 $o/run_$i.txt)"
done
