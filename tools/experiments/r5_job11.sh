#!/bin/bash
# round 5, GPU job 11: the new overflow / layout tests, then the whole GPU suite
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job11; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "group_queue_overflow or classic_tile_layout or wave_per_cell" > $o/new.txt 2>&1; rc=$?; echo "new tests exit=$rc"; tail -15 $o/new.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $o/all.txt 2>&1; rc=$?; echo "all exit=$rc"; tail -5 $o/all.txt
