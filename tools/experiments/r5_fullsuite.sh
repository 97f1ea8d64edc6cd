#!/bin/bash
# the whole GPU suite + smoke, log under gpurun_out/$1
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$1; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $o/all.txt 2>&1; rc=$?; echo "all exit=$rc"; tail -5 $o/all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $o/smoke.txt 2>&1; echo "smoke exit=$?"; tail -2 $o/smoke.txt
