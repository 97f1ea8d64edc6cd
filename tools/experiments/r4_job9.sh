#!/bin/bash
# round 4, GPU job 9: pre-test walk with induction variables + kNN-2 scan with the query-wide insert bound: parity, stage times, matcher timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4job9; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_matcher_gpu.py tests/test_golden.py tests/test_smoke_gpu.py -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; tail -3 $o/pytest.log
[ $rc -ne 0 ] && exit 1
for i in 1 2; do
python3 bench.py --batches-per-step 1 --no-cpu-baseline --no-pcie 2>$o/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'], d['stages_ms'])"
done
python3 bench.py --config c5 --batches-per-step 1 --no-cpu-baseline --no-pcie 2>>$o/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['ms_per_step'], d['stages_ms'], d['matcher'])"
python3 tools/time_matcher.py 2>/dev/null | tail -1
