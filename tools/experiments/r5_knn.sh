#!/bin/bash
# round 5: the kNN-2 scan's insert bound (ORBX_KNN_SHARED_BOUND 0 / 1 / 2): tests, then the matcher alone and the c5 bench line
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5knn; mkdir -p $o
timeout -k 10 500 python -m pytest tests/test_matcher_gpu.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest exit=$?"; tail -3 $o/pytest.txt
for b in "0 0" "2 0" "0 1" "2 1"; do
  set -- $b
  echo "== ORBX_KNN_SHARED_BOUND=$1 ORBX_KNN_PIPELINE=$2"
  export ORBX_KNN_PIPELINE=$2
  ORBX_KNN_SHARED_BOUND=$1 timeout -k 10 200 python3 tools/time_matcher.py 2>&1 | grep -v amdgpu.ids | tail -3
  ORBX_KNN_SHARED_BOUND=$1 timeout -k 10 300 python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['matcher'])"
done 2>&1 | tee $o/knn_bound.txt
