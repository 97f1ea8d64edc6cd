#!/bin/bash
# A/B of two builds on one box with all stage times: liborbx_prev.so against liborbx.so; optional quick parity first
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$1; mkdir -p $o
if [ -n "$2" ]; then
  timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "$2" > $o/pytest.txt 2>&1; rc=$?; echo "pytest exit=$rc"; tail -3 $o/pytest.txt
  [ $rc -ne 0 ] && exit 1
fi
run() { ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$1 python3 bench.py --config ${2:-c2} --batches-per-step 2 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ${2:-c2}', d['stages_ms'], d['ms_per_step']/2, d['value'])"; }
for i in 1 2 3; do run liborbx_prev.so; run liborbx.so; done 2>&1 | tee $o/ab.txt
for c in c4 c5 c3; do run liborbx_prev.so $c; run liborbx.so $c; done 2>&1 | tee -a $o/ab.txt
