#!/bin/bash
# round 5, GPU job 37: quadtree kernels without generic pointers: the parity file, A/B on the four workloads, single-frame latency
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job37; mkdir -p $o
timeout -k 10 700 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $o/pytest.txt 2>&1; rc=$?; echo "pytest exit=$rc"; tail -3 $o/pytest.txt
[ $rc -ne 0 ] && exit 1
run() { ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$1 python3 bench.py --config ${2:-c2} --batches-per-step 2 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ${2:-c2}', d['stages_ms'], d['ms_per_step']/2, d['value'])"; }
for c in c2 c5 c4 c3; do for i in 1 2; do run liborbx_prev.so $c; run liborbx.so $c; done; done 2>&1 | tee $o/ab.txt
for l in liborbx_prev.so liborbx.so; do echo "== $l"; ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$l python3 tools/time_single_frame.py 2>/dev/null | tail -4; done 2>&1 | tee $o/single.txt
