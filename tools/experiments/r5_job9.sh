#!/bin/bash
# round 5, GPU job 9: margin tiles + capped group queue (four workgroups per CU with wide tiles): parity, geometry, A/B
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job9; mkdir -p $o
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $o/pytest.txt 2>&1; rc=$?; echo "pytest exit=$rc"; tail -5 $o/pytest.txt
[ $rc -ne 0 ] && exit 1
ORBX_DEBUG=1 python3 tools/experiments/print_geometry.py 640 480 1000 1024 2>&1 | grep "LDS layout\|blocks/CU" > $o/geom.txt; cat $o/geom.txt
run() { env $1 python3 bench.py --batches-per-step 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['stages_ms']['pyramid+FAST'], d['ms_per_step'], d['value'])"; }
for i in 1 2; do
run ORBX_PF_LDS_FIT=1
run ORBX_PF_GQCAP=4000
done 2>&1 | tee $o/ab.txt
