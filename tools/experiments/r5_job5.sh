#!/bin/bash
# round 5, GPU job 5: the successor's L2 warm-up (ORBX_PF_AHEAD) - parity, then A/B against 0 (off), and the phase stamps
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job5; mkdir -p $o
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "batch or full_size or stagewise" > $o/pytest.txt 2>&1; echo "pytest exit=$?"; tail -3 $o/pytest.txt
run() { ORBX_PF_AHEAD=$1 python3 bench.py --batches-per-step 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead=$1', d['stages_ms']['pyramid+FAST'], d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do run 0; run -1; run 1536; run 384; done 2>&1 | tee $o/ab.txt
