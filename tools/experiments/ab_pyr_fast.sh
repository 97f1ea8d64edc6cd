#!/bin/bash
# A/B of two builds of the library on ONE box: the pyramid+FAST stage time and the step time of the c2 bench, alternating, N rounds.
# Usage on the GPU box: bash tools/experiments/ab_pyr_fast.sh liborbx_base.so liborbx.so [rounds]
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
a=$1; b=$2; n=${3:-3}
run() { ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$1 python3 bench.py --batches-per-step 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['stages_ms']['pyramid+FAST'], d['ms_per_step'], d['value'])"; }
for i in $(seq $n); do run $a; run $b; done
