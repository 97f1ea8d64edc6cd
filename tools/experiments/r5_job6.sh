#!/bin/bash
# round 5, GPU job 6: scheduling switches re-measured against the round-5 kernels (sub-batches of a device batch: the back end of
# sub-batch k under the front of k + 1)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job6; mkdir -p $o
run() { env $1 python3 bench.py --batches-per-step 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['stages_ms'], d['ms_per_step'], d['value'])"; }
for i in 1 2; do
run ORBX_SUBBATCHES=1
run ORBX_SUBBATCHES=2
run ORBX_SUBBATCHES=3
run ORBX_SUBBATCHES=4
done 2>&1 | tee $o/ab.txt
