#!/bin/bash
# k_pyr_fast, debug instance (phase mask compiled in): stage time with all phases (271 = 15 + an unused bit, so that the debug instance runs)
# against the mask without count + emission (263), without NMS + emission (259), without the FAST rounds (257): what the tail phases cost
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ph in 271 263 259 257 271; do
  echo -n "ORBX_PF_PHASES=$ph: "
  ORBX_PF_PHASES=$ph python3 bench.py --batches-per-step 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms']['pyramid+FAST'], d['ms_per_step'])"
done
