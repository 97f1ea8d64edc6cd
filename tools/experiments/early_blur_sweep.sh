#!/bin/bash
# Can the blur (matrix cores + HBM) run INSIDE the pyramid+FAST chain (vector-issue bound) if k_pyr_fast leaves it a workgroup's worth of
# LDS and wave slots?  ORBX_EARLY_BLUR starts the blur of level l as soon as level l exists; ORBX_PF_LDS_PAD pads k_pyr_fast's LDS so that
# three instead of four of its workgroups fit a CU (4 x 8 waves fill the CU's 32 wave slots, so the blur cannot enter otherwise).
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --batches-per-step 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'])"; }
for eb in 0 1; do for pad in 0 6000 11000 14000; do
  echo -n "early_blur $eb pf_lds_pad $pad: "; ORBX_EARLY_BLUR=$eb ORBX_PF_LDS_PAD=$pad run
done; done
