#!/bin/bash
export ORBX_STRICT_ENV=1
export ORBX_SUBBATCHES=1 ORBX_NO_BLUR=1
for ph in 15 31 11 9 1 0 14; do
  echo -n "phases=$ph "; ORBX_PF_PHASES=$ph timeout -k 10 200 python bench.py --batches-per-step 1 --steps 5 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms']['pyramid+FAST'])"
done
