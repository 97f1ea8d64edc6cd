#!/bin/bash
# finer ablation: 15 = all; +16 skip scoring; +32 skip queue push (and therefore scoring/NMS work)
export ORBX_STRICT_ENV=1
for ph in 15 31 47 3 35 1 0; do
  echo -n "phases=$ph "; ORBX_PF_PHASES=$ph timeout -k 10 200 python bench.py --batches-per-step 1 --steps 5 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms']['pyramid+FAST'])"
done
