#!/bin/bash
# phase ablation of k_pyr_fast (timing only; outputs are wrong when phases are disabled)
export ORBX_STRICT_ENV=1
for ph in 0 1 2 3 6 7 15; do
  echo -n "phases=$ph "; ORBX_PF_PHASES=$ph timeout -k 10 200 python bench.py --batches-per-step 1 --steps 5 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms'])"
done
