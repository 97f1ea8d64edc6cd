#!/bin/bash
export ORBX_STRICT_ENV=1
export ORBX_NO_BLUR=1
for st in 0 4 3 2 1; do
  echo -n "qt_stop=$st "; ORBX_QT_STOP=$st timeout -k 10 200 python bench.py --batches-per-step 1 --steps 5 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms'])"
done
