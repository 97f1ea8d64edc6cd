#!/bin/bash
# k_pyr_fast: queue slots of the pre-test by per-lane LDS atomics (ORBX_PF_LANE_ALLOC=1) against ballot prefix counts: parity + stage time
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "ORBX_PF_LANE_ALLOC=$v"
  ORBX_PF_LANE_ALLOC=$v timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "stagewise or full_size or image_content" 2>&1 | tail -1
  ORBX_PF_LANE_ALLOC=$v python3 bench.py --config c2 --no-cpu-baseline --no-pcie --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms'], d['roofline']['avg_duration_ms'])"
done
