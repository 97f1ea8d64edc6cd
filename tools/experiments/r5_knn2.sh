#!/bin/bash
# round 5: k_knn2_scan_mfma variants - ORBX_KNN_PIPELINE 0 (tile by tile), 1 (pipelined, guarded), 2 (pipelined, unguarded keys interleaved with the products)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5knn2; mkdir -p $o
for p in 2 1 0; do
  echo "== ORBX_KNN_PIPELINE=$p"
  ORBX_KNN_PIPELINE=$p timeout -k 10 300 python -m pytest tests/test_matcher_gpu.py -x -q -m gpu -k "knn2 and not subprocess" 2>&1 | tail -1
  ORBX_KNN_PIPELINE=$p timeout -k 10 200 python3 tools/time_matcher.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600
  ORBX_KNN_PIPELINE=$p timeout -k 10 300 python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['matcher']['ms_per_call'], d['matcher']['frac_of_mfma_peak'], d['matcher']['frac_of_16_cycle_datasheet_peak'])"
done 2>&1 | tee $o/knn_pipeline.txt
