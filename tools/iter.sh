#!/bin/bash
# Quick iteration on the GPU box: extractor parity tests, then the bench line without the CPU baseline.
# Usage: bash tools/iter.sh <tag> [extra bench args]
tag=${1:-it}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/$tag/pytest.log 2>&1
rc=$?
tail -4 gpurun_out/$tag/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-pcie "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -5 gpurun_out/$tag/bench.err; exit 1; }
python3 - gpurun_out/$tag/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("fps", d["value"], "ms/step", d["ms_per_step"], "stages", d["stages_ms"])
PY
