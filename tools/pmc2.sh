#!/bin/bash
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ORBX_SUBBATCHES=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --batches-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > gpurun_out/pmc2.log 2>&1
f=$(find gpurun_out/pmc2 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if k.startswith('k_'): print(k, {c: f"{x/3:.4g}" for c,x in v.items()})
PY
