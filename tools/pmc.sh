#!/bin/bash
# PMC counter passes for the bench (own runs, --kernel-trace only, as the pool requires)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, counters
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc_$1 -- python3 bench.py --batches-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > gpurun_out/pmc_$1.log 2>&1
  f=$(find gpurun_out/pmc_$1 -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    print(k, {c: f"{x:.4g}" for c,x in v.items()})
PY
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
run b "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"
