#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc3_$1 -- python3 bench.py --batches-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > gpurun_out/pmc3_$1.log 2>&1
f=$(find gpurun_out/pmc3_$1 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'k_' in k: print(k, {c: f"{x/3:.4g}" for c,x in v.items()})
PY
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
run b "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"
run c "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_GDS"
