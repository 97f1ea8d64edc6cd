#!/bin/bash
# PMC passes for the kNN-2 scan kernel (tools/time_matcher.py: 1 / 8 / 64 / 512 pairs of 4000 x 4000).  Own runs with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, counters
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/kpmc_$1 -- python3 tools/time_matcher.py > gpurun_out/kpmc_$1.log 2>&1
  f=$(find gpurun_out/kpmc_$1 -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0]
    if 'knn2_scan' not in k: continue
    g=int(r['Grid_Size']) if 'Grid_Size' in r else 0
    agg[(k,g)][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,g,r['Counter_Name'])]+=1
for (k,g),v in sorted(agg.items(), key=lambda x:x[0][1]):
    print(k, 'grid', g, {c: f"{x/n[(k,g,c)]:.4g}" for c,x in v.items()})
PY
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"
run b "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"
