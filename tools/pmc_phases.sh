#!/bin/bash
# Dynamic instruction profile of k_pyr_fast by phase: the kernel returns at stop point k (bits 4..7 of ORBX_PF_PHASES), so
# the differences between consecutive runs are the VALU / LDS / SALU instructions and the time each phase adds.
# Usage on the GPU box: bash tools/pmc_phases.sh > gpurun_out/pmc_phases.txt
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 1 2 3 4 5 6 7 0; do  # (stop points: see PyrFastArgs::phases)
export ORBX_PF_PHASES=$((15 + 16 * k))
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmcph_$k -- python3 bench.py --batches-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > gpurun_out/pmcph_$k.log 2>&1
f=$(find gpurun_out/pmcph_$k -name "*counter_collection.csv" | head -1)
python3 - "$f" $k <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_pyr_fast' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
print("stop", sys.argv[2], {c: f"{x/3:.4g}" for c,x in sorted(agg.items())})
PY
done
