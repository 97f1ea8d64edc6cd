#!/bin/bash
# PMC passes for the blur kernel (kernels run one at a time under counter collection, so these are the numbers of k_blur3 alone).
# Own runs with --kernel-trace only, as the pool requires.  Prints per-launch-set sums (8 level launches) for kernels named k_blur*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, counters
  rm -rf gpurun_out/bpmc_$1
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/bpmc_$1 -- python3 bench.py --batches-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > gpurun_out/bpmc_$1.log 2>&1
  rc=$?
  f=$(find gpurun_out/bpmc_$1 -name "*counter_collection.csv" 2>/dev/null | sort | sed -n 1p)
  if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "pass $1 ($2): rocprofv3 exit $rc, no counter file - see gpurun_out/bpmc_$1.log"; return 1; fi
  python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    n[(k, r['Counter_Name'])]+=1
for k,v in agg.items():
    if 'blur' in k: print(k, {c: f"{x/(n[(k,c)]/8):.4g}" for c,x in v.items()}, 'launches', {n[(k,c)] for c in v})
PY
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
run b "SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS"
# (the TCC block has 4 counter slots per pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots": five in one pass made rocprofv3 abort
# with "Request exceeds the capabilities of the hardware")
run c "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run c2 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run d "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"
run e "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
