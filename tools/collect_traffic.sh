#!/bin/bash
# Per-kernel PMC passes of the bench command (MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots): FETCH_SIZE and WRITE_SIZE in
# separate runs (they do not fit one TCC pass; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950), and one SQ
# run with the VALU instruction count (for the VALU-issue roofline of k_pyr_fast) and GRBM_GUI_ACTIVE (effective clock).
# Each run is --kernel-trace + --pmc only.  tools/summarize_traffic.py turns the CSVs into the two JSON files that
# bench.py reads (profiles/hbm_traffic_latest.json, profiles/pyr_fast_counters_latest.json).
# Usage on the GPU box: bash tools/collect_traffic.sh [bench args, e.g. --config c3]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE gpurun_out/traffic/SQ
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic/$c -- python3 bench.py --batches-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie "$@" > gpurun_out/traffic/$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/traffic/SQ -- python3 bench.py --batches-per-step 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie "$@" > gpurun_out/traffic/SQ.log 2>&1
python3 tools/summarize_traffic.py gpurun_out/traffic
cat gpurun_out/traffic/pyr_fast_counters.json
