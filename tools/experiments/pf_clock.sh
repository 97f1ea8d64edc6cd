#!/bin/bash
# The clock the chip holds under the extraction kernels: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, median per kernel and grid size
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pf_clock; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $o/t -- python3 bench.py --batches-per-step 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $o/log.txt 2>&1
python3 - $(find $o/t -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    g = (k, int(r["Grid_Size"]))
    agg[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[g]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for g, v in sorted(agg.items()):
    med = lambda a: sorted(a)[len(a) // 2]
    ns = med(v["ns"]); gui = med(v["GRBM_GUI_ACTIVE"])
    if ns < 20000: continue
    print(f"{g[0][:40]:40s} grid {g[1]:9d}: {ns / 1e3:8.1f} us, clock {gui / 8.0 / ns:.2f} GHz, {med(v['SQ_INSTS_VALU']) / 1e6:.1f} M vector instr")
PY
