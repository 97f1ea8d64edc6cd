#!/bin/bash
# The clock the chip holds under the kNN-2 matrix-core scan: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, per launch of
# tools/time_matcher.py (1 / 8 / 64 / 512 pairs of 4000 x 4000), and the matrix-core busy cycles of the same launches.
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/knn_clock; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $o/t -- python3 tools/time_matcher.py > $o/log.txt 2>&1
python3 - $(find $o/t -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "knn2_scan" not in r["Kernel_Name"]: continue
    g = int(r["Grid_Size"])
    agg[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[g]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for g, v in sorted(agg.items()):
    med = lambda a: sorted(a)[len(a) // 2]
    ns = med(v["ns"]); gui = med(v["GRBM_GUI_ACTIVE"]); mf = med(v["SQ_INSTS_MFMA"]); busy = med(v["SQ_VALU_MFMA_BUSY_CYCLES"])
    clk = gui / 8.0 / ns
    print(f"grid {g}: {ns / 1e3:8.1f} us, clock {clk:.2f} GHz, {mf:.3g} MFMA wave-instr, {med(v['SQ_INSTS_VALU']) / mf:.2f} vector instr per MFMA, "
          f"matrix pipe busy {busy / 1024 / (ns * clk):.2f} of the kernel's cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES), "
          f"22-cycle issue floor {mf * 22 / 1024 / (ns * clk):.2f}")
PY
