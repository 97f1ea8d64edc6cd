#!/bin/bash
# Practical vector-issue rate of k_pyr_fast's inner bodies (tools/experiments/ring_rate.hip): SQ_INSTS_VALU and kernel durations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/ring_rate; rm -rf $o; mkdir -p $o
./tools/experiments/ring_rate > $o/plain.txt 2>&1; cat $o/plain.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $o/prof -- ./tools/experiments/ring_rate > $o/prof.txt 2>&1
python3 - $o <<'PY'
import csv, glob, sys, os, collections
o = sys.argv[1]
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(o, "prof", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob(os.path.join(o, "prof", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)): cnt[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in dur:
    d = dur[k][-1]; c = {n: v[-1] for n, v in cnt[k].items()}
    if "SQ_INSTS_VALU" not in c: continue
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / d if d else 0
    print(f"{k:18s} {d/1e3:9.1f} us  VALU {c['SQ_INSTS_VALU']:.4g}  SALU {c.get('SQ_INSTS_SALU',0):.3g}  LDS {c.get('SQ_INSTS_LDS',0):.3g}  clock {clk:.2f} GHz  "
          f"cycles per VALU per SIMD {d * clk * 1024 / c['SQ_INSTS_VALU']:.2f}  (at 2.4 GHz: {d * 2.4 * 1024 / c['SQ_INSTS_VALU']:.2f})  active_valu/insts {c.get('SQ_ACTIVE_INST_VALU',0)/c['SQ_INSTS_VALU']:.2f}  conflicts/lds {c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_INSTS_LDS',1),1):.2f}")
PY
