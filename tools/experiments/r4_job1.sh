#!/bin/bash
# round 4, GPU job 1: issue probes, ring-rate bodies (round-3 formulation vs explicit v_bitop3), parity + bench of the new build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4job1; mkdir -p $o
bash tools/experiments/run_issue_probe.sh > $o/issue_probe.txt 2>&1; echo "issue_probe exit=$?"
cat $o/issue_probe.txt
for b in ring_rate_r3 ring_rate; do
  timeout -k 10 60 ./tools/experiments/$b > $o/$b.txt 2>&1; cat $o/$b.txt
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $o/prof_$b -- ./tools/experiments/$b > $o/prof_$b.txt 2>&1
done
python3 - $o <<'PY'
import csv, glob, sys, os, collections
o = sys.argv[1]
for b in ("ring_rate_r3", "ring_rate"):
    dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(o, "prof_" + b, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)): dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for f in glob.glob(os.path.join(o, "prof_" + b, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)): cnt[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in dur:
        d = dur[k][-1]; c = {n: v[-1] for n, v in cnt[k].items()}
        if "SQ_INSTS_VALU" not in c: continue
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / d if d else 0
        print(f"{b:13s} {k:18s} {d/1e3:9.1f} us  VALU {c['SQ_INSTS_VALU']:.4g}  SALU {c.get('SQ_INSTS_SALU',0):.3g}  LDS {c.get('SQ_INSTS_LDS',0):.3g}  clock {clk:.2f} GHz  "
              f"cycles per VALU per SIMD {d * clk * 1024 / c['SQ_INSTS_VALU']:.2f}  (at 2.4 GHz: {d * 2.4 * 1024 / c['SQ_INSTS_VALU']:.2f})")
PY
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit=$rc" >> $o/pytest_gpu.log; tail -3 $o/pytest_gpu.log
[ $rc -eq 0 ] && timeout -k 10 400 python bench.py > $o/bench_c2.json 2> $o/bench_c2.err; cat $o/bench_c2.json
