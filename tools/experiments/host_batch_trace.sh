#!/bin/bash
# Timeline of ONE orbx_extract_batch call (host frames in, results out): memory copies and kernels with start / end relative to the call's
# first event.  rocprofv3 kernel + memory-copy trace (no counters).  Usage on the GPU box: bash tools/experiments/host_batch_trace.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/hbtrace; rm -rf $o; mkdir -p $o
cat > $o/run.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import __graft_entry__ as graft, bench
pkg = graft.load_package()
cfg = bench.CONFIGS["c2"]
left, _ = bench.make_frames(pkg, cfg, 0, 1024)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1024)
for i in range(3):
    t0 = time.perf_counter(); ex.extract_batch(left, cfg["lap"]); print("call", i, (time.perf_counter() - t0) * 1e3, "ms")
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/prof -- python3 $o/run.py > $o/run.log 2>&1
cat $o/run.log | grep call
python3 - $o <<'PY'
import csv, glob, sys, os
o = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(o, "prof", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:40], r.get("Stream_Id", "")))
for f in glob.glob(os.path.join(o, "prof", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), r.get("Stream_Id", "")))
ev.sort()
# the last call = events after the last gap of > 2 ms before a big H2D
big = [i for i, e in enumerate(ev) if e[2].startswith("C") and "HOST_TO_DEVICE" in e[2].upper() and int(e[2].split()[-1]) > 10_000_000]
start = big[-8] if len(big) >= 8 else (big[-4] if len(big) >= 4 else 0)
# walk back to the first large copy of the last call: calls have the same number of big copies
n_per = len(big) // 3 if len(big) >= 3 else len(big)
start = big[-n_per] if n_per else 0
t0 = ev[start][0]
last = None
for s, e, name, st in ev[start:]:
    if name.startswith("K") and last and last[2] == name and last[3] == st and s - last[1] < 20000:
        last = (last[0], e, name, st, last[4] + 1); continue
    if last: print(f"{(last[0]-t0)/1e6:8.3f} - {(last[1]-t0)/1e6:8.3f} ms  stream {last[3]:>3}  {last[2]} x{last[4]}")
    last = (s, e, name, st, 1)
if last: print(f"{(last[0]-t0)/1e6:8.3f} - {(last[1]-t0)/1e6:8.3f} ms  stream {last[3]:>3}  {last[2]} x{last[4]}")
PY
