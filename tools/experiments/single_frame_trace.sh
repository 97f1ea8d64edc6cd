#!/bin/bash
# Kernel timeline of ONE single-frame extraction (device-resident): rocprofv3 --kernel-trace, last call of a short run.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/sf.py <<'PY'
import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
pkg = importlib.import_module("wut-cuda-orb-slam3_amd")
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
d = torch.from_numpy(pkg.synth_frame(0, 640, 480)[None]).cuda()
for _ in range(20):
    ex.extract_batch_device(d.data_ptr(), 1, 640, 480, 640, 640 * 480); ex.sync()
PY
rm -rf gpurun_out/sftrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sftrace -- python3 /tmp/sf.py > gpurun_out/sftrace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/sftrace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call = last group starting with k_pyr_fast after an orient kernel
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_orient")]
start = idx[-2] + 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start: idx[-1] + 1]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f} us  {r["Kernel_Name"][:40]}  grid {r.get("Grid_Size", "")}')
PY
