#!/bin/bash
# k_pyr_fast per pyramid level: duration of each of the 8 launches of a c2 batch (rocprofv3 kernel trace), ns per source pixel, next to the
# level's LDS per workgroup / workgroups per CU (ORBX_DEBUG).  Usage on the GPU box: bash tools/experiments/pf_per_level.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pf_per_level; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python3 bench.py --batches-per-step 1 --steps 4 --warmup 1 --no-cpu-baseline --no-pcie > $o/bench.json 2> $o/err.txt
f=$(find $o/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_pyr_fast" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sizes = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
nb = len(rows) // 8
for l in range(8):
    d = [int(rows[b * 8 + l]["End_Timestamp"]) - int(rows[b * 8 + l]["Start_Timestamp"]) for b in range(1, nb)]
    w, h = sizes[l]
    px = w * h * 1024
    print(f"level {l} {w}x{h}: {sum(d) / len(d) / 1e3:8.1f} us   {sum(d) / len(d) / px * 1e3:.4f} ps per source pixel   grid {rows[l]['Grid_Size_X'] if 'Grid_Size_X' in rows[l] else ''} lds {rows[l].get('LDS_Block_Size', '')}")
PY
