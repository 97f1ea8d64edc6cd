"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/collect_traffic.sh) per kernel and per batch.
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by rocprofv3's derived counters (value * 1024 = bytes)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
batches = 3  # bench.py --steps 2 --warmup 1
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(float)
    calls = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].split("(")[0]
            agg[k] += float(r["Counter_Value"])
            calls[k] += 1
    for k in agg:
        out.setdefault(k, {})[c + "_bytes_per_batch"] = agg[k] * 1024.0 / batches
        out[k]["dispatches_per_batch"] = calls[k] / batches
print(json.dumps(out, indent=1))
