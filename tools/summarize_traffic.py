"""Summarises the rocprofv3 PMC passes of tools/collect_traffic.sh per kernel and per batch of the bench command
(bench.py --batches-per-step 1 --steps 2 --warmup 1 = 3 batches).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in units of 1024 B.
Writes <dir>/summary.json (every kernel) and <dir>/pyr_fast_counters.json (the flat record of the north-star kernel that
bench.py reads from profiles/pyr_fast_counters_latest.json)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
batches = 3


def norm(name):  # "void k_pyr_fast<false>(PyrFastArgs)" -> "k_pyr_fast"
    name = name.split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return name.split("<")[0]


out = {}
for d, counters, scale in (("FETCH_SIZE", ["FETCH_SIZE"], 1024.0), ("WRITE_SIZE", ["WRITE_SIZE"], 1024.0),
                           ("SQ", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], 1.0)):
    files = glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c not in counters:
                continue
            k = norm(r["Kernel_Name"])
            agg[k][c] += float(r["Counter_Value"])
            if c == counters[0]:
                calls[k] += 1
    for k in agg:
        for c in agg[k]:
            name = c + ("_bytes_per_batch" if scale != 1.0 else "_per_batch")
            out.setdefault(k, {})[name] = agg[k][c] * scale / batches
        out[k]["dispatches_per_batch"] = calls[k] / batches
    # durations of the SQ pass (for the effective clock: GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    if d == "SQ":
        for f in glob.glob(os.path.join(root, d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = norm(r["Kernel_Name"])
                out.setdefault(k, {})["ns_per_batch_profiled"] = out.get(k, {}).get("ns_per_batch_profiled", 0.0) + \
                    (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / batches
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
pf = dict(out.get("k_pyr_fast", {}))
if "GRBM_GUI_ACTIVE_per_batch" in pf and pf.get("ns_per_batch_profiled"):
    pf["effective_clock_ghz_profiled"] = round(pf["GRBM_GUI_ACTIVE_per_batch"] / 8.0 / pf["ns_per_batch_profiled"], 3)
# cost of the kernel's instruction mix in cycles per wave-instruction on one SIMD: the measured issue costs of the two opcode classes
# (profiles/r04_issue_probe_v4_classes_hysteresis.txt: full rate 2.1 cycles, slow class 4.1) weighted with the kernel's static mix
# (tools/sq_slowmode.py annotate: 47 % slow-class) = 3.0
pf["valu_cycles_per_instr_mix"] = 3.0
pf["kernel"] = "k_pyr_fast"
# provenance: bench.py drops these numbers when the kernel sources have changed since (bench.kernel_source_sha256)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (no torch import at module level)
pf["kernel_source_sha256"] = bench.kernel_source_sha256()
# .git does not travel to the GPU box: tools/stamp_commit.sh writes HEAD (+ "-dirty") into .git_commit before the gpurun call
commit = os.environ.get("ORBX_COMMIT", "")
if not commit:
    try:
        commit = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".git_commit")).read().strip()
    except OSError:
        commit = ""
if not commit:
    raise SystemExit("summarize_traffic.py: no commit to stamp - run tools/stamp_commit.sh before the gpurun call (or set ORBX_COMMIT)")
pf["commit"] = commit
pf["bench"] = "bench.py --config c2 --batches-per-step 1 --steps 2 --warmup 1 (3 batches of 1024 frames)"
json.dump(pf, open(os.path.join(root, "pyr_fast_counters.json"), "w"), indent=1)
print(json.dumps(pf, indent=1))
