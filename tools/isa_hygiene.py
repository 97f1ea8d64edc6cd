#!/usr/bin/env python3
"""Per-kernel counts of what a latency- or issue-bound kernel should not contain, over the device assembly of the build
(wut-cuda-orb-slam3_amd/csrc/build/*.fix.s): flat accesses (a generic pointer: counts against both wait counters), scratch accesses (spills),
per-lane 64-bit arithmetic (v_*_u64 / b64: usually a signed index on a uniform base that could be base + unsigned 32-bit offset) and full
32-bit multiplies (v_mul_lo_u32 / v_mul_hi_u32: ~8 issue cycles; v_mul_u32_u24 does for operands below 2^24).
Usage: python3 tools/isa_hygiene.py [min_instructions]      (round 5: DESIGN 6 "Quadtree kernels without generic pointers", k_blur3)"""
import glob, os, re, sys

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wut-cuda-orb-slam3_amd", "csrc", "build")
min_n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for fn in sorted(glob.glob(os.path.join(root, "*.fix.s"))):
    cur, stats = None, {}
    for ln in open(fn):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            cur = m.group(1)
            stats[cur] = dict(n=0, flat=0, scratch=0, u64=0, mul32=0)
            continue
        if cur is None:
            continue
        if ln.startswith("\t.end_amdhsa_kernel"):
            cur = None
            continue
        t = ln.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        st, op = stats[cur], t[0]
        st["n"] += 1
        st["flat"] += op.startswith("flat_")
        st["scratch"] += op.startswith("scratch_")
        st["u64"] += op.startswith("v_") and any(k in op for k in ("u64", "b64", "i64"))
        st["mul32"] += op.startswith(("v_mul_lo_u32", "v_mul_hi_u32 "))
    print("==", os.path.basename(fn))
    for k, v in stats.items():
        if v["n"] >= min_n:
            print(f"{k[:64]:64s} n={v['n']:6d} flat={v['flat']:4d} scratch={v['scratch']:4d} v64={v['u64']:4d} mul32={v['mul32']:3d}")
