#!/usr/bin/env python3
"""Timing of the brute-force Hamming matcher (BASELINE config 5: 4000 x 4000 kNN-2): one pair and batches of pairs,
device-resident, HIP-event timed on the matcher's stream via torch events on an external stream wrapper.
Usage: python tools/time_matcher.py [nq] [nt]"""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch

pkg = g.load_package()
lib = pkg.load_library()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rng = np.random.default_rng(1)
out = {}
for P in (1, 8, 64, 512):
    q = torch.from_numpy(rng.integers(0, 256, (P, nq, 32), dtype=np.uint8)).cuda()
    t = torch.from_numpy(rng.integers(0, 256, (P, nt, 32), dtype=np.uint8)).cuda()
    idx = torch.empty((P, nq, 2), dtype=torch.int32, device="cuda")
    dist = torch.empty((P, nq, 2), dtype=torch.int32, device="cuda")
    m = pkg.ORBmatcher(nq, nt)
    def call():
        rc = lib.orbx_match_knn2_batch_device(m._h, P, ctypes.c_void_p(q.data_ptr()), nq * 32, None, None, nq, ctypes.c_void_p(t.data_ptr()),
                                              nt * 32, None, None, nt, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()))
        assert rc == 0, rc
    torch.cuda.synchronize()
    for _ in range(3):
        call()
    lib.orbx_matcher_sync(m._h)
    reps = 200 if P <= 8 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    lib.orbx_matcher_sync(m._h)
    dt = (time.perf_counter() - t0) / reps
    pairs = P * nq * nt
    # 8 v_xor + 8 v_bcnt + key + min + med3 = 19 VALU lane-ops per descriptor pair
    out[f"P{P}"] = {"ms_per_call": round(dt * 1e3, 4), "pairs_per_s": round(P / dt, 1), "G_descriptor_pairs_per_s": round(pairs / dt / 1e9, 2),
                    "T_lane_ops_per_s": round(pairs * 19 / dt / 1e12, 2)}
    m.close()
print(json.dumps({"workload": f"kNN-2 {nq} x {nt}, 256-bit Hamming", "results": out}))
