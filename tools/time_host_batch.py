#!/usr/bin/env python3
"""Where the time of the host-pointer batch call (orbx_extract_batch, SURVEY 8(d)'s metric) goes: raw link rates of this box
(pageable / page-locked, both directions), then the call itself from pageable and from page-locked arrays.
Usage on the GPU box: python tools/time_host_batch.py [--config c2] > gpurun_out/host_batch.txt"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    cfg = bench.CONFIGS[args.config]
    W, H, B = cfg["w"], cfg["h"], cfg["units"]
    left, _ = bench.make_frames(pkg, cfg, 0, B)
    nbytes = left.nbytes
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    pin = torch.from_numpy(left.reshape(-1)).pin_memory()
    page = torch.from_numpy(left.reshape(-1).copy())

    def rate(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return nbytes * n / (time.perf_counter() - t0) / 1e9

    print(f"link, {nbytes / 1e6:.1f} MB per copy: H2D pageable {rate(lambda: dev.copy_(page)):.1f} GB/s, H2D page-locked "
          f"{rate(lambda: dev.copy_(pin, non_blocking=True)):.1f} GB/s, D2H pageable {rate(lambda: page.copy_(dev)):.1f} GB/s, "
          f"D2H page-locked {rate(lambda: pin.copy_(dev, non_blocking=True)):.1f} GB/s")
    ex = pkg.ORBextractor(cfg["nf"], 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    pinned_in = pkg.host_array(left.shape, np.uint8)
    pinned_in[...] = left
    out_page, out_lock = ex.output_arrays(B), ex.output_arrays(B, page_locked=True)
    with pkg.host_registered(left) as reg:  # the caller's own array, page-locked in place (orbx_host_register)
        ex.extract_batch(reg, cfg["lap"], out=out_page)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ex.extract_batch(reg, cfg["lap"], out=out_page)
        dt = (time.perf_counter() - t0) / args.reps
        print(f"orbx_extract_batch registered in, pageable out: {dt * 1e3:.2f} ms per {B} frames = {B / dt:.0f} frames/s, input {nbytes / dt / 1e9:.1f} GB/s")
    for tag, arr, out in (("pageable in / out", left, out_page), ("page-locked in, pageable out", pinned_in, out_page),
                          ("page-locked in / out", pinned_in, out_lock), ("pageable in, fresh result arrays per call", left, None)):
        ex.extract_batch(arr, cfg["lap"], out=out)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ex.extract_batch(arr, cfg["lap"], out=out)
        dt = (time.perf_counter() - t0) / args.reps
        print(f"orbx_extract_batch {tag}: {dt * 1e3:.2f} ms per {B} frames = {B / dt:.0f} frames/s, input {nbytes / dt / 1e9:.1f} GB/s", flush=True)
    # device-resident for comparison
    d = torch.from_numpy(left).cuda()
    ex.extract_batch_device(d.data_ptr(), B, W, H, W, W * H, cfg["lap"]); ex.sync()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ex.extract_batch_device(d.data_ptr(), B, W, H, W, W * H, cfg["lap"])
    ex.sync()
    dt = (time.perf_counter() - t0) / args.reps
    print(f"device-resident: {dt * 1e3:.2f} ms per {B} frames")


if __name__ == "__main__":
    main()
