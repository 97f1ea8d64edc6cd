#!/usr/bin/env python3
"""bench.py — headline benchmark of the ORB front-end hot path (BASELINE.json metric).

One "step" = one pass of ORBextractor::operator() (pyramid -> FAST/NMS -> quadtree -> IC-angle -> blur -> rBRIEF
-> pack) over ONE batch of 1024 synthetic 640x480 mono frames, nFeatures=1000, 8 levels, scale 1.2
(BASELINE.json configs[1]) per GPU.  Inputs are resident in HBM before the timed region starts; results stay in
HBM (the PCIe-inclusive rate of the host-pointer API is reported in DESIGN.md, never here).

Multi-GPU (--gpus N under torch.distributed.run): frames shard across ranks, one process per GPU, no data-path
collective (weak scaling: every rank runs its own 1024-frame batch).  With --gather the per-batch result
records are all-gathered over RCCL inside the timed region (north_star's "trivial gather"); off by default
because the path itself has no exchange step.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

W, H, NFEAT, NLEVELS, SCALE, INI_TH, MIN_TH = 640, 480, 1000, 8, 1.2, 20, 7
BATCH = 1024
# algorithmic HBM bytes per 640x480 frame of the fused pyramid+FAST stage
# = read sum(P_0..7) + write sum(P_1..7) = 950,532 + 643,332
ALGO_BYTES_PYR_FAST_PER_FRAME = 1_593_864
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(orc, frames: np.ndarray, threads: int):
    """Oracle (CPU restatement of the reference algorithm, kind="port") timed on the host cores on a bounded sample."""
    ref = orc.Oracle(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
    ref.extract_batch(frames[: min(len(frames), threads)], (0, 1000), threads=threads)  # warm-up
    t0 = time.perf_counter()
    counts, _, _, _ = ref.extract_batch(frames, (0, 1000), threads=threads)
    dt = time.perf_counter() - t0
    return len(frames) / dt, dt, int(counts.sum())


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="frames per GPU per step (default: the metric's 1024)")
    ap.add_argument("--gather", action="store_true", help="all-gather result records over RCCL inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=768, help="frames in the CPU-baseline sample")
    ap.add_argument("--pcie", action="store_true", help="also time the host-pointer API (H2D + D2H included); reported separately")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = graft.load_package()
    B = args.batch
    # synthetic frames of this rank's shard: global frame index = rank * B + i
    host = np.empty((B, H, W), np.uint8)
    for i in range(B):
        host[i] = pkg.synth_frame(rank * B + i, W, H)
    frames = torch.from_numpy(host).cuda(local_rank)
    torch.cuda.synchronize()

    ex = pkg.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=W, max_height=H, max_batch=B)
    ex.set_profiling(True)
    d_kps, d_desc, d_n, d_mono = ex.device_outputs()
    cap = ex.capacity

    class _DevView:  # zero-copy torch view of a liborbx device buffer (for the RCCL gather)
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    gather_bufs = None
    if args.gather and world > 1:
        views = [torch.as_tensor(_DevView(d_kps, B * cap * 28), device=f"cuda:{local_rank}"),
                 torch.as_tensor(_DevView(d_desc, B * cap * 32), device=f"cuda:{local_rank}"),
                 torch.as_tensor(_DevView(d_n, B * 4), device=f"cuda:{local_rank}"),
                 torch.as_tensor(_DevView(d_mono, B * 4), device=f"cuda:{local_rank}")]
        gather_bufs = [(v, torch.empty(world * v.numel(), dtype=torch.uint8, device=f"cuda:{local_rank}")) for v in views]

    def step():
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, (0, 1000))
        if gather_bufs is not None:
            ex.sync()  # results of this rank are complete; gather the fixed-capacity records over RCCL / xGMI
            for local, full in gather_bufs:
                dist.all_gather_into_tensor(full, local)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ex.stage_times()  # reset the event ring
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stage_ms, nb = ex.stage_times()

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    n, mono, _, _ = ex.download_outputs(B)
    total_kps = int(n.sum())

    pcie_fps = pcie_pinned_fps = None
    if args.pcie and rank == 0:
        ex.extract_batch(host, (0, 1000))
        t1 = time.perf_counter()
        for _ in range(3):
            ex.extract_batch(host, (0, 1000))
        pcie_fps = 3 * B / (time.perf_counter() - t1)
        # the same call with the frames in page-locked host memory (what a capture pipeline that owns its buffers would pass)
        pinned = torch.from_numpy(host).pin_memory().numpy()
        ex.extract_batch(pinned, (0, 1000))
        t1 = time.perf_counter()
        for _ in range(3):
            ex.extract_batch(pinned, (0, 1000))
        pcie_pinned_fps = 3 * B / (time.perf_counter() - t1)

    if rank == 0:
        frames_total = world * B * args.steps
        fps = frames_total / dt
        pyr_ms = float(stage_ms[0])
        achieved = ALGO_BYTES_PYR_FAST_PER_FRAME * B / (pyr_ms * 1e-3) / 1e9 if pyr_ms > 0 else 0.0
        # HBM bytes of the same launch set from the committed PMC passes (tools/collect_traffic.sh: FETCH_SIZE and
        # WRITE_SIZE in separate rocprofv3 runs; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
        if os.path.exists(tpath) and B == BATCH:
            try:
                t = json.load(open(tpath))["k_pyr_fast"]
                traffic = int(2 * t["FETCH_SIZE_bytes_per_batch"] + t["WRITE_SIZE_bytes_per_batch"])
            except Exception:
                traffic = None
        out = {
            "metric": "ORB extract throughput (640x480, 1000 feat, 8 levels)",
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_frame": round(dt / args.steps * 1e3 / B, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"batch of {B} synthetic 640x480 mono frames per GPU, nFeatures=1000, 8 levels, scale 1.2 (BASELINE configs[1])",
                       "frames_per_gpu_per_step": B, "keypoints_per_frame": round(total_kps / B, 1),
                       "parallelism": f"frame-sharded x{world}", "gather": bool(gather_bufs is not None)},
            "stages_ms": {name: round(float(ms), 4) for name, ms in zip(pkg.STAGE_NAMES, stage_ms)},
            "roofline": {"bound": "hbm", "kernel": "k_pyr_fast (8 per-level launches = one pyramid+FAST pass)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch_set": ALGO_BYTES_PYR_FAST_PER_FRAME * B,
                         "avg_duration_ms": round(pyr_ms, 4), "batches_timed": int(nb)},
        }
        if pcie_fps is not None:
            out["pcie_inclusive_frames_per_s"] = round(pcie_fps, 1)
            out["pcie_inclusive_pinned_input_frames_per_s"] = round(pcie_pinned_fps, 1)
        if not args.no_cpu_baseline:
            orc = graft.load_oracle()
            threads = min(os.cpu_count() or 1, 16)
            sample = host[: min(args.cpu_sample, B)]
            cfps, cdt, _ = cpu_baseline(orc, sample, threads)
            out["cpu_baseline"] = {"value": round(cfps, 2), "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": f"{len(sample)} of the same synthetic frames, one frame per task on {threads} threads, {cdt:.1f} s wall"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
