#!/usr/bin/env python3
"""bench.py — headline benchmark of the ORB front-end hot path (BASELINE.json metric).

Default (`--config c2`, BASELINE.json configs[1], the configuration the metric is quoted on): one "step" = one pass of
ORBextractor::operator() (pyramid -> FAST/NMS -> quadtree -> IC-angle -> blur -> rBRIEF -> pack) over ONE batch of 1024
synthetic 640x480 mono frames, nFeatures=1000, 8 levels, scale 1.2, per GPU - over `--batches-per-step` (c2: 18) DISTINCT batches per step
(18,432 frames, 5.7 GB of the 288 GB of HBM), so the timed region covers 360 batches / 1.1 s at the default 20 steps.  `value` = frames/s with the inputs resident in HBM before
the timed region and the results left in HBM (`value_definition` in the line says so).  SURVEY §8(d)'s metric - the PCIe-inclusive
rate of the host-pointer API orbx_extract_batch, H2D of the frames and D2H of keypoints + descriptors inside the call, page-locked
caller buffers - is the top-level `metric_8d_frames_per_s` (two batches in flight on the handle through orbx_extract_batch_submit / _wait; `pcie_inclusive_*` per kind of
caller memory and for one call at a time), measured on EVERY rank
at every N (each rank its own host buffers, barrier on both sides, aggregate = all ranks' frames / slowest rank's wall time) - never `value`.

Other workloads (`--config`): c3 = 752x480 stereo pairs, nFeatures=1200, extraction of both images +
Frame::ComputeStereoMatches per pair; c4 = 512x512 mono, nFeatures=1500; c5 = 1920x1080 stereo pairs, nFeatures=4000,
extraction + brute-force kNN-2 Hamming cross-match (4000 x 4000 per pair).  Stereo steps run the two extractors and the
device-resident batched matcher entry point; nothing crosses PCIe inside a step.

Multi-GPU: `python bench.py --gpus N` starts its own N ranks (a child `python -m torch.distributed.run`, one process per GPU over RCCL;
the parent never touches the GPU); under an external launcher (RANK set) it is a rank.  Frames shard across ranks, no data-path collective.
--scaling weak (default): every rank runs its own batch per step.  --scaling strong: a fixed stream (c3: the 3,682 pairs of
EuRoC MH01) is dealt frame f -> rank f mod N and every rank works through its share in batches.  At N > 1 the fixed-capacity
result records of every batch are all-gathered over RCCL INSIDE the timed region and placed by frame index on the device (north_star's
"trivial gather"; wut-cuda-orb-slam3_amd/sharding.py, the code path tests/test_sharding_gloo.py covers) and `gather_ms` reports the
collective's share per step; --no-gather leaves it out, --gather forces it at N = 1 (world-1 all-gather = a copy).  A rehearsal on
fewer devices than ranks (gloo rendezvous) has no RCCL: `gather_ms` is null there.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import concurrent.futures
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

NLEVELS, SCALE, INI_TH, MIN_TH = 8, 1.2, 20, 7
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PCIE_PEAK_GBS = 63.0        # MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s (spec)
MFMA_I8_16x16x64_CYCLES = 22.0   # measured back-to-back issue of v_mfma_i32_16x16x64_i8 on one SIMD (tools/experiments/mfma_i8_rate.hip; 16 by the data sheet)
N_SIMD, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md)
CONFIGS = {
    # units = what one GPU processes per step; a stereo unit is a pair (2 frames)
    "c2": dict(w=640, h=480, nf=1000, stereo=False, units=1024, match=None, lap=(0, 1000), stream=8192,
               name="batch of {u} synthetic 640x480 mono frames per GPU, nFeatures=1000, 8 levels, scale 1.2 (BASELINE configs[1])"),
    "c3": dict(w=752, h=480, nf=1200, stereo=True, units=512, match="stereo", lap=(0, 0), stream=3682,
               name="batch of {u} synthetic 752x480 stereo pairs per GPU (EuRoC MH01 geometry), nFeatures=1200: extract L + R, "
                    "Frame::ComputeStereoMatches per pair (BASELINE configs[2])"),
    "c4": dict(w=512, h=512, nf=1500, stereo=False, units=1024, match=None, lap=(0, 1000), stream=8192,
               name="batch of {u} synthetic 512x512 mono frames per GPU (TUM-VI geometry), nFeatures=1500 (BASELINE configs[3])"),
    "c5": dict(w=1920, h=1080, nf=4000, stereo=True, units=128, match="knn2", lap=(0, 0), stream=1024,
               name="batch of {u} synthetic 1920x1080 stereo pairs per GPU, nFeatures=4000: extract L + R, brute-force 256-bit "
                    "Hamming kNN-2 cross-match per pair (BASELINE configs[4])"),
}
MB, MBF = 0.11, 47.9  # stereo baseline [m] and baseline x fx (EuRoC-like) for ComputeStereoMatches


def kernel_source_sha256():
    """Hash of the HIP sources liborbx.so is built from: stamps PMC summaries (tools/summarize_traffic.py) and tells bench.py whether a
    committed counter file still describes the kernels it is about to time."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "wut-cuda-orb-slam3_amd", "csrc")
    # (the Makefile and tools/sq_slowmode.py too: the device assembly passes through the s_nop pass before it is assembled)
    for name in (os.path.join(d, "orbx_kernels.hip"), os.path.join(d, "orbx_internal.h"), os.path.join(d, "Makefile"),
                 os.path.join(ROOT, "tools", "sq_slowmode.py")):
        h.update(open(name, "rb").read())
    return h.hexdigest()


def level_pixels(ex):
    return [ex.level_size(l)[0] * ex.level_size(l)[1] for l in range(NLEVELS)]


def make_frames(pkg, cfg, first, count):
    """Synthetic frames of units [first, first + count): mono [count, H, W]; stereo (left, right) with the right image a
    horizontally shifted copy (disparity 6 + (unit mod 24) px)."""
    w, h = cfg["w"], cfg["h"]
    left = np.empty((count, h, w), np.uint8)
    for i in range(count):
        left[i] = pkg.synth_frame(first + i, w, h)
    if not cfg["stereo"]:
        return left, None
    right = np.empty_like(left)
    for i in range(count):
        right[i] = np.roll(left[i], -(6 + (first + i) % 24), axis=1)
    return left, right


# ---- CPU baseline (the oracle = CPU restatement of the reference algorithm, kind "port") -----------------------------------
def _cpu_units(orc, cfg, left, right, threads):
    """Processes every unit of the sample once with `threads` host threads; returns wall seconds."""
    nf = cfg["nf"]
    if not cfg["stereo"]:
        ref = orc.Oracle(nf, SCALE, NLEVELS, INI_TH, MIN_TH)
        t0 = time.perf_counter()
        ref.extract_batch(left, cfg["lap"], threads=threads)
        return time.perf_counter() - t0
    pool = [(orc.Oracle(nf, SCALE, NLEVELS, INI_TH, MIN_TH), orc.Oracle(nf, SCALE, NLEVELS, INI_TH, MIN_TH)) for _ in range(threads)]

    def one(args):
        slot, i = args
        ol, orr = pool[slot]
        _, kl, dl = ol.extract(left[i], cfg["lap"])
        _, kr, dr = orr.extract(right[i], cfg["lap"])
        if cfg["match"] == "stereo":
            orc.stereo_matches(ol, orr, kl, dl, kr, dr, MB, MBF)
        else:
            orc.knn2(dl, dr)

    t0 = time.perf_counter()
    if threads == 1:
        for i in range(len(left)):
            one((0, i))
    else:  # unit i on worker i mod threads (ctypes releases the GIL inside the oracle)
        with concurrent.futures.ThreadPoolExecutor(threads) as ex:
            futs = [ex.submit(lambda s=s: [one((s, i)) for i in range(s, len(left), threads)]) for s in range(threads)]
            [f.result() for f in futs]
    return time.perf_counter() - t0


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_leg(cfg, cfg_key, left, right, budget_s):
    """Times the oracle on a bounded sample of the same workload: all host threads (<= 16) and one thread, -O3 build; then the same
    two runs of a -O3 -march=native build compiled here (the GPU box's CPU differs from the build container's)."""
    orc = graft.load_oracle()
    threads = min(os.cpu_count() or 1, 16)
    frames_per_unit = 2 if cfg["stereo"] else 1

    def measure(orc_mod):
        # all threads: repeat the sample until ~budget_s of wall time
        n_units, wall = 0, 0.0
        _cpu_units(orc_mod, cfg, left[:threads], None if right is None else right[:threads], threads)  # warm-up
        while wall < budget_s and n_units < 16 * len(left):
            wall += _cpu_units(orc_mod, cfg, left, right, threads)
            n_units += len(left)
        n1 = max(2, min(len(left), int(round(0.2 * budget_s * n_units / wall / threads))))
        w1 = _cpu_units(orc_mod, cfg, left[:n1], None if right is None else right[:n1], 1)
        return n_units * frames_per_unit / wall, wall, n_units, w1 / (n1 * frames_per_unit) * 1e3, n1

    fps, wall, n_units, ms1, n1 = measure(orc)
    out = {"value": round(fps, 2), "unit": "frames/s", "cores": threads, "kind": "port", "cpu_model": _cpu_model(),
           "ms_per_frame_1thread": round(ms1, 3),
           "sample": f"{n_units} units ({n_units * frames_per_unit} frames) of the same synthetic {cfg_key} workload on {threads} threads, "
                     f"{wall:.1f} s wall; 1-thread figure on {n1} units; build g++ -O3 (oracle/Makefile)"}
    # -march=native variant in a child process (the oracle library is chosen at import time)
    try:
        tmp = tempfile.mkdtemp(prefix="orbx_native_")
        so = os.path.join(tmp, "liborb_oracle_native.so")
        subprocess.run(["g++", "-O3", "-march=native", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread", "-shared", "-o", so,
                        os.path.join(ROOT, "oracle", "orb_oracle.cpp")], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        np.save(os.path.join(tmp, "left.npy"), left)
        if right is not None:
            np.save(os.path.join(tmp, "right.npy"), right)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", tmp, "--config", cfg_key, "--cpu-budget", str(budget_s)],
                           env=dict(os.environ, ORB_ORACLE_LIB=so), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        child = json.loads(r.stdout.strip().splitlines()[-1])
        out["march_native"] = {"value": child["fps"], "ms_per_frame_1thread": child["ms1"], "build": "g++ -O3 -march=native"}
    except Exception as e:  # the baseline above stands; say why the variant is missing
        out["march_native"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    return out


def cpu_child(tmp, cfg_key, budget_s):
    cfg = CONFIGS[cfg_key]
    orc = graft.load_oracle()
    left = np.load(os.path.join(tmp, "left.npy"))
    right = np.load(os.path.join(tmp, "right.npy")) if cfg["stereo"] else None
    threads = min(os.cpu_count() or 1, 16)
    fpu = 2 if cfg["stereo"] else 1
    _cpu_units(orc, cfg, left[:threads], None if right is None else right[:threads], threads)
    n_units, wall = 0, 0.0
    while wall < budget_s and n_units < 16 * len(left):
        wall += _cpu_units(orc, cfg, left, right, threads)
        n_units += len(left)
    n1 = max(2, min(len(left), int(round(0.2 * budget_s * n_units / wall / threads))))
    w1 = _cpu_units(orc, cfg, left[:n1], None if right is None else right[:n1], 1)
    print(json.dumps({"fps": round(n_units * fpu / wall, 2), "ms1": round(w1 / (n1 * fpu) * 1e3, 3)}))


def self_launch(ngpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a
    CHILD process (never exec: the driver's own process may have initialised the GPU) on a free local port and pass its output through."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif r.returncode == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return r.returncode


# ---- main ------------------------------------------------------------------------------------------------------------------
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="workload (default c2 = the metric's configuration)")
    ap.add_argument("--batch", type=int, default=0, help="units (frames / stereo pairs) per GPU per step (default: the config's)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--batches-per-step", type=int, default=0,
                    help="distinct batches one step passes over (weak scaling); default: c2 18 (timed region >= 1 s at 20 steps), other configs 5")
    ap.add_argument("--gather", action="store_true", help="all-gather the result records over RCCL inside the timed region (default at N > 1)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1 without the result gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=5.0, help="seconds of all-thread CPU work per oracle build in the baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=0, help="units in the CPU-baseline sample (default: by config)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-pointer (PCIe-inclusive) timing")
    ap.add_argument("--pcie", action="store_true", help="(kept for compatibility: the PCIe-inclusive timing is on by default)")
    ap.add_argument("--cpu-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_child:
        cpu_child(args.cpu_child, args.config, args.cpu_budget)
        return
    if args.batches_per_step <= 0:
        args.batches_per_step = 18 if args.config == "c2" else 5
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process never touches the GPU (no torch / HIP import so far); it starts the N ranks as
        # child processes under torch.distributed.run, forwards rank 0's JSON line and returns the launcher's exit code
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1 (RANK is set: launched by something that is not torch.distributed.run?)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    # (rehearsal on a box with fewer GPUs than ranks: the ranks share the devices round-robin and rendezvous over gloo - RCCL needs one
    # device per rank, so the result gather is then not available; chosen automatically, or forced with ORBX_BENCH_BACKEND=gloo)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("ORBX_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
    rehearsal = backend != "nccl" and world > 1
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            if args.gather:
                raise SystemExit("--gather needs the nccl (RCCL) backend")
            dist.init_process_group(backend)
    # the result gather is part of the multi-GPU job (north_star): on by default at N > 1 over RCCL
    do_gather = (args.gather or (world > 1 and backend == "nccl")) and not args.no_gather

    pkg = graft.load_package()
    import importlib
    sharding = importlib.import_module("wut_cuda_orb_slam3_amd.sharding")
    cfg = CONFIGS[args.config]
    W, H, NFEAT, stereo = cfg["w"], cfg["h"], cfg["nf"], cfg["stereo"]
    B = args.batch or cfg["units"]
    fpu = 2 if stereo else 1

    # ---- this rank's units -------------------------------------------------------------------------------------------------
    # weak: units rank * B .. rank * B + B - 1, the same batch every step.  strong: global stream of cfg["stream"] units dealt
    # f -> rank f mod world; the rank works through its share in sub-batches of B, `steps` passes over the whole stream.
    if args.scaling == "weak":
        per_rank = B * max(args.batches_per_step, 1)
        my_units = np.arange(rank * per_rank, rank * per_rank + per_rank)
        global_units = world * per_rank
    else:
        global_units = cfg["stream"]
        my_units = sharding.frames_of_rank(global_units, world, rank)
    # contiguous synthetic ids are generated once; unit u uses synthetic frame id u
    chunks = []  # (d_left, d_right, count)
    host_first = None
    for c0 in range(0, len(my_units), B):
        ids = my_units[c0: c0 + B]
        left = np.empty((len(ids), H, W), np.uint8)
        right = np.empty_like(left) if stereo else None

        def fill(j):
            l1, r1 = make_frames(pkg, cfg, int(ids[j]), 1)
            left[j] = l1[0]
            if stereo:
                right[j] = r1[0]

        with concurrent.futures.ThreadPoolExecutor(min(os.cpu_count() or 1, 8)) as tp:  # (the generator is C code behind ctypes: no GIL)
            list(tp.map(fill, range(len(ids))))
        if host_first is None:
            host_first = (left, right)
        # units of this chunk over all ranks: weak = every rank its own B; strong = the global range [world * c0, world * (c0 + B))
        gtotal = world * len(ids) if args.scaling == "weak" else min(global_units, world * (c0 + B)) - world * c0
        chunks.append((torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev) if stereo else None, len(ids), gtotal))
    torch.cuda.synchronize()

    exl = pkg.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=W, max_height=H, max_batch=B)
    exr = pkg.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=W, max_height=H, max_batch=B) if stereo else None
    exl.set_profiling(True)
    cap = exl.capacity
    matcher = pkg.ORBmatcher(cap, cap, device=local_rank) if stereo else None
    m_out = None
    if stereo:
        if cfg["match"] == "stereo":
            m_out = (torch.empty((B, cap), dtype=torch.float32, device=dev), torch.empty((B, cap), dtype=torch.float32, device=dev))
        else:
            m_out = (torch.empty((B, cap, 2), dtype=torch.int32, device=dev), torch.empty((B, cap, 2), dtype=torch.int32, device=dev))
    lib = pkg.load_library()
    import ctypes

    def match(n):
        if cfg["match"] == "stereo":
            rc = lib.orbx_compute_stereo_matches_frames(matcher._h, exl._h, exr._h, n, MB, MBF, ctypes.c_void_p(m_out[0].data_ptr()),
                                                        ctypes.c_void_p(m_out[1].data_ptr()))
        else:
            rc = lib.orbx_match_knn2_frames(matcher._h, exl._h, exr._h, n, 0, ctypes.c_void_p(m_out[0].data_ptr()), ctypes.c_void_p(m_out[1].data_ptr()))
        if rc != 0:
            raise SystemExit(f"matcher call failed: {lib.orbx_last_error().decode()}")

    # result gather (north_star's "trivial gather"): zero-copy views of the left extractor's HBM outputs, RCCL all-gather,
    # device-side placement by frame index; ordered against the library's stream without a host synchronisation
    gather_state = None
    if do_gather:
        ext = torch.cuda.ExternalStream(exl.stream(), device=dev)
        gather_state = {"ext": ext, "events": [], "timed": False}

    def gather(n, gtotal):
        cur = torch.cuda.current_stream()
        cur.wait_stream(gather_state["ext"])            # results of this rank are complete
        if gather_state["timed"]:                       # the collective + placement alone, on the stream it runs on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        local = sharding.device_output_tensors(exl, n, dev)
        rec = sharding.gather_frame_records(local, gtotal)  # slot = frame index within the chunk's global range
        if gather_state["timed"]:
            e1.record(cur)
            gather_state["events"].append((e0, e1))
        gather_state["ext"].wait_stream(cur)            # the next batch overwrites the buffers only after the gather has read them
        return rec

    def step():
        for d_left, d_right, n, gtotal in chunks:
            exl.extract_batch_device(d_left.data_ptr(), n, W, H, W, W * H, cfg["lap"])
            if stereo:
                exr.extract_batch_device(d_right.data_ptr(), n, W, H, W, W * H, cfg["lap"])
                match(n)
            if gather_state is not None:
                gather(n, gtotal)

    def fence():
        exl.sync()
        if stereo:
            exr.sync()
            matcher.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    exl.stage_times()  # reset the event ring
    if gather_state is not None:
        gather_state["timed"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stage_ms, nb = exl.stage_times()
    gather_ms = None
    if gather_state is not None:
        gather_state["timed"] = False
        gather_ms = sum(a.elapsed_time(b) for a, b in gather_state["events"]) / max(args.steps, 1)

    def all_reduce(x, op):
        if world == 1:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return float(t.item())

    dt = all_reduce(dt, "max")
    if gather_ms is not None:
        gather_ms = all_reduce(gather_ms, "max")

    n_last = chunks[-1][2]
    n, mono, _, _ = exl.download_outputs(n_last)
    total_kps = int(n.sum())

    # ---- every rank: SURVEY 8(d)'s metric - orbx_extract_batch with host frames in (H2D) and keypoints + descriptors out (D2H) inside
    # the call.  Three callers: everything pageable; frames page-locked; frames AND result arrays page-locked (orbx_host_alloc).  Result
    # arrays are the caller's and reused between calls, as the reference's caller reuses its containers.  At N > 1 every rank runs the
    # leg on its own buffers at the same time (barrier on both sides): the ranks share the host's DRAM and root complexes, which is the
    # scaling question of this metric.  Aggregate = all ranks' frames / the slowest rank's wall time.
    pcie = None
    if not args.no_pcie:
        hl, hr = host_first
        np_units = min(len(hl), chunks[0][2])
        hl, hr = hl[:np_units], (hr[:np_units] if stereo else None)

        def locked(a):
            p = pkg.host_array(a.shape, a.dtype)
            p[...] = a
            return p

        pl, pr = locked(hl), (locked(hr) if stereo else None)
        outs = {False: (exl.output_arrays(np_units), exr.output_arrays(np_units) if stereo else None),
                True: (exl.output_arrays(np_units, page_locked=True), exr.output_arrays(np_units, page_locked=True) if stereo else None)}

        def host_call(a, b, lockedOut):
            # a stereo pair's two extractors on two host threads, as the reference runs them (src/Frame.cc:124-127): the copies of one
            # handle's sub-batches fill the link while the other one's kernels run (ctypes releases the GIL inside the call)
            # - with page-locked frames.  Two callers with PAGEABLE frames contend inside the runtime (it pins the pages of a copy behind
            # the call): measured 15 % slower than one call after the other, so those run in turn.
            t = None
            if stereo and a is pl:
                t = threading.Thread(target=lambda: exr.extract_batch(b, cfg["lap"], out=outs[lockedOut][1]))
                t.start()
            exl.extract_batch(a, cfg["lap"], out=outs[lockedOut][0])
            if t:
                t.join()
            elif stereo:
                exr.extract_batch(b, cfg["lap"], out=outs[lockedOut][1])

        def fresh_arrays(ex):
            # the reference's own calling pattern: `_descriptors.create(n, 32, CV_8U)` and `_keypoints = vector<KeyPoint>(n)` allocate NEW
            # result arrays in every call (src/ORBextractor.cc:1251,1257) - untouched pages, first written inside the call
            return (np.empty((np_units,), np.int32), np.empty((np_units,), np.int32), np.empty((np_units, ex.capacity), pkg.KEYPOINT_DTYPE),
                    np.empty((np_units, ex.capacity, 32), np.uint8))

        pcie = {}
        reps = 5
        for tag, (a, b, lo) in (("pageable", (hl, hr, False)), ("pinned_input", (pl, pr, False)), ("page_locked", (pl, pr, True)),
                                ("fresh_result_arrays", (pl, pr, "fresh"))):
            if lo == "fresh":
                outs["fresh"] = (fresh_arrays(exl), fresh_arrays(exr) if stereo else None)
            host_call(a, b, lo)
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                if lo == "fresh":
                    outs["fresh"] = (fresh_arrays(exl), fresh_arrays(exr) if stereo else None)   # (np.empty: a mapping, no page is touched here)
                host_call(a, b, lo)
            wall = time.perf_counter() - t1
            if world > 1:
                dist.barrier()
            pcie[tag] = {"rank_frames_per_s": reps * np_units * fpu / wall,
                         "frames_per_s": all_reduce(reps * np_units * fpu, "sum") / all_reduce(wall, "max")}

        # the same API with two batches in flight per handle (orbx_extract_batch_submit / _wait): the frames of a batch cross the link - into the
        # handle's second staging buffer - under the kernels and the copy-out of the batch before, so the link never waits for the GPU.  One
        # host thread; stereo: the left and the right handle in turn, two batches in flight on each.
        lanes = [(exl, pl)] + ([(exr, pr)] if stereo else [])
        lane_outs = [[ex_.output_arrays(np_units, page_locked=True) for _ in range(2)] for ex_, _ in lanes]
        for (ex_, fr_), oo in zip(lanes, lane_outs):        # warm-up: one synchronous call each (allocates nothing inside the timed part
            ex_.extract_batch(fr_, cfg["lap"], out=oo[0])     # but the second staging buffer, which the first double submit below creates)
        for (ex_, fr_), oo in zip(lanes, lane_outs):
            ex_.extract_batch_submit(fr_, cfg["lap"], out=oo[0]); ex_.extract_batch_submit(fr_, cfg["lap"], out=oo[1])
        for (ex_, fr_), oo in zip(lanes, lane_outs):
            ex_.extract_batch_wait(); ex_.extract_batch_wait()
        if world > 1:
            dist.barrier()
        calls = 2 * reps
        t1 = time.perf_counter()
        for (ex_, fr_), oo in zip(lanes, lane_outs):
            ex_.extract_batch_submit(fr_, cfg["lap"], out=oo[0]); ex_.extract_batch_submit(fr_, cfg["lap"], out=oo[1])
        for k in range(2, calls):
            for (ex_, fr_), oo in zip(lanes, lane_outs):
                ex_.extract_batch_wait()
                ex_.extract_batch_submit(fr_, cfg["lap"], out=oo[k & 1])
        for (ex_, fr_), oo in zip(lanes, lane_outs):
            ex_.extract_batch_wait(); ex_.extract_batch_wait()
        wall = time.perf_counter() - t1
        if world > 1:
            dist.barrier()
        pcie["page_locked_pipelined"] = {"rank_frames_per_s": calls * np_units * fpu / wall,
                                         "frames_per_s": all_reduce(calls * np_units * fpu, "sum") / all_reduce(wall, "max")}

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- rank 0: roofline of the dominant kernel, matcher block, PCIe-inclusive rate, CPU baseline ---------------------------------
    d_left0, d_right0, n0, _ = chunks[0]
    # pyramid + FAST launch set alone on the GPU (stereo configs run two handles side by side in the timed loop)
    if stereo or any(c[2] != n0 for c in chunks):
        for _ in range(2):
            exl.extract_batch_device(d_left0.data_ptr(), n0, W, H, W, W * H, cfg["lap"])
        exl.sync()
        exl.stage_times()
        for _ in range(5):
            exl.extract_batch_device(d_left0.data_ptr(), n0, W, H, W, W * H, cfg["lap"])
        exl.sync()
        solo_ms, solo_nb = exl.stage_times()
    else:
        solo_ms, solo_nb = stage_ms, nb
    px = level_pixels(exl)
    algo_bytes_frame = sum(px) + sum(px[1:])  # read every level once, write levels 1..7 once
    pyr_ms = float(solo_ms[0])
    achieved = algo_bytes_frame * n0 / (pyr_ms * 1e-3) / 1e9 if pyr_ms > 0 else 0.0
    hbm = {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
    roof = {"bound": "hbm", "kernel": "k_pyr_fast (8 per-level launches = one pyramid+FAST pass over the batch)", **hbm,
            "algorithmic_bytes_per_launch_set": algo_bytes_frame * n0, "avg_duration_ms": round(pyr_ms, 4), "batches_timed": int(solo_nb)}
    # HBM bytes and VALU instruction count of the same launch set from the committed PMC passes (tools/collect_traffic.sh: separate
    # rocprofv3 runs; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md).  They are NOT measured in this run, so they
    # carry their source, and they are dropped when the kernel sources have changed since they were taken (sha256 recorded by
    # tools/summarize_traffic.py): a stale file must not dress up a new kernel.
    cpath = os.path.join(ROOT, "profiles", "pyr_fast_counters_latest.json")
    if args.config == "c2" and n0 == 1024:
        try:
            c = json.load(open(cpath))
            if c.get("kernel_source_sha256") != kernel_source_sha256():
                raise ValueError(f"counters were taken on other kernel sources ({str(c.get('kernel_source_sha256'))[:12]} != "
                                 f"{kernel_source_sha256()[:12]}): re-run tools/collect_traffic.sh")
            src = {"source": "profiles/pyr_fast_counters_latest.json", "taken_at_commit": c.get("commit"),
                   "kernel_source_sha256": c["kernel_source_sha256"][:16], "note": "PMC passes of an earlier run of the same command, not this run"}
            roof["traffic"] = int(2 * c["FETCH_SIZE_bytes_per_batch"] + c["WRITE_SIZE_bytes_per_batch"])
            roof["traffic_source"] = src
            valu = float(c["SQ_INSTS_VALU_per_batch"])
            cyc = float(c.get("valu_cycles_per_instr_mix", 3.0))
            # the kernel is bound by vector-instruction issue, not by HBM: wave-instructions per second against what 1024 SIMDs issue at
            # 2 cycles per instruction (the best case; at the measured cost of this instruction mix, profiles/r01_valu_microbench.txt:
            # 2.4-2.7 cycles for add/and/bitop3/shift, 4.1-4.5 for min/max/mul/alignbyte/SDWA forms, the peak is lower: `peak_mix`)
            ach = valu / (pyr_ms * 1e-3) / 1e9
            # `roofline` stays the HBM roofline SURVEY 8(d) / the bench contract define (achieved = algorithmic bytes / launch-set time,
            # peak 8 TB/s); the vector-issue view of the same launch set rides along as `valu_issue`, against the 2-cycle issue peak
            # (1024 SIMD-32 x 2.4 GHz / 2) and against the peak of this instruction mix by class (`cycles_per_instr_mix`)
            roof["valu_issue"] = {"achieved": round(ach, 1), "peak": round(N_SIMD * CLOCK_GHZ / 2.0, 1), "unit": "G wave-instr/s",
                                  "frac": round(ach / (N_SIMD * CLOCK_GHZ / 2.0), 4), "peak_mix": round(N_SIMD * CLOCK_GHZ / cyc, 1),
                                  "frac_mix": round(ach / (N_SIMD * CLOCK_GHZ / cyc), 3), "valu_wave_instr_per_launch_set": int(valu),
                                  "cycles_per_instr_mix": cyc, "clock_ghz": CLOCK_GHZ, "valu_source": src}
        except Exception as e:  # the HBM block above stands (it is measured live); say why the counter-based part is missing
            roof["counters_dropped"] = f"{type(e).__name__}: {e}"[:240]

    frames_total = global_units * fpu * args.steps
    fps = frames_total / dt
    out = {
        "metric": "ORB extract throughput (640x480, 1000 feat, 8 levels)" if args.config == "c2" else
                  f"ORB front-end throughput ({W}x{H}{' stereo' if stereo else ''}, {NFEAT} feat, 8 levels)",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_frame": round(dt / frames_total * 1e3, 6),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "value_definition": "frames/s of the whole job with the frames resident in HBM when the timed region starts and the results left in HBM"
                            + (", result gather over RCCL inside the timed region" if gather_state is not None else "")
                            + "; SURVEY 8(d)'s PCIe-inclusive rate is metric_8d_frames_per_s",
        "gather_ms": None if gather_ms is None else round(gather_ms, 4),
        **({"gather_note": "rehearsal (gloo rendezvous, ranks share devices): no RCCL gather"} if rehearsal else {}),
        "config": {"workload": cfg["name"].format(u=B), "config_id": args.config, "units_per_gpu_per_step": int(len(my_units)),
                   "batches_per_step": len(chunks), "distinct_frames_per_gpu": int(len(my_units)) * fpu,
                   "frames_per_unit": fpu, "keypoints_per_frame": round(total_kps / n_last, 1),
                   "parallelism": f"frame-sharded x{world}" + (f", stream of {global_units} units dealt f mod {world}" if args.scaling == "strong" else ""),
                   "gather": bool(gather_state is not None),
                   **({"rehearsal": f"{world} ranks share {ndev} device(s), gloo rendezvous (not a scaling measurement)"} if rehearsal else {})},
        "stages_ms": {name: round(float(ms), 4) for name, ms in zip(pkg.STAGE_NAMES, stage_ms)},
        "roofline": roof,
    }
    if stereo:
        out["pairs_per_s"] = round(fps / 2, 1)
        # the matcher alone: device-resident batch call, wall-timed with a stream sync (on the first chunk's extraction results)
        exl.extract_batch_device(d_left0.data_ptr(), n0, W, H, W, W * H, cfg["lap"])
        exr.extract_batch_device(d_right0.data_ptr(), n0, W, H, W, W * H, cfg["lap"])
        match(n0); matcher.sync()
        reps = 10
        t1 = time.perf_counter()
        for _ in range(reps):
            match(n0)
        matcher.sync()
        mt = (time.perf_counter() - t1) / reps
        blk = {"entry_point": "orbx_compute_stereo_matches_frames" if cfg["match"] == "stereo" else "orbx_match_knn2_frames",
               "pairs_per_call": n0, "ms_per_call": round(mt * 1e3, 4), "pairs_per_s": round(n0 / mt, 1)}
        if cfg["match"] == "knn2":
            nq = float(total_kps) / n_last
            dpairs = nq * nq * n0
            # k_knn2_scan_mfma: a 16 x 16 tile of (train row, query) pairs = 4 x v_mfma_i32_16x16x64_i8 of 16 cycles on one SIMD's
            # matrix core -> 256 pairs per 64 cycles.  (The vector-ALU scan it replaced: 8 v_xor + 8 v_bcnt + key, min, med3 = 65 issue
            # cycles per 64 pairs, ORBX_KNN_MFMA=0.)
            # (peak: at the MEASURED issue interval of that instruction, 22 cycles - the data-sheet 16 cycles is not reachable in a bare loop)
            peak = N_SIMD * 256 * CLOCK_GHZ * 1e9 / (4 * MFMA_I8_16x16x64_CYCLES)
            peak16 = N_SIMD * 256 * CLOCK_GHZ * 1e9 / 64.0
            valu_peak = N_SIMD * 64 * CLOCK_GHZ * 1e9 / 65.0
            blk.update({"descriptor_pairs_per_s": round(dpairs / mt, 0), "mfma_peak_descriptor_pairs_per_s": round(peak, 0),
                        "frac_of_mfma_peak": round(dpairs / mt / peak, 3), "mfma_issue_cycles_measured": MFMA_I8_16x16x64_CYCLES,
                        "frac_of_16_cycle_datasheet_peak": round(dpairs / mt / peak16, 3),
                        "valu_scan_issue_peak_descriptor_pairs_per_s": round(valu_peak, 0)})
        out["matcher"] = blk
    if pcie is not None:
        res = {tag: v["frames_per_s"] for tag, v in pcie.items()}
        out["metric_8d_frames_per_s"] = round(res["page_locked_pipelined"], 1)
        out["metric_8d_definition"] = ("SURVEY 8(d): the host-pointer batch API, host frames in (H2D) and keypoints + descriptors out (D2H) inside the calls, page-locked "
                                       "caller buffers, two batches in flight on the handle from one host thread (orbx_extract_batch_submit / _wait: a batch's frames "
                                       "cross the link under the kernels and the copy-out of the one before); one call at a time = pcie_inclusive_page_locked_frames_per_s; all ranks at the same "
                                       "time, all ranks' frames / slowest rank's wall time")
        out["pcie_inclusive_frames_per_s"] = round(res["pageable"], 1)
        out["pcie_inclusive_pinned_input_frames_per_s"] = round(res["pinned_input"], 1)
        out["pcie_inclusive_page_locked_frames_per_s"] = round(res["page_locked"], 1)
        out["pcie_inclusive_fresh_result_arrays_frames_per_s"] = round(res["fresh_result_arrays"], 1)
        out["pcie_inclusive_page_locked_pipelined_frames_per_s"] = round(res["page_locked_pipelined"], 1)
        bytes_in = W * H
        kpf = float(total_kps) / n_last
        payload_out = kpf * 60.0 + 8.0     # keypoints (28 B) + descriptors (32 B) per feature, two counts per frame
        direct_out = cap * 60.0 + 8.0      # page-locked result arrays laid out like the handle's: full-capacity rows cross the link
        link = {}
        for tag in res:
            gbs = res[tag] * bytes_in / 1e9 / world   # per GPU link
            link[tag] = {"frames_per_s": round(res[tag], 1), "achieved": round(gbs, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s H2D per GPU link",
                         "frac": round(gbs / PCIE_PEAK_GBS, 3), "d2h_payload_gbs": round(res[tag] * payload_out / 1e9 / world, 2),
                         "d2h_link_gbs": round(res[tag] * (direct_out if tag.startswith("page_locked") else payload_out) / 1e9 / world, 2)}
        out["pcie_roofline"] = {"bound": "pcie_h2d", "peak_note": "PCIe Gen5 x16, 63 GB/s per direction and GPU (MI355X_MICROARCH.md); frames cross H2D "
                                "while results cross D2H (full duplex), so the input direction is the bound; d2h_link_gbs = bytes that really crossed "
                                "(full-capacity rows in the page-locked direct mode), d2h_payload_gbs = the n[f] rows the caller uses; fresh_result_arrays = page-locked "
                                "frames, but NEW pageable result arrays in every call, as the reference's caller allocates them (src/ORBextractor.cc:1251,1257): "
                                "the first-touch page faults of the copy-out are inside the call", **link}
        out["pcie_inclusive_note"] = ("orbx_extract_batch: host frames in (H2D), keypoints + descriptors out (D2H), per call; extraction only"
                                      + (" (both images of a pair; page-locked frames: one host thread per extractor, pageable frames: one call after the other)" if stereo else "")
                                      + (f"; {world} ranks at the same time, each on its own host buffers" if world > 1 else ""))
    if not args.no_cpu_baseline and world == 1:
        hl, hr = host_first
        ns = args.cpu_sample or {"c2": 1024, "c3": 128, "c4": 1024, "c5": 16}[args.config]
        ns = min(ns, len(hl))
        out["cpu_baseline"] = cpu_baseline_leg(cfg, args.config, hl[:ns], hr[:ns] if stereo else None, args.cpu_budget)
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
