"""Frame sharding across GPUs: one process per GPU, frame (or stereo pair) f -> rank f mod G, each
rank extracts its own frames with a private handle, and the fixed-capacity result records are gathered so that the
output slot of a frame is its frame index, independent of completion order.

The extraction path itself has no data-path collective; the only communication is this result gather.  It works on
torch tensors wherever they live: on the GPU node the tensors are zero-copy views of the handle's HBM output buffers
and the collective is RCCL over xGMI (backend "nccl"); in the CPU tests they are host tensors and the backend is gloo.
`bench.py --gather` and `tests/test_sharding_gloo.py` both call `gather_frame_records`, so the code the multi-GPU run
executes is the code the tests cover.  The record of one frame is {keypoint[cap] (28 B), desc[cap][32], int32 n,
int32 mono} (60.5 KB at nFeatures=1000), kept as four tensors with the frame slot as leading dimension.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

KP_BYTES = 28
DESC_BYTES = 32
FIELDS = ("kps", "desc", "n", "mono")


def frames_of_rank(n_frames: int, world: int, rank: int) -> np.ndarray:
    """Global frame indices owned by `rank` (round-robin: f mod world == rank), in increasing order."""
    return np.arange(rank, n_frames, world, dtype=np.int64)


def slots_per_rank(n_frames: int, world: int) -> int:
    """Record slots every rank contributes to the gather (ranks with one frame fewer pad with an empty record)."""
    return (n_frames + world - 1) // world


def record_bytes(cap: int) -> int:
    return 8 + cap * (KP_BYTES + DESC_BYTES)


def device_output_tensors(ex, nframes: int, device: str) -> Dict[str, "object"]:
    """Zero-copy torch views of an ORBextractor's HBM output buffers for the first `nframes` frames of its last batch:
    kps [nframes, cap * 28] u8, desc [nframes, cap * 32] u8, n [nframes] i32, mono [nframes] i32."""
    import torch

    class _View:  # __cuda_array_interface__ wrapper of a raw device address
        def __init__(self, ptr, shape, typestr):
            self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}

    d_kps, d_desc, d_n, d_mono = ex.device_outputs()
    cap = ex.capacity
    return {
        "kps": torch.as_tensor(_View(d_kps, (nframes, cap * KP_BYTES), "|u1"), device=device),
        "desc": torch.as_tensor(_View(d_desc, (nframes, cap * DESC_BYTES), "|u1"), device=device),
        "n": torch.as_tensor(_View(d_n, (nframes,), "<i4"), device=device),
        "mono": torch.as_tensor(_View(d_mono, (nframes,), "<i4"), device=device),
    }


def gather_frame_records(local: Dict[str, "object"], n_frames: int, group=None, out: Optional[dict] = None) -> Dict[str, "object"]:
    """All-gathers the per-frame records and places them so that slot == global frame index, on the tensors' own device.

    local[field]: tensor [b, ...] of this rank's frames in increasing global order (b = len(frames_of_rank)); every field is
    padded to slots_per_rank rows (n = -1 marks a padding slot), gathered with all_gather_into_tensor - rank-major
    [world, slots, ...] - and interleaved back to frame order: frame f = slot * world + rank, i.e. a transpose of the two
    leading dimensions, done by the device.  Returns {field: [n_frames, ...]}.  `out` may hold pre-allocated gather buffers
    ({field: [world * slots, ...]}) to keep allocations out of a timed loop."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    slots = slots_per_rank(n_frames, world)
    result = {}
    for name in FIELDS:
        t = local[name]
        b = t.shape[0]
        if b < slots:  # pad: only ranks with one frame fewer than the others
            pad = torch.zeros((slots - b,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            if name == "n":
                pad -= 1
            t = torch.cat([t, pad], 0)
        t = t.contiguous()
        if world == 1:
            full = t
        else:
            full = out[name] if out is not None else torch.empty((world * slots,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, t, group=group)
        # [world, slots, ...] -> [slots, world, ...] -> frame order
        g = full.reshape((world, slots) + tuple(t.shape[1:])).transpose(0, 1).reshape((slots * world,) + tuple(t.shape[1:]))
        result[name] = g[:n_frames]
    return result


def extract_sharded(extract_fn: Callable[[np.ndarray], dict], frames_of: Callable[[np.ndarray], np.ndarray], n_frames: int, group=None):
    """Runs `extract_fn(images[b, H, W]) -> {kps, desc, n, mono}` (torch tensors, rows = this rank's frames) on this rank's
    frames f mod world == rank and gathers the records of all ranks in frame order (see gather_frame_records).
    `frames_of(indices)` materialises the images of the given global frame indices (each rank reads only its own)."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = frames_of_rank(n_frames, world, rank)
    local = extract_fn(frames_of(mine))  # (a rank without frames still returns correctly shaped empty tensors)
    return gather_frame_records(local, n_frames, group)


def records_to_numpy(rec: Dict[str, "object"], cap: int, kp_dtype):
    """(n, mono, kps[n_frames, cap] structured, desc[n_frames, cap, 32]) host arrays of gathered records."""
    import torch  # noqa: F401

    n = rec["n"].cpu().numpy()
    mono = rec["mono"].cpu().numpy()
    kps = np.ascontiguousarray(rec["kps"].cpu().numpy()).view(kp_dtype).reshape(-1, cap)
    desc = rec["desc"].cpu().numpy().reshape(-1, cap, DESC_BYTES)
    return n, mono, kps, desc
