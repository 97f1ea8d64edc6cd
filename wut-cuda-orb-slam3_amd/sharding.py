"""Frame sharding across GPUs: one process per GPU, frame (or stereo pair) f -> rank f mod G, each
rank extracts its own frames with a private handle, and fixed-capacity result records are gathered so that the
output slot of a frame is its frame index, independent of completion order.

The extraction path itself has no data-path collective; the only communication is this result gather
(torch.distributed: backend "nccl" == RCCL over xGMI on the GPU node, "gloo" in the CPU tests).  The record of
one frame is {int32 n, int32 mono, keypoint[cap] (28 B), desc[cap][32]} (60.5 KB at nFeatures=1000).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

KP_BYTES = 28
DESC_BYTES = 32


def frames_of_rank(n_frames: int, world: int, rank: int) -> np.ndarray:
    """Global frame indices owned by `rank` (round-robin: f mod world == rank), in increasing order."""
    return np.arange(rank, n_frames, world, dtype=np.int64)


def slots_per_rank(n_frames: int, world: int) -> int:
    """Record slots every rank contributes to the gather (ranks with one frame fewer pad with an empty record)."""
    return (n_frames + world - 1) // world


def record_bytes(cap: int) -> int:
    return 8 + cap * (KP_BYTES + DESC_BYTES)


def pack_records(n: np.ndarray, mono: np.ndarray, kps: np.ndarray, desc: np.ndarray, cap: int, slots: int) -> np.ndarray:
    """[slots, record_bytes] uint8 from per-frame outputs (n[b], mono[b], kps[b, cap] structured, desc[b, cap, 32])."""
    b = len(n)
    rec = np.zeros((slots, record_bytes(cap)), np.uint8)
    hdr = np.zeros((slots, 2), np.int32)
    hdr[:b, 0] = n
    hdr[:b, 1] = mono
    hdr[b:, 0] = -1  # padding slot
    rec[:, :8] = hdr.view(np.uint8).reshape(slots, 8)
    if b:
        rec[:b, 8: 8 + cap * KP_BYTES] = np.ascontiguousarray(kps).view(np.uint8).reshape(b, cap * KP_BYTES)
        rec[:b, 8 + cap * KP_BYTES:] = np.ascontiguousarray(desc).reshape(b, cap * DESC_BYTES)
    return rec


def unpack_records(rec: np.ndarray, cap: int, kp_dtype) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    hdr = np.ascontiguousarray(rec[:, :8]).view(np.int32).reshape(-1, 2)
    kps = np.ascontiguousarray(rec[:, 8: 8 + cap * KP_BYTES]).view(kp_dtype).reshape(-1, cap)
    desc = np.ascontiguousarray(rec[:, 8 + cap * KP_BYTES:]).reshape(-1, cap, DESC_BYTES)
    return hdr[:, 0].copy(), hdr[:, 1].copy(), kps, desc


def interleave_gathered(gathered: np.ndarray, n_frames: int, world: int) -> np.ndarray:
    """gathered: [world, slots, R] (rank-major, as all_gather returns) -> [n_frames, R] in frame order."""
    slots = gathered.shape[1]
    out = np.empty((n_frames,) + gathered.shape[2:], gathered.dtype)
    for r in range(world):
        idx = frames_of_rank(n_frames, world, r)
        out[idx] = gathered[r, : len(idx)]
    assert slots == slots_per_rank(n_frames, world)
    return out


def extract_sharded(extract_fn: Callable[[np.ndarray], tuple], frames_of: Callable[[np.ndarray], np.ndarray], n_frames: int,
                    cap: int, kp_dtype, group=None, device: Optional[str] = None):
    """Runs `extract_fn(images[b,H,W]) -> (n, mono, kps, desc)` on this rank's frames and all-gathers the records.

    `frames_of(indices)` materialises the images of the given global frame indices (each rank reads only its own).
    Returns (n[n_frames], mono[n_frames], kps[n_frames, cap], desc[n_frames, cap, 32]) on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = frames_of_rank(n_frames, world, rank)
    slots = slots_per_rank(n_frames, world)
    if len(mine):
        n, mono, kps, desc = extract_fn(frames_of(mine))
    else:
        n = mono = np.zeros(0, np.int32)
        kps = np.zeros((0, cap), kp_dtype)
        desc = np.zeros((0, cap, DESC_BYTES), np.uint8)
    rec = pack_records(n, mono, kps, desc, cap, slots)
    if world == 1:
        gathered = rec[None]
    else:
        t = torch.from_numpy(rec)
        if device is not None:
            t = t.to(device)
        out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)  # rank-major concatenation along dim 0
        gathered = out.cpu().numpy().reshape(world, slots, -1)
    full = interleave_gathered(gathered, n_frames, world)
    return unpack_records(full, cap, kp_dtype)
