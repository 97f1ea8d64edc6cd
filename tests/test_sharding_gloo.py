"""N>1 path on CPU: frame sharding + ordered result gather with torch.distributed (gloo, world_size 2 and 3).
The per-frame extractor is the CPU oracle here (no GPU in this container); on the GPU node the same functions
(sharding.extract_sharded / gather_frame_records, also used by bench.py --gather) run with backend nccl (RCCL) on zero-copy
views of the HIP extractor's HBM output buffers."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import __graft_entry__ as g
    pkg = g.load_package()
    orc = g.load_oracle()
    import importlib
    sharding = importlib.import_module("wut_cuda_orb_slam3_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    ref = orc.Oracle(300, 1.2, 8, 20, 7)
    cap = ref.capacity

    def extract(imgs):  # records as torch tensors, like device_output_tensors() of the HIP extractor (here on the CPU)
        if len(imgs) == 0:
            n = mono = np.zeros(0, np.int32)
            kps = np.zeros((0, cap), orc.KEYPOINT_DTYPE)
            desc = np.zeros((0, cap, 32), np.uint8)
        else:
            n, mono, kps, desc = ref.extract_batch(imgs, (0, 1000), threads=2, want_outputs=True)
        return {"kps": torch.from_numpy(np.ascontiguousarray(kps).view(np.uint8).reshape(len(n), cap * 28)),
                "desc": torch.from_numpy(np.ascontiguousarray(desc).reshape(len(n), cap * 32)),
                "n": torch.from_numpy(np.ascontiguousarray(n, np.int32)), "mono": torch.from_numpy(np.ascontiguousarray(mono, np.int32))}

    def frames_of(idx):
        return np.stack([orc.synth_frame(int(i), 320, 240) for i in idx]) if len(idx) else np.zeros((0, 240, 320), np.uint8)

    rec = sharding.extract_sharded(extract, frames_of, n_frames)
    n, mono, kps, desc = sharding.records_to_numpy(rec, cap, orc.KEYPOINT_DTYPE)
    if rank == 0:
        q.put((n, mono, kps.tobytes(), desc.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 5), (3, 4), (2, 1)])
def test_sharded_extraction_matches_single_process(orc, world, n_frames):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    [p.start() for p in procs]
    import queue as _queue
    res = None
    for _ in range(240):
        try:
            res = q.get(timeout=1)
            break
        except _queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    [p.join(timeout=60) for p in procs]
    assert res is not None, [p.exitcode for p in procs]
    n, mono, kb, db = res
    assert all(p.exitcode == 0 for p in procs)
    ref = orc.Oracle(300, 1.2, 8, 20, 7)
    imgs = np.stack([orc.synth_frame(i, 320, 240) for i in range(n_frames)])
    rn, rmono, rk, rd = ref.extract_batch(imgs, (0, 1000), threads=2, want_outputs=True)
    assert np.array_equal(n, rn) and np.array_equal(mono, rmono)
    assert kb == rk.tobytes() and db == rd.tobytes()  # slot == frame index, independent of rank / completion order


def test_shard_helpers(pkg):
    import importlib
    sh = importlib.import_module("wut_cuda_orb_slam3_amd.sharding")
    assert list(sh.frames_of_rank(10, 4, 1)) == [1, 5, 9]
    assert sh.slots_per_rank(10, 4) == 3 and sh.slots_per_rank(8, 8) == 1
    cover = np.sort(np.concatenate([sh.frames_of_rank(37, 8, r) for r in range(8)]))
    assert np.array_equal(cover, np.arange(37))
    assert sh.record_bytes(1003) == 8 + 1003 * 60
    # single-process gather: padding + transpose back to frame order
    import torch
    local = {"kps": torch.arange(3 * 4, dtype=torch.uint8).reshape(3, 4), "desc": torch.zeros((3, 8), dtype=torch.uint8),
             "n": torch.tensor([5, 6, 7], dtype=torch.int32), "mono": torch.tensor([1, 2, 3], dtype=torch.int32)}
    out = sh.gather_frame_records(local, 3)
    assert out["n"].tolist() == [5, 6, 7] and torch.equal(out["kps"], local["kps"])
