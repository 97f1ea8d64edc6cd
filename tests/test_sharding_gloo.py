"""N>1 path on CPU: frame sharding + ordered result gather with torch.distributed (gloo, world_size 2 and 3).
The per-frame extractor is the CPU oracle here (no GPU in this container); on the GPU node the same code runs
with backend nccl (RCCL) and the HIP extractor."""
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
    ref = orc.Oracle(300, 1.2, 8, 20, 7)
    cap = ref.capacity

    def extract(imgs):
        n, mono, kps, desc = ref.extract_batch(imgs, (0, 1000), threads=2, want_outputs=True)
        return n, mono, kps, desc

    def frames_of(idx):
        return np.stack([orc.synth_frame(int(i), 320, 240) for i in idx])

    n, mono, kps, desc = sharding.extract_sharded(extract, frames_of, n_frames, cap, orc.KEYPOINT_DTYPE)
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
    cap = 5
    n = np.array([3, 0], np.int32)
    mono = np.array([1, 0], np.int32)
    kps = np.zeros((2, cap), pkg.KEYPOINT_DTYPE)
    kps["x"][0, :3] = [1.5, 2.5, 3.5]
    desc = np.arange(2 * cap * 32, dtype=np.uint8).reshape(2, cap, 32)
    rec = sh.pack_records(n, mono, kps, desc, cap, 3)
    assert rec.shape == (3, sh.record_bytes(cap))
    n2, m2, k2, d2 = sh.unpack_records(rec, cap, pkg.KEYPOINT_DTYPE)
    assert list(n2) == [3, 0, -1] and list(m2) == [1, 0, 0]
    assert np.array_equal(k2[:2], kps) and np.array_equal(d2[:2], desc)
