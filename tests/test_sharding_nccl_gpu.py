"""N > 1 on real GPUs: two ranks, one MI355X each, the HIP extractor per rank and the RCCL gather of the result records (backend
"nccl"), against one rank's result.  Skips on a box with fewer than two devices (the round's 1-GPU box); the same functions run over
gloo in tests/test_sharding_gloo.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import importlib

    import torch
    import torch.distributed as dist

    import __graft_entry__ as g
    pkg = g.load_package()
    sharding = importlib.import_module("wut_cuda_orb_slam3_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=rank, max_width=640, max_height=480, max_batch=max(1, (n_frames + world - 1) // world))
    dev = f"cuda:{rank}"

    def extract(imgs):
        if len(imgs) == 0:
            return {"kps": torch.zeros((0, ex.capacity * 28), dtype=torch.uint8, device=dev), "desc": torch.zeros((0, ex.capacity * 32), dtype=torch.uint8, device=dev),
                    "n": torch.zeros(0, dtype=torch.int32, device=dev), "mono": torch.zeros(0, dtype=torch.int32, device=dev)}
        d = torch.from_numpy(imgs).to(dev)
        ex.extract_batch_device(d.data_ptr(), len(imgs), 640, 480, 640, 640 * 480, (0, 1000))
        ex.sync()
        return sharding.device_output_tensors(ex, len(imgs), dev)

    def frames_of(idx):
        return np.stack([pkg.synth_frame(int(i), 640, 480) for i in idx]) if len(idx) else np.zeros((0, 480, 640), np.uint8)

    rec = sharding.extract_sharded(extract, frames_of, n_frames)
    n, mono, kps, desc = sharding.records_to_numpy(rec, ex.capacity, pkg.KEYPOINT_DTYPE)
    if rank == 0:
        q.put((n, mono, kps.tobytes(), desc.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 2])
def test_extract_sharded_nccl_two_ranks(pkg, n_frames):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (one per rank): RCCL does not share a device between ranks")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    [p.start() for p in procs]
    res = q.get(timeout=300)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    n, mono, kb, db = res
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=n_frames)
    imgs = np.stack([pkg.synth_frame(i, 640, 480) for i in range(n_frames)])
    rn, rmono, rk, rd = ex.extract_batch(imgs, (0, 1000))
    assert np.array_equal(n, rn) and np.array_equal(mono, rmono)
    kps = np.frombuffer(kb, pkg.KEYPOINT_DTYPE).reshape(n_frames, -1)
    desc = np.frombuffer(db, np.uint8).reshape(n_frames, -1, 32)
    for f in range(n_frames):
        assert kps[f][: n[f]].tobytes() == rk[f][: n[f]].tobytes() and desc[f][: n[f]].tobytes() == rd[f][: n[f]].tobytes()
