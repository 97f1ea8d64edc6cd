"""bench.py --gpus N must start its own ranks when called without a launcher (the way the round driver calls it)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_starts_n_ranks_without_touching_the_gpu_in_the_parent():
    """CPU container: no HIP device, so every rank must fail loudly ("needs a HIP device": no CPU fallback) - which shows that
    `python bench.py --gpus 2` spawned torch.distributed.run with two ranks from a parent that never imported torch."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = ""  # (on a GPU box this test still exercises the failure path)
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a HIP device") == 2, r.stderr[-2000:]
    assert '"metric"' not in r.stdout


def test_kernel_source_hash_is_stable():
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_sha256()
    assert len(h) == 64 and h == bench.kernel_source_sha256()
