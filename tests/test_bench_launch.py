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


import json  # noqa: E402

import pytest  # noqa: E402


def _run_bench(extra, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_carries_the_8d_metric_and_the_contract_keys():
    """N = 1, small batch: `value` keeps its definition (HBM-resident) and says so; SURVEY 8(d)'s PCIe-inclusive rate is a top-level key;
    a step passes over several distinct batches; the roofline block is the HBM one."""
    d = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "64", "--batches-per-step", "3", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "value_definition", "metric_8d_frames_per_s", "pcie_inclusive_frames_per_s",
              "pcie_inclusive_page_locked_frames_per_s", "pcie_inclusive_fresh_result_arrays_frames_per_s", "pcie_roofline", "gather_ms"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["gather_ms"] is None
    assert d["config"]["batches_per_step"] == 3 and d["config"]["units_per_gpu_per_step"] == 192
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s" and 0 < d["roofline"]["frac"] < 1
    assert d["metric_8d_frames_per_s"] == d["pcie_inclusive_page_locked_pipelined_frames_per_s"] > 0 and d["pcie_inclusive_page_locked_frames_per_s"] > 0
    assert "d2h_link_gbs" in d["pcie_roofline"]["page_locked"] and d["pcie_roofline"]["fresh_result_arrays"]["frames_per_s"] > 0


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_runs_the_host_leg_on_every_rank():
    """`python bench.py --gpus 2` on a box with fewer devices than ranks: gloo rendezvous, the ranks share the device.  The N > 1 code path
    must time the host-pointer (PCIe-inclusive) leg on EVERY rank and aggregate it; the RCCL gather is not available: gather_ms null."""
    d = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--batches-per-step", "2", "--no-cpu-baseline"],
                   {"ORBX_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["units_per_gpu_per_step"] == 128
    assert d["gather_ms"] is None and "rehearsal" in d["gather_note"] and "rehearsal" in d["config"]
    assert d["metric_8d_frames_per_s"] > 0 and d["pcie_inclusive_frames_per_s"] > 0
    assert "2 ranks at the same time" in d["pcie_inclusive_note"]


@pytest.mark.gpu
def test_bench_gather_inside_the_timed_region_at_world_one():
    """--gather at N = 1 (a world-1 all-gather = the placement copy): the gather code path of the N > 1 run and its timing key."""
    d = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "64", "--batches-per-step", "2", "--no-cpu-baseline", "--no-pcie", "--gather"])
    assert d["config"]["gather"] is True and d["gather_ms"] is not None and d["gather_ms"] >= 0
    assert "gather" in d["value_definition"]
