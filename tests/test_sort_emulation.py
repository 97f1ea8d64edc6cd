"""The libstdc++ std::sort emulation used by the quadtree kernel must reproduce the real std::sort permutation,
ties included (reference src/ORBextractor.cc:569-582, 708)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulation_matches_libstdcxx(tmp_path):
    exe = tmp_path / "test_sort"
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "test_sort_emulation.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode()
    assert out.startswith("OK"), out


def test_oracle_sort_probe_is_a_sorted_permutation(orc):
    rng = np.random.default_rng(0)
    cnt = rng.integers(2, 6, 500).astype(np.int32)
    ulx = (rng.integers(0, 4, 500) * 76).astype(np.int32)
    order = orc.sort_pairs(cnt, ulx)
    assert sorted(order.tolist()) == list(range(500))
    keys = list(zip(cnt[order].tolist(), ulx[order].tolist()))
    assert keys == sorted(keys)
    # unstable: at least one tie group is NOT in original index order (this is what makes emulation necessary)
    unstable = any(order[i] > order[i + 1] for i in range(499) if keys[i] == keys[i + 1])
    assert unstable
