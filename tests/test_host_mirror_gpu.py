"""Builds and runs the C++ host-mirror parity test (tests/cpp/test_host_mirror.cpp) on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = tmp_path / "test_host_mirror"
    pkg = os.path.join(ROOT, "wut-cuda-orb-slam3_amd")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-o", str(exe),
           "-L" + pkg, "-lorbx", "-L" + os.path.join(ROOT, "oracle"), "-lorb_oracle",
           "-Wl,-rpath," + pkg, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_host_mirror_compiles(built, tmp_path):
    """CPU: the C++ mirror compiles and links against the C ABI (no compute call)."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_host_mirror_parity(built, tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and "OK host mirror" in out.stdout, out.stdout
