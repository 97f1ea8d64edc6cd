"""pytest configuration: `gpu` marker, shared fixtures (package, oracle), native builds on demand."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

os.environ.setdefault("ORBX_STRICT_ENV", "1")   # a mistyped ORBX_* switch in a test must fail, not silently run the default


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    so = os.path.join(graft.PKG_DIR, "liborbx.so")
    oso = os.path.join(ROOT, "oracle", "liborb_oracle.so")
    if not (os.path.exists(so) and os.path.exists(oso)):
        graft.build()
    return True


@pytest.fixture(scope="session")
def pkg(built):
    return graft.load_package()


@pytest.fixture(scope="session")
def orc(built):
    return graft.load_oracle()


@pytest.fixture(scope="session")
def gpu_available(pkg):
    return pkg.device_count() > 0
