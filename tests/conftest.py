import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libshimmer_hip.so — host-side entry points work without a GPU; render entry points need one."""
    from shimmer_amd import abi
    if not abi.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
    return abi.load_library()


@pytest.fixture(scope="session")
def orc():
    import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def golden():
    import json
    return json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())


@pytest.fixture(scope="session")
def gpu_lib(lib):
    if lib.shm_device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box (there is no CPU fallback)")
    return lib
