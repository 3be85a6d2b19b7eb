import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hostbuild"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """Every GPU test starts with the LDS of every CU full of NaNs (mi_debug_poison_lds).  LDS keeps what the last kernel on a CU left in it:
    a step kernel that reads a slot before writing it depends on which test ran before -- that is how the Humanoid's limb-wave kernel
    produced a non-finite sensor torque once in ~40 full-suite runs (an unused self-contact slot's garbage times a zero force).  With the
    poison such a read is a NaN in every run."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    try:
        import ctypes as C
        import torch
        from isaacgymenvs_amd import native
        if torch.cuda.is_available():
            L = native.lib()
            L.mi_debug_poison_lds.argtypes = [C.c_uint, C.c_void_p]
            L.mi_debug_poison_lds(0x7FC00000, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    except Exception:       # noqa: BLE001 -- the poison is an aid; a test must not fail on it
        pass
    yield
