import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the checker libraries (C restatement; _ref only where /root/reference exists)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-j8"], check=True,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def root():
    return ROOT


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture
def mpc_factory():
    """GPU solver factory; HIP extension must be present (no fallback).  Function-scoped: a handle owns every device
    pool its horizon can reach from qmpc_setup on (hundreds of MB at long horizons) -- freed when the test ends."""
    from quadruped_ctrl_amd.binding import BatchedConvexMPC

    made = []

    def make(b, max_batch=None):
        m = BatchedConvexMPC(0, max_batch=max_batch or max(int(b["batch"]), 1), max_horizon=max(16, int(b["horizon"])))
        m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
        made.append(m)
        return m

    yield make
    for m in made:
        m.close()
