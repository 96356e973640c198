import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is torch-CPU: on the GPU box's 256 hardware threads the default thread count oversubscribes the convolutions
    # (bench.py's recorded sweep: 16 threads 0.54 images/s, 64 -> 0.26, 128 -> 0.16, 256 -> 0.01) - the oracle-heavy full-size tests
    # run several times faster at 16
    try:
        import torch
        if (os.cpu_count() or 1) > 16:
            torch.set_num_threads(16)
    except Exception:
        pass


@pytest.fixture(scope="session")
def seg():
    import segamd
    return segamd
