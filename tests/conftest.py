import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # before the first HIP call (iterseg_amd/__init__.py has the reason)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    # 8 OpenMP threads on this shared 8-core box collapse into spin contention (a tiny U-Net
    # forward goes from 0.05 s to 2 s); the oracle's problems here are small.
    try:
        import torch
        torch.set_num_threads(min(4, os.cpu_count() or 1))
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture
def flood_knobs():
    """Sets ISX_* environment knobs for one test: the library reads them once per context (isx_ctx_create), so the process-wide
    watershed context is told to read them again (isx_ctx_reload_knobs) -- and once more after they are restored."""
    from iterseg_amd import watershed as ws
    saved = {}

    def apply(env):
        for k, v in env.items():
            saved.setdefault(k, os.environ.get(k))
            os.environ[k] = v
        ws.get_context().reload_knobs()
    yield apply
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    ws.get_context().reload_knobs()
