import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs small matrices: more than a handful of intra-op threads only adds synchronisation cost
    # (on a 256-thread host the oracle tests ran 10x slower with the default thread count)
    try:
        import torch
        torch.set_num_threads(min(8, torch.get_num_threads()))
    except Exception:  # pragma: no cover
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped automatically when no GPU is visible (the CPU tier also passes -m 'not gpu')."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
