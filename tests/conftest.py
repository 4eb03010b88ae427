import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """``gpu``-marked tests are skipped (not failed) where they cannot run: no MI355X, or libqv2x.so not built."""
    import torch
    reason = None
    if not torch.cuda.is_available():
        reason = "no GPU (torch.cuda.is_available() is False)"
    elif not os.path.exists(os.path.join(ROOT, "quantv2x_amd", "libqv2x.so")):
        reason = "quantv2x_amd/libqv2x.so is not built"
    if reason is None:
        return
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class _G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return _G()
