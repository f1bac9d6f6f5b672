import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def api():
    """The C-ABI library.  Built in-tree by __graft_entry__.build(); if missing it is built here (hipcc
    cross-compiles without a GPU)."""
    from lsc_dr_planner_amd import api as A
    from lsc_dr_planner_amd import build as B

    if not os.path.exists(A.LIB_PATH):
        B.build()
    A.lib()
    return A


@pytest.fixture(scope="session")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch
