import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def ctx():
    """GPU context; the HIP library MUST be loadable here (no fallback)."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from lra_amd.context import Context
    c = Context(0)
    yield c
    c.close()
