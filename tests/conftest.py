import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def ops():
    """The product op surface (HIP). Fails loudly when the library is missing: no fallback."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from neuralmagic_vllm_amd import _custom_ops
    from neuralmagic_vllm_amd import _lib
    _lib.lib()
    return _custom_ops
