import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture
def tune():
    """Sets NMX_* tuning overrides of the loaded library for one test (the library reads the environment only once, at
    load): tune(NMX_GEMM_WIDE="2,2,1"); None clears. Everything touched is cleared again afterwards."""
    from neuralmagic_vllm_amd import _lib
    touched = set()

    def set_(**kw):
        for k, v in kw.items():
            _lib.set_tuning(k, v)
            touched.add(k)

    yield set_
    for k in touched:
        _lib.set_tuning(k, None)


@pytest.fixture(scope="session")
def ops():
    """The product op surface (HIP). Fails loudly when the library is missing: no fallback."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from neuralmagic_vllm_amd import _custom_ops
    from neuralmagic_vllm_amd import _lib
    _lib.lib()
    return _custom_ops
