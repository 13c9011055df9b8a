import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; built on demand (hipcc cross-compiles without a GPU)."""
    from pemp_amd import build, _lib
    build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def exact_eval_variants(monkeypatch):
    """Evaluation convs restricted to the variants that are bit-identical to each other (no split-K for small row counts:
    pemp_amd.ops.EVAL_SPLITK): what the "one episode per step equals the batched step bit for bit" tests state."""
    from pemp_amd import ops
    monkeypatch.setattr(ops, "EVAL_SPLITK", False)
