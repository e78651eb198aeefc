import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("hts-train-world_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference (oracle/_ref); only present where `make -C oracle ref` has run."""
    from oracle.bindings import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libworld_ref.so not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def gpu(pkg):
    """(torch, world module, Context) on cuda:0 -- the HIP extension is mandatory here."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    W = pkg.world
    W.load_library()
    ctx = W.Context(stream_ptr=torch.cuda.current_stream().cuda_stream)
    yield torch, W, ctx
    ctx.close()
