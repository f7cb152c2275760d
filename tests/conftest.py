import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device); run with -m gpu on the GPU box")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The plain-C CPU oracle (test infrastructure only)."""
    import oracle as _oracle

    _oracle.build()
    return _oracle


@pytest.fixture(autouse=True)
def _seed_everything():
    """Every test starts from the same RNG state (host and every HIP device): tensors drawn without an explicit generator
    are reproducible, so a tolerance that holds once holds on every run."""
    torch.manual_seed(1234)
    yield
