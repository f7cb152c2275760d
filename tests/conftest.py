import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _prebuild():
    """Everything the CPU tier builds on first use - the product library (hipcc cross-compiles), the plain-C oracle, the host build of the
    kernels - once, here, before the worker processes of a parallel run start (they would otherwise race for the same object files)."""
    import kornia_amd.build as _product
    import oracle as _oracle

    _product.build()
    _oracle.build()
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu

        build_emu.build()


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """No HIP device (the build container): the CPU tier - ~1 400 tests, most of them the shipped kernels on the host - runs on several
    worker processes when pytest-xdist is importable: ~4 minutes instead of ~10 on 8 host threads.  `-n ...` on the command line,
    KM_TESTS_SERIAL=1, --pdb or --collect-only keep pytest's own behaviour; with a HIP device (`-m gpu` on the GPU box) nothing changes:
    one process owns the device."""
    opt = config.option
    if (hasattr(config, "workerinput") or os.environ.get("KM_TESTS_SERIAL", "") not in ("", "0") or torch.cuda.is_available()
            or not hasattr(opt, "numprocesses") or opt.numprocesses is not None or getattr(opt, "dist", "no") != "no"
            or getattr(opt, "usepdb", False) or getattr(opt, "collectonly", False) or getattr(opt, "help", False) or getattr(opt, "version", 0)):
        return None
    workers = min(6, (os.cpu_count() or 1) - 1)
    if workers < 2 or not any(os.path.isdir(str(a).split("::")[0]) for a in config.args):  # (single files: not worth the workers' start-up)
        return None
    try:
        _prebuild()
    except Exception as e:  # (the serial run reports a failing build through the test that needs it)
        print(f"[conftest] prebuild failed ({type(e).__name__}: {e}); running serially", file=sys.stderr)
        return None
    opt.numprocesses = workers  # (pytest-xdist's own pytest_cmdline_main, next in line, turns this into worker processes)
    opt.dist = "load"
    return None


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
