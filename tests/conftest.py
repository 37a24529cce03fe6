import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make_oracle():
    so = os.path.join(ROOT, "oracle", "libbzx_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return so


@pytest.fixture(scope="session")
def oracle():
    _make_oracle()
    from bzx_ctypes import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def bzx():
    """The product library on cuda:0.  No fallback: a missing .so or device is a hard failure."""
    # torch bundles its own HIP runtime: initialise it first (as bench.py does) so that tests which hand torch
    # device tensors to the C ABI share one runtime with libbzx.so
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from bzx_ctypes import BzxLib
    lib = BzxLib()
    yield lib
    lib.close()
