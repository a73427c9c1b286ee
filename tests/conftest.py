import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_weights(g, prefix="w/"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden


sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def _reset_lib():
    import harness
    from vame_amd import _lib
    harness.uninstall()
    _lib._lib = None


@pytest.fixture(scope="module")
def emu():
    """Run vame_amd on the host-emulated build of the kernel sources (tests/emu/harness.py) for this module."""
    import harness
    harness.build()
    lib = harness.install()
    yield lib
    _reset_lib()


@pytest.fixture(scope="module")
def hip():
    """The real gfx950 library on a GPU box."""
    import torch
    from vame_amd import _lib
    _reset_lib()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    if not os.path.exists(_lib.LIB_PATH):                      # normally shipped by __graft_entry__.build(); hipcc is on the box too
        import subprocess
        subprocess.run(["make", "-s", "vame_amd/libvame_hip.so"], cwd=ROOT, check=True)
    _lib.lib()
    yield _lib
