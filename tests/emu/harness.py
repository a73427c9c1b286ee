"""Test-suite harness: run the product package on the HOST EMULATION build of the kernel sources (libvame_emu.so).

The product (vame_amd/) knows nothing about the emulator: it binds libvame_hip.so, takes torch's HIP
stream and refuses CPU tensors.  For the CPU test-suite this module substitutes those three host hooks
(library handle, stream handle, device checks) so that the same Python engine drives the same `.hip`
kernel sources compiled for the host (tests/emu/hip_emu.*).  Nothing here is shipped or measured.

Also usable as a launcher for a script under the harness (the 2-rank torchrun plumbing test of bench.py):
    python tests/emu/harness.py <script.py> [args...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU_SO = os.path.join(ROOT, "tests", "emu", "libvame_emu.so")
_saved = None


def install(path=EMU_SO):
    global _saved
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from vame_amd import _lib
    if _saved is None:
        _saved = dict(_lib=_lib._lib, device=_lib.device, require_device_tensor=_lib.require_device_tensor,
                      stream_handle=_lib.stream_handle)
    _lib._lib = _lib._bind(path)
    _lib.device = lambda index=None: torch.device("cpu")
    _lib.require_device_tensor = lambda t: None
    _lib.stream_handle = lambda: None
    return _lib


def uninstall():
    global _saved
    from vame_amd import _lib
    if _saved is not None:
        _lib._lib = None
        _lib.device, _lib.require_device_tensor, _lib.stream_handle = _saved["device"], _saved["require_device_tensor"], _saved["stream_handle"]
        _saved = None


def build():
    import subprocess
    subprocess.run(["make", "-s", "-j8", "tests/emu/libvame_emu.so"], cwd=ROOT, check=True)


if __name__ == "__main__":
    import runpy
    install()
    os.environ["VAME_BENCH_LAUNCHER"] = os.path.abspath(__file__)      # bench.py --gpus N re-launches itself through this wrapper
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")
