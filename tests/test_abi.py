"""The C-ABI shared library loads without a GPU and exports every symbol include/vame_hip.h declares
(no compute calls here); the argument validation / error-code path is exercised through the emulator build."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "vame_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vame_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_python_binding():
    from vame_amd import _lib
    assert sorted(_lib.EXPORTS) == header_symbols()


def test_product_library_exports_declared_abi():
    subprocess.run(["make", "-s", "vame_amd/libvame_hip.so"], cwd=ROOT, check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "vame_amd", "libvame_hip.so"))
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.vame_version() >= 100


def test_error_codes_and_messages(emu):
    L = emu.lib()
    x = torch.zeros(16)
    rc = L.vame_gemm_f32(0, 4, 4, x.data_ptr(), 4, 0, 0, 0, x.data_ptr(), 4, 0, 0, 0, None, x.data_ptr(), 4, 0, 1, None, 0, 0, None)
    assert rc == -2 and b"empty problem" in L.vame_last_error()
    rc = L.vame_gemm_f32(4, 4, 4, x.data_ptr(), 4, 1, 0, 0, x.data_ptr(), 4, 0, 0, 0, None, x.data_ptr(), 4, 0, 1, None, 0, 0, None)
    assert rc == -4                                            # A k-major with B n-major: unsupported layout
    rc = L.vame_gru_seq_fwd_f32(None, 1, 4, 32, None)
    assert rc == -1
    d = torch.zeros(16, dtype=torch.int64)
    rc = L.vame_gru_seq_fwd_f32(d.data_ptr(), 1, 4, 48, None)
    assert rc in (-1, -4)
    rc = L.vame_window_gather_f32(x.data_ptr(), 4, 2, None, 0, 1, 8, x.data_ptr(), None)
    assert rc == -2 and b"bad shape" in L.vame_last_error()
    with pytest.raises(emu.VameHipError):
        emu.check(rc, "vame_window_gather_f32")
    # cooperative GRU kernels: refuse what cannot be resident at once / bad row ranges / unsupported hidden sizes
    assert L.vame_gru_coop_supported(2, 256, 256) == 1 and L.vame_gru_coop_supported(2, 512, 256) == 1
    assert L.vame_gru_coop_supported(2, 513, 256) == 0 and L.vame_gru_coop_supported(4, 257, 256) == 0
    assert L.vame_gru_coop_supported(2, 64, 64) == 0 and L.vame_gru_coop_supported(2, 1024, 128) == 1
    flags = torch.zeros(64, dtype=torch.int32)
    epoch = torch.tensor([1, 0], dtype=torch.int32)
    rc = L.vame_gru_coop_fwd_f32(d.data_ptr(), 2, 4096, 256, 0, 0, flags.data_ptr(), 64, epoch.data_ptr(), flags.data_ptr(), None)
    assert rc == -4 and b"one workgroup per CU" in L.vame_last_error()
    rc = L.vame_gru_coop_fwd_f32(d.data_ptr(), 1, 64, 256, 16, 32, flags.data_ptr(), 64, epoch.data_ptr(), flags.data_ptr(), None)
    assert rc == -2 and b"row range" in L.vame_last_error()
    # the launch checks the caller's flag / hand-off buffer against what it will write (a C caller sizing it the old way: refused, not overrun)
    rc = L.vame_gru_coop_fwd_f32(d.data_ptr(), 1, 64, 256, 0, 0, flags.data_ptr(), 64, epoch.data_ptr(), flags.data_ptr(), None)
    assert rc == -2 and b"flags holds 64 ints" in L.vame_last_error()
    # training-set preparation: shape checks
    xd = torch.zeros(16, dtype=torch.float64)
    rc = L.vame_prep_savgol_f64(xd.data_ptr(), 1, 3, 3, xd.data_ptr(), 5, xd.data_ptr() + 64, 3, None)
    assert rc == -2 and b"odd window" in L.vame_last_error()
    rc = L.vame_prep_savgol_f64(xd.data_ptr(), 1, 8, 8, xd.data_ptr(), 5, xd.data_ptr(), 8, None)
    assert rc == -1                                            # in-place filtering is refused


def test_no_cpu_fallback_without_library(monkeypatch):
    """The product binding refuses to run without libvame_hip.so instead of falling back."""
    from vame_amd import _lib
    saved = _lib._lib
    _lib._lib = None
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "vame_amd", "does_not_exist.so"))
    try:
        with pytest.raises(_lib.VameHipError):
            _lib.lib()
    finally:
        _lib._lib = saved


def test_cpu_tensors_rejected_by_product_path():
    """Without the test harness the product refuses CPU tensors and refuses to start without a GPU (no CPU fallback)."""
    import harness
    from vame_amd import _lib, ops
    was_installed = harness._saved is not None            # (this module's other tests run on the emulator harness)
    harness.uninstall()
    try:
        with pytest.raises(_lib.VameHipError):
            ops.axpy(torch.zeros(4), 1.0, torch.zeros(4), 4)
        if not torch.cuda.is_available():
            with pytest.raises(_lib.VameHipError):         # the drivers (train_model, pose_segmentation, bench.py) start here
                _lib.device()
    finally:
        _lib._lib = None
        if was_installed:
            harness.install()


def test_bench_quotes_pmc_traffic_only_for_the_same_kernel_sources(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a committed rocprofv3 summary only if that summary carries the source id compiled into
    the library that is running (vame_source_id()); another build's number is not quoted."""
    import importlib
    import json
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    key = "gemm_kernel TN M=768 N=256 K=122880 x6 grouped"
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "lib_source_id", lambda: "abc123")
    (prof / "r09_pmc_hbm_traffic.json").write_text(json.dumps({"source_id": "other", "by_bench_key": {key: {"hbm_bytes_per_launch_corrected": 1}}}))
    assert bench.pmc_traffic(key) is None
    (prof / "r10_pmc_hbm_traffic.json").write_text(json.dumps({"source_id": "abc123", "by_bench_key": {key: {"hbm_bytes_per_launch_corrected": 42}}}))
    assert bench.pmc_traffic(key) == 42
    assert bench.pmc_traffic("some other kernel") is None
    monkeypatch.setattr(bench, "lib_source_id", lambda: "unidentified")
    assert bench.pmc_traffic(key) is None
