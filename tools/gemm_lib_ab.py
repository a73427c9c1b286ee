#!/usr/bin/env python3
"""One GEMM shape on several builds of the library, interleaved (HIP-event medians, checked against a float64 product on a corner):
python tools/gemm_lib_ab.py M N K akm bkm lib1.so lib2.so ..."""
import os, subprocess, sys, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if sys.argv[1] == "--child":
    sys.path.insert(0, R)
    lib, M, N, K, akm, bkm = sys.argv[2], *[int(v) for v in sys.argv[3:8]]
    from vame_amd import _lib
    _lib._lib = _lib._bind(os.path.abspath(lib))
    import torch
    from vame_amd import ops
    from vame_amd.ops import Operand
    torch.manual_seed(0)
    A = torch.randn((K, M) if akm else (M, K), device="cuda")
    B = torch.randn((K, N) if bkm else (N, K), device="cuda")
    C = torch.empty(M, N, device="cuda")
    run = lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    a = (A.t() if akm else A)[:64].double(); b = (B if bkm else B.t()).double()
    err = float(((a @ b) - C[:64].double()).abs().max())
    ms = sorted(ts)[len(ts) // 2]
    print(json.dumps({"ms": ms, "tf": 2.0 * M * N * K / ms * 1e-9, "err": err}))
    sys.exit(0)
M, N, K, akm, bkm = sys.argv[1:6]
libs = sys.argv[6:]
for rnd in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, __file__, "--child", lib, M, N, K, akm, bkm], capture_output=True, text=True).stdout
        j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        print(f"{os.path.basename(lib):26s} M={M} N={N} K={K} akm={akm} bkm={bkm}: {j['ms'] * 1e3:8.1f} us  {j['tf']:6.1f} TF  max err {j['err']:.2e}", flush=True)
