"""Lock-step vs wave-specialised BPTT at H = 128 (interleaved, warmed; production library)."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import kernel_cases as kc
dev = "cuda"
for H, B, T in ((128, 4096, 30), (128, 8192, 30), (128, 1024, 30)):
    x, st, Y, hN = kc.run_gru_fwd(dev, H, B, T, seed=1)
    dY = torch.randn(B, T, 2 * H, device=dev); dhN = torch.randn(B, 2 * H, device=dev)
    def run(ws, n=3):
        os.environ["VAME_GRU_WS"] = ws
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); kc._run_gru_bwd(dev, H, B, T, st, Y, dY, dhN); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        return ts
    for _ in range(8): run("0", 2); run("2", 2)
    a, b = [], []
    for _ in range(6): a += run("0"); b += run("2")
    a.sort(); b.sort()
    print(f"H={H} B={B} T={T}: lock-step {a[len(a)//2]:.0f} us, wave-specialised {b[len(b)//2]:.0f} us")
