"""GPU check of the wave-specialised BPTT kernel against the lock-step one (bit-for-bit) + timing of both (2-stream launch, B=4096, T=30)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import kernel_cases as kc
dev = "cuda"
for H, B, T in ((256, 100, 30), (128, 70, 5), (256, 4096, 30)):
    x, st, Y, hN = kc.run_gru_fwd(dev, H, B, T, seed=1)
    rng = np.random.default_rng(5)
    dY = torch.from_numpy(rng.standard_normal((B, T, 2 * H)).astype(np.float32)).cuda(); dhN = torch.from_numpy(rng.standard_normal((B, 2 * H)).astype(np.float32)).cuda()
    res = {}
    for ws in ("1", "0"):
        os.environ["VAME_GRU_WS"] = ws
        outs = kc._run_gru_bwd(dev, H, B, T, st, Y, dY, dhN)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); kc._run_gru_bwd(dev, H, B, T, st, Y, dY, dhN); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[ws] = (outs, sorted(ts)[2])
    same = all(torch.equal(a, b) for oa, ob in zip(res["1"][0], res["0"][0]) for a, b in zip(oa[:3], ob[:3]))
    fl = 2 * 2.0 * 3 * H * H * B * T
    print(f"H={H} B={B} T={T}: bit-identical={same}  ws {res['1'][1]:.0f} us ({fl/res['1'][1]/1e6:.1f} TF)  lock-step {res['0'][1]:.0f} us ({fl/res['0'][1]/1e6:.1f} TF)  [timings include the python wrapper]")
