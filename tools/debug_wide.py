import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from kernel_cases import run_gru_fwd, N_
from oracle import vame_oracle as vo
H, B, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x, st, Y, hN = run_gru_fwd("cuda", H, B, T)
Yn = N_(Y)
for d, s in enumerate(st):
    out, hn, _ = vo.gru_dir_forward(x, s["h0"], s["W_ih"], s["W_hh"], s["b_ih"], s["b_hh"], reverse=bool(d))
    err = np.abs(Yn[:, 1:T + 1, d * H:(d + 1) * H] - out)
    print("dir", d, "max err per t", err.max((0, 2)))
    print("   per 32-col block", err.reshape(B, T, H // 32, 32).max((0, 1, 3)).round(4))
    print("   per row (first 40)", err.max((1, 2))[:40].round(4))
