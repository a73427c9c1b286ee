#!/usr/bin/env python3
"""Gaussian HMM (next row N1): seconds per EM iteration and per Viterbi decode on the MI355X for N latents of dimension 30, K = 15
states (the reference's defaults), plus the numpy restatement of hmmlearn's algorithm (oracle/hmm_oracle.py, pure-python time loops)
on a small sample for scale.  usage: python tools/hmm_bench.py [N]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from kernel_cases import _hmm_data
from vame_amd.analysis.hmm_hip import GaussianHMMHIP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K, D = 15, 30
rng = np.random.default_rng(0)
X, _ = _hmm_data(rng, N, K, D)
means0 = X[rng.choice(N, K, replace=False)].astype(np.float64)
m = GaussianHMMHIP(K, n_iter=3).fit(X[:50000], means=means0)            # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
m = GaussianHMMHIP(K, n_iter=10, tol=-1e30).fit(X, means=means0)
torch.cuda.synchronize()
t_fit = time.perf_counter() - t0
t0 = time.perf_counter()
lp, path = m.decode(X)
torch.cuda.synchronize()
t_dec = time.perf_counter() - t0
from oracle.hmm_oracle import GaussianHMMOracle
n_cpu = 20000
ref = GaussianHMMOracle(K, n_iter=1)
ref.init_params(X[:n_cpu].astype(np.float64), means0)
t0 = time.perf_counter()
ref.e_step(X[:n_cpu].astype(np.float64))
t_cpu = time.perf_counter() - t0
print(json.dumps(dict(N=N, K=K, D=D, em_iterations=len(m.history_), seconds_per_em_iteration=round(t_fit / len(m.history_), 4),
                      frames_per_s_em=round(N * len(m.history_) / t_fit), viterbi_seconds=round(t_dec, 4), loglik_first_last=[m.history_[0], m.history_[-1]],
                      oracle_numpy_e_step=dict(frames=n_cpu, seconds=round(t_cpu, 3), frames_per_s=round(n_cpu / t_cpu)),
                      note="includes the host M-step and the K Cholesky factorisations per iteration")))
