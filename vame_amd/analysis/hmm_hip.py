"""Gaussian HMM parameterisation of the latents on the MI355X (reference: hmmlearn's
`GaussianHMM(n_components, covariance_type="full", n_iter=100).fit(X)` / `.predict(X)` at
vame/analysis/pose_segmentation.py:145-158).

hmmlearn's Baum-Welch (version 0.2.8 as published: uniform start / transition probabilities, k-means means, data covariance +
min_covar, EM until the log-likelihood gain drops below tol = 1e-2 or n_iter, scalar covariance prior 1e-2) with the per-frame
work -- emission densities, forward / backward recursions, sufficient statistics, Viterbi -- as float64 HIP kernels
(vame_amd/csrc/hmm.hip: chunk-parallel scans) and the M-step (K small matrices) on the host in numpy.  The object pickles like
the reference's (`results/hmm_trained.pkl`) and answers `predict`.  Restatement checked against oracle/hmm_oracle.py; hmmlearn itself
is not available in this image, so parity with the library is unpinned.
"""
import numpy as np
import torch

from .. import _lib, ops


def _ptr(t):
    return ops.addr(t) or None


class GaussianHMMHIP:
    def __init__(self, n_components, covariance_type="full", n_iter=100, tol=1e-2, min_covar=1e-3, startprob_prior=1.0, transmat_prior=1.0,
                 means_prior=0.0, means_weight=0.0, covars_prior=1e-2, covars_weight=1.0, random_state=None, chunk=None):
        if covariance_type != "full":
            raise ValueError("GaussianHMMHIP implements covariance_type='full' (what the reference uses)")
        if not 1 <= n_components <= 32:
            raise ValueError("GaussianHMMHIP supports 1..32 states")
        self.n_components, self.covariance_type, self.n_iter, self.tol, self.min_covar = n_components, covariance_type, n_iter, tol, min_covar
        self.startprob_prior, self.transmat_prior = startprob_prior, transmat_prior
        self.means_prior, self.means_weight, self.covars_prior, self.covars_weight = means_prior, means_weight, covars_prior, covars_weight
        self.random_state, self.chunk = random_state, chunk      # frames per chunk of the parallel scans (None: by sequence length)
        self.history_ = []

    # ------------------------------------------------------------------ device plumbing
    def _chunk(self, N):
        # the chunk chain is sequential (one workgroup): keep it to a few thousand links, and every chunk at least 64 frames
        if self.chunk:
            return int(self.chunk)
        L = 64
        while L < 1024 and N > 2048 * L:
            L *= 2
        return L

    def _upload(self, X):
        Xh = np.ascontiguousarray(X, dtype=np.float32)
        if Xh.ndim != 2 or Xh.shape[1] > 64:
            raise ValueError("GaussianHMMHIP: X must be (n_samples, n_features <= 64)")
        return torch.from_numpy(Xh).to(_lib.device())

    def _buffers(self, N, D, dev):
        K, L = self.n_components, self._chunk(N)
        f64 = dict(dtype=torch.float64, device=dev)
        L_ = _lib.lib()
        return dict(logB=torch.empty(N * K, **f64), bexp=torch.empty(N * K, **f64), rowmax=torch.empty(N, **f64), alpha=torch.empty(N * K, **f64),
                    cnorm=torch.empty(N, **f64), gamma=torch.empty(N * K, **f64), R=torch.empty(N * K, **f64),
                    ws=torch.empty(int(L_.vame_hmm_ws_doubles(N, K, L)), **f64), stats=torch.empty(int(L_.vame_hmm_stats_doubles(K, D)), **f64),
                    sws=torch.empty(int(L_.vame_hmm_stats_ws_doubles(K, D)), **f64))

    def _emission(self, Xd, b):
        """Densities of every frame under the current means_ / covars_ (hmmlearn/stats.py: Cholesky solve) -> logB, bexp, rowmax."""
        N, D = Xd.shape
        K = self.n_components
        linv, logconst = np.empty((K, D, D)), np.empty(K)
        for c in range(K):
            try:
                chol = np.linalg.cholesky(self.covars_[c])
            except np.linalg.LinAlgError:
                chol = np.linalg.cholesky(self.covars_[c] + 1e-7 * np.eye(D))          # (stats.py retries with min_covar = 1e-7)
            linv[c] = np.linalg.inv(chol)
            logconst[c] = -0.5 * (D * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diagonal(chol))))
        dev = Xd.device
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
        mean_d, linv_d, lc_d = to(self.means_), to(linv), to(logconst)
        rc = _lib.lib().vame_hmm_emission_f64(_ptr(Xd), N, D, _ptr(mean_d), _ptr(linv_d), _ptr(lc_d), K, _ptr(b["logB"]), _ptr(b["bexp"]),
                                              _ptr(b["rowmax"]), ops._stream())
        _lib.check(rc, "vame_hmm_emission_f64")

    def _e_step(self, Xd, b):
        """Forward / backward / statistics on the device; returns (log-likelihood, stats dict on the host)."""
        N, D = Xd.shape
        K, L = self.n_components, self._chunk(N)
        dev = Xd.device
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
        sp_d, tm_d = to(self.startprob_), to(self.transmat_)
        lib, st = _lib.lib(), ops._stream()
        self._emission(Xd, b)
        _lib.check(lib.vame_hmm_forward_f64(_ptr(b["bexp"]), N, K, _ptr(sp_d), _ptr(tm_d), L, _ptr(b["alpha"]), _ptr(b["cnorm"]), _ptr(b["ws"]), st),
                   "vame_hmm_forward_f64")
        _lib.check(lib.vame_hmm_backward_f64(_ptr(b["bexp"]), N, K, _ptr(tm_d), _ptr(b["alpha"]), L, _ptr(b["gamma"]), _ptr(b["R"]), _ptr(b["ws"]), st),
                   "vame_hmm_backward_f64")
        _lib.check(lib.vame_hmm_stats_f64(_ptr(Xd), N, D, K, _ptr(b["alpha"]), _ptr(b["gamma"]), _ptr(b["R"]), _ptr(b["cnorm"]), _ptr(b["rowmax"]),
                                          _ptr(b["stats"]), _ptr(b["sws"]), st), "vame_hmm_stats_f64")
        s = b["stats"].cpu().numpy()
        o = 0
        post = s[o:o + K]; o += K
        start = s[o:o + K]; o += K
        trans = s[o:o + K * K].reshape(K, K) * self.transmat_ if N > 1 else np.zeros((K, K)); o += K * K
        obs = s[o:o + K * D].reshape(K, D); o += K * D
        loglik = float(s[o]); o += 1
        obsobs = s[o:o + K * D * D].reshape(K, D, D)
        return loglik, dict(post=post, start=start, trans=trans, obs=obs, obsobs=obsobs)

    # ------------------------------------------------------------------ hmmlearn's host-side steps
    def _init_params(self, X, means=None):
        K, D = self.n_components, X.shape[1]
        self.startprob_ = np.full(K, 1.0 / K)
        self.transmat_ = np.full((K, K), 1.0 / K)
        if means is None:                                  # GaussianHMM._init: k-means cluster centres (here: the GPU k-means of this package)
            from .kmeans_hip import KMeansHIP
            means = KMeansHIP(K, n_init=10, random_state=self.random_state).fit(X).cluster_centers_
        self.means_ = np.array(means, dtype=np.float64)
        cv = np.cov(np.asarray(X, dtype=np.float64).T) + self.min_covar * np.eye(D)
        self.covars_ = np.tile(np.atleast_2d(cv)[None], (K, 1, 1))

    def _m_step(self, stats):
        """`_BaseHMM._do_mstep` + `GaussianHMM._do_mstep` (covariance_type "full"), hmmlearn 0.2.8."""
        sp = np.maximum(self.startprob_prior - 1 + stats["start"], 0)
        sp = np.where(self.startprob_ == 0, 0, sp)
        self.startprob_ = sp / sp.sum()
        tm = np.maximum(self.transmat_prior - 1 + stats["trans"], 0)
        tm = np.where(self.transmat_ == 0, 0, tm)
        rs = tm.sum(1, keepdims=True)
        rs[rs == 0] = 1
        self.transmat_ = tm / rs
        denom = stats["post"][:, None]
        self.means_ = (self.means_weight * self.means_prior + stats["obs"]) / (self.means_weight + denom)
        K, D = self.means_.shape
        meandiff = self.means_ - self.means_prior
        cv_num = np.empty((K, D, D))
        for c in range(K):
            obsmean = np.outer(stats["obs"][c], self.means_[c])
            cv_num[c] = (self.means_weight * np.outer(meandiff[c], meandiff[c]) + stats["obsobs"][c] - obsmean - obsmean.T
                         + np.outer(self.means_[c], self.means_[c]) * stats["post"][c])
        cvweight = max(self.covars_weight - D, 0)
        self.covars_ = (self.covars_prior + cv_num) / (cvweight + stats["post"][:, None, None])

    # ------------------------------------------------------------------ API
    def fit(self, X, lengths=None, means=None):
        if lengths is not None:
            raise NotImplementedError("GaussianHMMHIP.fit: one sequence (the reference concatenates all files into one)")
        X = np.asarray(X)
        Xd = self._upload(X)
        self._init_params(X, means)
        b = self._buffers(Xd.shape[0], Xd.shape[1], Xd.device)
        self.history_ = []
        for _ in range(self.n_iter):
            loglik, stats = self._e_step(Xd, b)
            self._m_step(stats)
            self.history_.append(loglik)
            if len(self.history_) >= 2 and self.history_[-1] - self.history_[-2] < self.tol:       # ConvergenceMonitor.converged
                break
        return self

    def decode(self, X):
        Xd = self._upload(X)
        N, D = Xd.shape
        K, L = self.n_components, self._chunk(N)
        dev = Xd.device
        f64 = dict(dtype=torch.float64, device=dev)
        lib = _lib.lib()
        b = dict(logB=torch.empty(N * K, **f64), bexp=torch.empty(N * K, **f64), rowmax=torch.empty(N, **f64))
        self._emission(Xd, b)
        with np.errstate(divide="ignore"):
            to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
            lsp, ltm = to(np.log(self.startprob_)), to(np.log(self.transmat_))
        path = torch.empty(N, dtype=torch.int32, device=dev)
        logprob = torch.empty(1, **f64)
        ws = torch.empty(int(lib.vame_hmm_ws_doubles(N, K, L)), **f64)
        bws = torch.empty(int(lib.vame_hmm_viterbi_ws_bytes(N, K, L)), dtype=torch.uint8, device=dev)
        _lib.check(lib.vame_hmm_viterbi_f64(_ptr(b["logB"]), N, K, _ptr(lsp), _ptr(ltm), L, path.data_ptr(), _ptr(logprob), _ptr(ws), bws.data_ptr(),
                                            ops._stream()), "vame_hmm_viterbi_f64")
        return float(logprob.cpu()[0]), path.cpu().numpy()

    def predict(self, X, lengths=None):
        return self.decode(X)[1]

    def score(self, X):
        Xd = self._upload(X)
        b = self._buffers(Xd.shape[0], Xd.shape[1], Xd.device)
        return self._e_step(Xd, b)[0]
