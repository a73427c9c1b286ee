"""CPU oracle for the HMM parameterisation of the latents -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference calls `hmmlearn.hmm.GaussianHMM(n_components=n_cluster, covariance_type="full", n_iter=100)`, `.fit(X)` and
`.predict(X)` (vame/analysis/pose_segmentation.py:145-158).  hmmlearn is a third-party dependency that is NOT vendored under
/root/reference and NOT installed in this image (VAME.yaml:30 lists it unpinned), so nothing of it can be run or imported here.
This file restates its published algorithm -- hmmlearn 0.2.8, `hmmlearn/base.py` (`_BaseHMM.fit`, `_do_forward_pass`,
`_do_backward_pass`, `_compute_posteriors`, `_accumulate_sufficient_statistics`, `_do_mstep`, `ConvergenceMonitor`),
`hmmlearn/hmm.py` (`GaussianHMM._init`, `._accumulate_sufficient_statistics`, `._do_mstep` for covariance_type="full"),
`hmmlearn/stats.py` (`_log_multivariate_normal_density_full`) and `hmmlearn/_hmmc.pyx` (`_viterbi`) -- in plain numpy float64,
log domain, exactly as hmmlearn computes it.

PARITY UNPINNED against hmmlearn itself: there is no hmmlearn in the container to generate golden vectors from, and the reference
ships none.  What IS pinned (tests/test_oracle.py::test_hmm_oracle_vs_exhaustive_enumeration): the density against
scipy.stats.multivariate_normal; likelihood, state posteriors, expected transition counts and the Viterbi path against a brute-force
enumeration of all K^N state paths of tiny chains (no recursion shared with this file); the M step against its closed forms with
hmmlearn 0.2.8's published prior defaults.  Not pinned: that those defaults (covars_prior 1e-2, covars_weight 1, tol 1e-2, k-means
initialisation of the means, min_covar 1e-3) are the ones the installed hmmlearn of a user would apply -- they are restated from
the published source, as is the reference's call site (constructor arguments, fit-then-predict on the concatenated latents).
Only tests/ (and tools/hmm_bench.py's CPU leg) import this module.
"""
import numpy as np
from scipy import linalg
from scipy.special import logsumexp


def log_mask_zero(a):
    with np.errstate(divide="ignore"):
        return np.log(a)


def log_mvn_density_full(X, means, covars, min_covar=1e-7):
    """hmmlearn/stats.py `_log_multivariate_normal_density_full`: (N, K) log N(x_t; mu_k, Sigma_k) through a Cholesky solve."""
    n_samples, n_dim = X.shape
    out = np.empty((n_samples, len(means)))
    for c, (mu, cv) in enumerate(zip(means, covars)):
        try:
            chol = linalg.cholesky(cv, lower=True)
        except linalg.LinAlgError:
            chol = linalg.cholesky(cv + min_covar * np.eye(n_dim), lower=True)
        log_det = 2 * np.sum(np.log(np.diagonal(chol)))
        sol = linalg.solve_triangular(chol, (X - mu).T, lower=True).T
        out[:, c] = -0.5 * (np.sum(sol ** 2, axis=1) + n_dim * np.log(2 * np.pi) + log_det)
    return out


def forward_log(log_startprob, log_transmat, framelogprob):
    """`_hmmc._forward`: log alpha (N, K); log-likelihood = logsumexp of the last row."""
    N, K = framelogprob.shape
    fwd = np.empty((N, K))
    fwd[0] = log_startprob + framelogprob[0]
    for t in range(1, N):
        fwd[t] = logsumexp(fwd[t - 1][:, None] + log_transmat, axis=0) + framelogprob[t]
    return logsumexp(fwd[-1]), fwd


def backward_log(log_transmat, framelogprob):
    """`_hmmc._backward`: log beta (N, K)."""
    N, K = framelogprob.shape
    bwd = np.zeros((N, K))
    for t in range(N - 2, -1, -1):
        bwd[t] = logsumexp(log_transmat + (framelogprob[t + 1] + bwd[t + 1])[None, :], axis=1)
    return bwd


def viterbi_log(log_startprob, log_transmat, framelogprob):
    """`_hmmc._viterbi`: most likely state sequence and its log probability (first maximum wins ties, like `_argmax`)."""
    N, K = framelogprob.shape
    lattice = np.empty((N, K))
    lattice[0] = log_startprob + framelogprob[0]
    for t in range(1, N):
        lattice[t] = np.max(lattice[t - 1][:, None] + log_transmat, axis=0) + framelogprob[t]
    path = np.empty(N, dtype=np.int32)
    path[-1] = where = int(np.argmax(lattice[-1]))
    logprob = lattice[-1, where]
    for t in range(N - 2, -1, -1):
        where = int(np.argmax(lattice[t] + log_transmat[:, where]))
        path[t] = where
    return logprob, path


class GaussianHMMOracle:
    """GaussianHMM(covariance_type="full") with hmmlearn 0.2.8's defaults (the reference overrides only n_components, n_iter)."""

    def __init__(self, n_components, n_iter=100, tol=1e-2, min_covar=1e-3, startprob_prior=1.0, transmat_prior=1.0, means_prior=0.0,
                 means_weight=0.0, covars_prior=1e-2, covars_weight=1.0, random_state=None):
        self.n_components, self.n_iter, self.tol, self.min_covar = n_components, n_iter, tol, min_covar
        self.startprob_prior, self.transmat_prior = startprob_prior, transmat_prior
        self.means_prior, self.means_weight, self.covars_prior, self.covars_weight = means_prior, means_weight, covars_prior, covars_weight
        self.random_state = random_state
        self.history = []

    def init_params(self, X, means=None):
        """`GaussianHMM._init` + `_BaseHMM._init`: uniform start / transition probabilities, k-means centres as means (injected
        by the tests so that product and oracle start from the same point), the data covariance + min_covar for every state."""
        K, D = self.n_components, X.shape[1]
        self.startprob_ = np.full(K, 1.0 / K)
        self.transmat_ = np.full((K, K), 1.0 / K)
        if means is None:
            from sklearn.cluster import KMeans
            means = KMeans(n_clusters=K, random_state=self.random_state, n_init=10).fit(X).cluster_centers_
        self.means_ = np.array(means, dtype=np.float64)
        cv = np.cov(X.T) + self.min_covar * np.eye(D)
        self.covars_ = np.tile(cv[None], (K, 1, 1))

    def e_step(self, X):
        """One pass of `_BaseHMM.fit`'s inner loop for a single sequence: log-likelihood and sufficient statistics."""
        K = self.n_components
        logB = log_mvn_density_full(X, self.means_, self.covars_)
        log_T = log_mask_zero(self.transmat_)
        logprob, fwd = forward_log(log_mask_zero(self.startprob_), log_T, logB)
        bwd = backward_log(log_T, logB)
        log_gamma = fwd + bwd
        log_gamma -= logsumexp(log_gamma, axis=1, keepdims=True)
        post = np.exp(log_gamma)
        stats = dict(start=post[0].copy(), post=post.sum(0), obs=post.T @ X, obsobs=np.einsum("ij,ik,il->jkl", post, X, X))
        log_xi_sum = np.full((K, K), -np.inf)
        if X.shape[0] > 1:                                               # `_hmmc._compute_log_xi_sum`
            for t in range(X.shape[0] - 1):
                work = fwd[t][:, None] + log_T + (logB[t + 1] + bwd[t + 1])[None, :] - logprob
                log_xi_sum = np.logaddexp(log_xi_sum, work)
        stats["trans"] = np.exp(log_xi_sum)
        return logprob, stats, post

    def m_step(self, stats):
        """`_BaseHMM._do_mstep` + `GaussianHMM._do_mstep` (covariance_type "full")."""
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
        self.covars_ = (self.covars_prior + cv_num) / (cvweight + stats["post"][:, None, None])     # (scalar prior on every entry: as published)

    def fit(self, X, means=None):
        X = np.asarray(X, dtype=np.float64)
        self.init_params(X, means)
        self.history = []
        for it in range(self.n_iter):
            logprob, stats, _ = self.e_step(X)
            self.m_step(stats)
            self.history.append(logprob)
            # ConvergenceMonitor.converged: iter == n_iter or the last gain < tol
            if len(self.history) >= 2 and self.history[-1] - self.history[-2] < self.tol:
                break
        return self

    def predict(self, X):
        X = np.asarray(X, dtype=np.float64)
        logB = log_mvn_density_full(X, self.means_, self.covars_)
        return viterbi_log(log_mask_zero(self.startprob_), log_mask_zero(self.transmat_), logB)[1]

    def score(self, X):
        logB = log_mvn_density_full(np.asarray(X, dtype=np.float64), self.means_, self.covars_)
        return forward_log(log_mask_zero(self.startprob_), log_mask_zero(self.transmat_), logB)[0]
