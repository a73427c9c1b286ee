"""k-means over the latent vectors on the MI355X -- SURVEY 8(f) row N1 (opt-in: cfg['amd_gpu_kmeans']).

The reference calls scikit-learn (`KMeans(init='k-means++', n_clusters, random_state=42, n_init=20)`,
vame/analysis/pose_segmentation.py:141; per-file variant :179).  This class follows the same algorithm -- greedy
k-means++ seeding, Lloyd iterations on mean-centred data, sklearn's tolerance rule (tol * mean feature variance on the
squared centre shift, or unchanged labels), best of n_init by inertia -- with the E-step as a HIP scan kernel
(`vame_kmeans_assign_f32`) and the M-step as a deterministic split-K MFMA GEMM `onehot^T X` (`vame_gemm_f32`).
It is NOT bit-identical to scikit-learn (different random draws in the seeding, fp32 summation order), which is why the
default path of `pose_segmentation()` stays on scikit-learn; tests compare inertia and partition agreement instead.
"""
import numpy as np
import torch

from .. import _lib, ops


class KMeansHIP:
    def __init__(self, n_clusters, n_init=10, random_state=None, max_iter=300, tol=1e-4):
        self.n_clusters, self.n_init, self.random_state, self.max_iter, self.tol = n_clusters, n_init, random_state, max_iter, tol

    # ------------------------------------------------------------------ helpers
    def _dev(self, X):
        if torch.is_tensor(X):
            t = X
        else:
            t = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32))
        return t.to(device=_lib.device(), dtype=torch.float32).contiguous()

    def _assign(self, X, C, want_onehot):
        N, D = X.shape
        K = C.shape[0]
        Kp = (K + 3) // 4 * 4
        labels = torch.empty(N, dtype=torch.int32, device=X.device)
        mind2 = torch.empty(N, device=X.device)
        onehot = torch.empty(N, Kp, device=X.device) if want_onehot else None
        ops.kmeans_assign(X, N, D, C.contiguous(), K, labels, mind2, onehot, Kp)
        return labels, mind2, onehot

    def _dist2_to(self, X, point):
        return self._assign(X, point.reshape(1, -1), False)[1]

    def _init_pp(self, X, rs):
        """Greedy k-means++ (Arthur & Vassilvitskii; 2 + log K local trials, as scikit-learn's _kmeans_plusplus)."""
        N, D = X.shape
        K = self.n_clusters
        trials = 2 + int(np.log(K))
        centers = torch.empty(K, D, device=X.device)
        centers[0] = X[int(rs.randint(N))]
        closest = self._dist2_to(X, centers[0])
        pot = float(closest.sum(dtype=torch.float64))
        for c in range(1, K):
            rand_vals = torch.from_numpy(rs.uniform(size=trials) * pot).to(X.device)
            cand = torch.searchsorted(torch.cumsum(closest.double(), 0), rand_vals).clamp_(max=N - 1)
            best_pot, best_d, best_i = None, None, None
            for i in cand.tolist():
                d = torch.minimum(closest, self._dist2_to(X, X[i]))
                p = float(d.sum(dtype=torch.float64))
                if best_pot is None or p < best_pot:
                    best_pot, best_d, best_i = p, d, i
            centers[c] = X[best_i]
            closest, pot = best_d, best_pot
        return centers

    def _lloyd(self, X, centers, tol_abs):
        N, D = X.shape
        K = self.n_clusters
        Kp = (K + 3) // 4 * 4
        sums = torch.empty(Kp, D, device=X.device)
        counts = torch.empty(Kp, device=X.device)
        sk = max(1, min(512, N // 2048))
        ws = torch.empty(sk * Kp * D, device=X.device) if sk > 1 else None
        prev = None
        strict = False
        n_iter = 0
        for n_iter in range(1, self.max_iter + 1):
            labels, mind2, onehot = self._assign(X, centers, True)
            ops.gemm(Kp, D, N, ops.Operand(onehot, Kp), 1, ops.Operand(X, D), 1, sums, D, splitk=sk, ws=ws)     # sums = onehot^T X
            ops.colsum(onehot, 0, N, Kp, Kp, counts)
            cnt = counts[:K]
            new_centers = sums[:K] / cnt.clamp(min=1.0)[:, None]
            empty = (cnt == 0).nonzero().flatten()
            if empty.numel():                                   # relocate empty clusters to the points farthest from their centres
                far = torch.topk(mind2, empty.numel()).indices
                new_centers[empty] = X[far]
            if prev is not None and torch.equal(labels, prev):
                strict = True
                centers = new_centers
                break
            shift = float(((new_centers - centers) ** 2).sum())
            centers, prev = new_centers, labels
            if shift <= tol_abs:
                break
        if not strict:
            labels, mind2, _ = self._assign(X, centers, False)
        return labels, float(mind2.sum(dtype=torch.float64)), centers, n_iter

    # ------------------------------------------------------------------ sklearn-like API
    def fit(self, X):
        Xd = self._dev(X)
        mean = Xd.mean(0)
        Xc = Xd - mean
        tol_abs = float(Xc.var(0, unbiased=False).mean()) * self.tol
        rs = self.random_state if isinstance(self.random_state, np.random.RandomState) else np.random.RandomState(self.random_state)
        best = None
        for _ in range(self.n_init):
            centers = self._init_pp(Xc, rs)
            labels, inertia, centers, n_iter = self._lloyd(Xc, centers, tol_abs)
            if best is None or inertia < best[1]:
                best = (labels, inertia, centers, n_iter)
        labels, inertia, centers, n_iter = best
        self.cluster_centers_ = (centers + mean).cpu().numpy()
        self.labels_ = labels.cpu().numpy().astype(np.int32)
        self.inertia_, self.n_iter_ = inertia, n_iter
        return self

    def predict(self, X):
        Xd = self._dev(X)
        C = self._dev(self.cluster_centers_)
        return self._assign(Xd, C, False)[0].cpu().numpy().astype(np.int32)
