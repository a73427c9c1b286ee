// Gaussian HMM over the latents on the MI355X (second half): the parameterisation step the reference delegates to
// hmmlearn -- GaussianHMM(n_components, covariance_type="full", n_iter=100).fit(X) / .predict(X), vame/analysis/
// pose_segmentation.py:145-158.  float64 throughout (hmmlearn computes in float64; the work is small next to the RNN: per EM
// iteration N*K*D^2 fused multiply-adds for the second moments and N*K^2 for the recursions).
//
// The forward / backward / Viterbi recursions are sequential in time.  They are parallelised over CHUNKS of L frames in three
// passes: (1) summary -- every chunk propagates the K unit vectors through its frames, i.e. computes its K x K transfer operator
// (one lane per (chunk, unit vector), no communication; rows are re-normalised per frame and carry a log scale); (2) scan -- one
// workgroup chains the chunk operators to get the vector ENTERING every chunk; (3) apply -- every chunk re-runs its frames from the
// entering vector and writes the per-frame results.  The recursions are scale-free (the smoothed posteriors gamma_t and the
// transition responsibilities are normalised per frame), so the lost scale factors never matter; the log-likelihood comes from
// the per-frame normalisers of pass 3.  Emission densities enter as b_t[k] = exp(logB[t,k] - max_k logB[t,k]) <= 1.
#include "vame_common.h"
#include <math.h>

#define HMM_MAXK 32
#define HMM_TPB 256

// ------------------------------------------------------------------------------------------- emission densities
// logB[t,k] = logconst[k] - 0.5 * || Linv_k (x_t - mu_k) ||^2 with Linv_k the inverse of the lower Cholesky factor of Sigma_k
// (hmmlearn/stats.py _log_multivariate_normal_density_full); one thread per frame, one state's Linv (D x D doubles) in LDS at a time.
template <int DP>
__global__ __launch_bounds__(HMM_TPB) void hmm_emission_kernel(const float* __restrict__ X, int64_t N, int D, const double* __restrict__ mean,
                                                               const double* __restrict__ linv, const double* __restrict__ logconst, int K,
                                                               double* __restrict__ logB, double* __restrict__ bexp, double* __restrict__ rowmax) {
    __shared__ double Ls[DP * DP];
    __shared__ double mus[DP];
    const int64_t t = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x;
    const bool live = t < N;
    double x[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = (live && d < D) ? (double)X[t * D + d] : 0.0;
    double best = -INFINITY;
    for (int k = 0; k < K; ++k) {
        __syncthreads();
        for (int i = threadIdx.x; i < D * D; i += HMM_TPB) Ls[(i / D) * DP + i % D] = linv[(int64_t)k * D * D + i];
        if (threadIdx.x < D) mus[threadIdx.x] = mean[k * D + threadIdx.x];
        __syncthreads();
        double maha = 0.0;
        for (int i = 0; i < D; ++i) {
            double y = 0.0;
#pragma unroll
            for (int j = 0; j < DP; ++j)
                if (j <= i && j < D) y += Ls[i * DP + j] * (x[j] - mus[j]);
            maha += y * y;
        }
        const double lp = logconst[k] - 0.5 * maha;
        if (live) logB[t * K + k] = lp;
        best = fmax(best, lp);
    }
    if (live) {
        rowmax[t] = best;
        for (int k = 0; k < K; ++k) bexp[t * K + k] = exp(logB[t * K + k] - best);
    }
}

extern "C" int vame_hmm_emission_f64(const float* X, int64_t N, int D, const double* mean, const double* linv, const double* logconst, int K,
                                     double* logB, double* bexp, double* rowmax, void* stream) {
    VAME_CHECK_ARG(X && mean && linv && logconst && logB && bexp && rowmax, VAME_E_BADARG, "hmm_emission: null pointer");
    VAME_CHECK_ARG(N >= 1 && D >= 1 && D <= 64 && K >= 1 && K <= HMM_MAXK, VAME_E_SHAPE, "hmm_emission: N=%lld D=%d (<= 64) K=%d (<= %d)",
                   (long long)N, D, K, HMM_MAXK);
    const dim3 grid((unsigned)cdiv64(N, HMM_TPB));
    if (D <= 32) hipLaunchKernelGGL(hmm_emission_kernel<32>, grid, dim3(HMM_TPB), 0, (hipStream_t)stream, X, N, D, mean, linv, logconst, K, logB, bexp, rowmax);
    else hipLaunchKernelGGL(hmm_emission_kernel<64>, grid, dim3(HMM_TPB), 0, (hipStream_t)stream, X, N, D, mean, linv, logconst, K, logB, bexp, rowmax);
    VAME_LAUNCH_CHECK("hmm_emission");
    return VAME_OK;
}

// ------------------------------------------------------------------------------------------- chunked recursions
// Semiring of a recursion: SUM = (+, *) on probabilities (forward / backward), MAX = (max, +) on log probabilities (Viterbi).
// Row-vector step (forward, Viterbi):  v'[j] = (+)_l v[l] (*) A[l][j]   then (*) e[j]      (e = b_t resp. logB_t)
// Column-vector step (backward):       u'[i] = (+)_j A[i][j] (*) (e[j] (*) u[j])           (e = b_{t+1})
struct HmmChunking { int64_t N; int K, L; int64_t nchunks; };
__device__ __forceinline__ void chunk_range(const HmmChunking& c, int64_t ch, int64_t& lo, int64_t& hi) {
    // chunk ch owns the transitions INTO frames [lo, hi): frame 0 has no incoming transition
    lo = ch * c.L < 1 ? 1 : ch * c.L;
    hi = (ch + 1) * (int64_t)c.L < c.N ? (ch + 1) * (int64_t)c.L : c.N;
}

template <int KP, bool MAXPLUS, bool BACKWARD>
__device__ __forceinline__ void hmm_step(double (&v)[KP], const double* As, int K, const double* __restrict__ e) {
    double w[KP];
    if (!BACKWARD) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            double s = MAXPLUS ? -INFINITY : 0.0;
            if (j < K) {
#pragma unroll
                for (int l = 0; l < KP; ++l)
                    if (l < K) s = MAXPLUS ? fmax(s, v[l] + As[l * KP + j]) : s + v[l] * As[l * KP + j];
                s = MAXPLUS ? s + e[j] : s * e[j];
            }
            w[j] = s;
        }
    } else {
        double eu[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) eu[j] = j < K ? e[j] * v[j] : 0.0;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            double s = 0.0;
            if (i < K) {
#pragma unroll
                for (int j = 0; j < KP; ++j)
                    if (j < K) s += As[i * KP + j] * eu[j];
            }
            w[i] = s;
        }
    }
#pragma unroll
    for (int j = 0; j < KP; ++j) v[j] = w[j];
}

// (+,*) vectors shrink every frame (b <= 1, rows of A sum to one): divide by the largest entry and keep its log
template <int KP>
__device__ __forceinline__ double renorm(double (&v)[KP], int K) {
    double m = 0.0;
#pragma unroll
    for (int j = 0; j < KP; ++j)
        if (j < K) m = fmax(m, v[j]);
    if (!(m > 0.0)) return -INFINITY;                       // an impossible path: everything stays zero
    const double inv = 1.0 / m;
#pragma unroll
    for (int j = 0; j < KP; ++j) v[j] *= inv;
    return log(m);
}

template <int KP>
__device__ __forceinline__ void load_A(double* As, const double* __restrict__ A, int K) {
    for (int i = threadIdx.x; i < KP * KP; i += blockDim.x) {
        const int r = i / KP, c = i % KP;
        As[i] = (r < K && c < K) ? A[r * K + c] : 0.0;
    }
    __syncthreads();
}

// pass 1: P[ch][i][:] = unit vector i propagated through the chunk (re-normalised), S[ch][i] = its accumulated log scale
template <int KP, bool MAXPLUS, bool BACKWARD>
__global__ __launch_bounds__(HMM_TPB) void hmm_summary_kernel(HmmChunking c, const double* __restrict__ A, const double* __restrict__ E,
                                                              double* __restrict__ P, double* __restrict__ S) {
    __shared__ double As[KP * KP];
    load_A<KP>(As, A, c.K);
    const int64_t g = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x;
    const int64_t ch = g / KP;
    const int i = (int)(g % KP);
    if (ch >= c.nchunks || i >= c.K) return;
    int64_t lo, hi;
    chunk_range(c, ch, lo, hi);
    double v[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) v[j] = MAXPLUS ? (j == i ? 0.0 : -INFINITY) : (j == i ? 1.0 : 0.0);
    double s = 0.0;
    if (!BACKWARD) {
        for (int64_t t = lo; t < hi; ++t) {
            hmm_step<KP, MAXPLUS, false>(v, As, c.K, E + t * c.K);
            if (!MAXPLUS) s += renorm<KP>(v, c.K);
        }
    } else {
        for (int64_t t = hi - 1; t >= lo; --t) {                         // u_{t-1} = A (b_t . u_t)
            hmm_step<KP, false, true>(v, As, c.K, E + t * c.K);
            s += renorm<KP>(v, c.K);
        }
    }
    double* p = P + (ch * KP + i) * KP;
#pragma unroll
    for (int j = 0; j < KP; ++j) p[j] = v[j];
    S[ch * KP + i] = s;
}

// pass 2 (one workgroup of 64 threads): enter[ch] = vector entering chunk ch.  Forward / Viterbi: the state after frame lo-1;
// backward: the vector u at frame hi-1 ... chained through the chunk operators of pass 1.
template <int KP, bool MAXPLUS, bool BACKWARD>
__global__ __launch_bounds__(64) void hmm_scan_kernel(HmmChunking c, const double* __restrict__ init, const double* __restrict__ P,
                                                      const double* __restrict__ S, double* __restrict__ enter) {
    __shared__ double cur[KP], nxt[KP];
    const int j = threadIdx.x;
    if (j < KP) cur[j] = j < c.K ? init[j] : (MAXPLUS ? -INFINITY : 0.0);
    __syncthreads();
    // the chain is latency-bound (one dependent K x K product per chunk): the next chunk's operator column and scales are
    // fetched into registers while the current one is applied
    double pcol[KP], srow[KP];
    auto fetch = [&](int64_t n) {
        const int64_t ch = BACKWARD ? c.nchunks - 1 - n : n;
        const double* p = P + ch * KP * KP;
        const double* s = S + ch * KP;
#pragma unroll
        for (int i = 0; i < KP; ++i) { pcol[i] = (j < KP && i < c.K) ? p[i * KP + j] : 0.0; srow[i] = i < c.K ? s[i] : -INFINITY; }
    };
    fetch(0);
    for (int64_t n = 0; n < c.nchunks; ++n) {
        const int64_t ch = BACKWARD ? c.nchunks - 1 - n : n;
        if (j < KP) enter[ch * KP + j] = cur[j];
        double pc[KP], sr[KP];
#pragma unroll
        for (int i = 0; i < KP; ++i) { pc[i] = pcol[i]; sr[i] = srow[i]; }
        if (n + 1 < c.nchunks) fetch(n + 1);
        double acc = MAXPLUS ? -INFINITY : 0.0;
        if (j < c.K) {
            if (MAXPLUS) {
#pragma unroll
                for (int i = 0; i < KP; ++i)
                    if (i < c.K) acc = fmax(acc, cur[i] + pc[i]);
            } else {
                // forward: row vector times operator; backward: pass 1 propagated unit vector q backwards, i.e. stored column q of the
                // chunk's operator O as its "row" (O[i][q] = exp(S[q]) P[q][i]), so the same sum applies with i <-> q
                double smax = -INFINITY;
#pragma unroll
                for (int i = 0; i < KP; ++i)
                    if (i < c.K && cur[i] > 0.0) smax = fmax(smax, sr[i]);
#pragma unroll
                for (int i = 0; i < KP; ++i)
                    if (i < c.K && cur[i] > 0.0 && sr[i] > -INFINITY) acc += cur[i] * exp(sr[i] - smax) * pc[i];
            }
        }
        if (j < KP) nxt[j] = acc;
        __syncthreads();
        if (!MAXPLUS) {                       // normalise (direction only)
            double m = 0.0;
            for (int i = 0; i < c.K; ++i) m = fmax(m, nxt[i]);
            __syncthreads();
            if (j < KP) cur[j] = m > 0.0 ? nxt[j] / m : nxt[j];
        } else if (j < KP) cur[j] = nxt[j];
        __syncthreads();
    }
}

// pass 3, forward: alpha-hat and the per-frame normalisers (loglik = sum_t log cnorm[t] + rowmax[t])
template <int KP>
__global__ __launch_bounds__(HMM_TPB) void hmm_forward_apply_kernel(HmmChunking c, const double* __restrict__ A, const double* __restrict__ startprob,
                                                                   const double* __restrict__ bexp, const double* __restrict__ enter,
                                                                   double* __restrict__ alpha, double* __restrict__ cnorm) {
    __shared__ double As[KP * KP];
    load_A<KP>(As, A, c.K);
    const int64_t ch = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x;
    if (ch >= c.nchunks) return;
    int64_t lo, hi;
    chunk_range(c, ch, lo, hi);
    double v[KP];
    if (ch == 0) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) { v[j] = j < c.K ? startprob[j] * bexp[j] : 0.0; sum += v[j]; }
        cnorm[0] = sum;
#pragma unroll
        for (int j = 0; j < KP; ++j) { v[j] = sum > 0.0 ? v[j] / sum : 0.0; if (j < c.K) alpha[j] = v[j]; }
    } else {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) { v[j] = j < c.K ? enter[ch * KP + j] : 0.0; sum += v[j]; }
#pragma unroll
        for (int j = 0; j < KP; ++j) v[j] = sum > 0.0 ? v[j] / sum : 0.0;
    }
    for (int64_t t = lo; t < hi; ++t) {
        hmm_step<KP, false, false>(v, As, c.K, bexp + t * c.K);
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) sum += v[j];
        cnorm[t] = sum;
        const double inv = sum > 0.0 ? 1.0 / sum : 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) { v[j] *= inv; if (j < c.K) alpha[t * c.K + j] = v[j]; }
    }
}

// pass 3, backward: gamma_t = normalise(alpha_t . u_t) and R_t[j] = b_{t+1}[j] u_{t+1}[j] / Z_t with
// Z_t = sum_j (alpha_t A)[j] b_{t+1}[j] u_{t+1}[j], so that xi_t[i][j] = alpha_t[i] A[i][j] R_t[j] (rows of R_{N-1} are zero)
template <int KP>
__global__ __launch_bounds__(HMM_TPB) void hmm_backward_apply_kernel(HmmChunking c, const double* __restrict__ A, const double* __restrict__ bexp,
                                                                    const double* __restrict__ alpha, const double* __restrict__ enter,
                                                                    double* __restrict__ gamma, double* __restrict__ R) {
    __shared__ double As[KP * KP];
    load_A<KP>(As, A, c.K);
    const int64_t ch = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x;
    if (ch >= c.nchunks) return;
    int64_t lo, hi;
    chunk_range(c, ch, lo, hi);
    const int K = c.K;
    double u[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) u[j] = j < K ? enter[ch * KP + j] : 0.0;          // u at frame hi-1
    auto emit_gamma = [&](int64_t t) {
        double g[KP], sum = 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) { g[j] = j < K ? alpha[t * K + j] * u[j] : 0.0; sum += g[j]; }
        const double inv = sum > 0.0 ? 1.0 / sum : 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j)
            if (j < K) gamma[t * K + j] = g[j] * inv;
    };
    if (hi == c.N) {
#pragma unroll
        for (int j = 0; j < KP; ++j)
            if (j < K) R[(c.N - 1) * K + j] = 0.0;
    }
    for (int64_t t = hi - 1; t >= lo; --t) {
        emit_gamma(t);
        // pair (t-1, t): w = b_t . u_t ; q = alpha_{t-1} A ; Z = q . w ; R_{t-1} = w / Z ; u_{t-1} = A w (re-normalised)
        double w[KP], q[KP], Z = 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) w[j] = j < K ? bexp[t * K + j] * u[j] : 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            double s = 0.0;
            if (j < K) {
#pragma unroll
                for (int l = 0; l < KP; ++l)
                    if (l < K) s += alpha[(t - 1) * K + l] * As[l * KP + j];
            }
            q[j] = s;
            Z += s * w[j];
        }
        const double invZ = Z > 0.0 ? 1.0 / Z : 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j)
            if (j < K) R[(t - 1) * K + j] = w[j] * invZ;
        hmm_step<KP, false, true>(u, As, K, bexp + t * K);
        renorm<KP>(u, K);
    }
    if (ch == 0) emit_gamma(0);                                                  // frame 0 belongs to no transition range
}

// pass 3 + backtracking, Viterbi.  delta is re-run inside the chunk from the entering scores and the argmax predecessors are
// kept as bytes; then every chunk follows them back from each possible end state (endmap), one workgroup chains the chunks'
// end states from the last frame backwards, and every chunk writes its part of the path.
template <int KP>
__global__ __launch_bounds__(HMM_TPB) void hmm_viterbi_apply_kernel(HmmChunking c, const double* __restrict__ logA, const double* __restrict__ logstart,
                                                                   const double* __restrict__ logB, const double* __restrict__ enter,
                                                                   unsigned char* __restrict__ psi, double* __restrict__ last) {
    __shared__ double As[KP * KP];
    for (int i = threadIdx.x; i < KP * KP; i += blockDim.x) {
        const int r = i / KP, q = i % KP;
        As[i] = (r < c.K && q < c.K) ? logA[r * c.K + q] : -INFINITY;
    }
    __syncthreads();
    const int64_t ch = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x;
    if (ch >= c.nchunks) return;
    int64_t lo, hi;
    chunk_range(c, ch, lo, hi);
    const int K = c.K;
    double v[KP];
    if (ch == 0) {
#pragma unroll
        for (int j = 0; j < KP; ++j) v[j] = j < K ? logstart[j] + logB[j] : -INFINITY;
    } else {
#pragma unroll
        for (int j = 0; j < KP; ++j) v[j] = j < K ? enter[ch * KP + j] : -INFINITY;
    }
    for (int64_t t = lo; t < hi; ++t) {
        double w[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            double best = -INFINITY;
            int arg = 0;
            if (j < K) {
#pragma unroll
                for (int l = 0; l < KP; ++l)
                    if (l < K) { const double s = v[l] + As[l * KP + j]; if (s > best) { best = s; arg = l; } }   // first maximum wins
                psi[t * K + j] = (unsigned char)arg;
                best += logB[t * K + j];
            }
            w[j] = best;
        }
#pragma unroll
        for (int j = 0; j < KP; ++j) v[j] = w[j];
    }
    if (hi == c.N) {
#pragma unroll
        for (int j = 0; j < KP; ++j)
            if (j < K) last[j] = v[j];
    }
}

__global__ __launch_bounds__(HMM_TPB) void hmm_viterbi_endmap_kernel(HmmChunking c, const unsigned char* __restrict__ psi, unsigned char* __restrict__ endmap) {
    const int64_t g = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x;
    const int64_t ch = g / c.K;
    const int j = (int)(g % c.K);
    if (ch >= c.nchunks) return;
    int64_t lo, hi;
    chunk_range(c, ch, lo, hi);
    int s = j;                                                                   // state at frame hi-1
    for (int64_t t = hi - 1; t >= lo; --t) s = psi[t * c.K + s];                 // -> state at frame lo-1
    endmap[ch * c.K + j] = (unsigned char)s;
}

__global__ __launch_bounds__(64) void hmm_viterbi_chain_kernel(HmmChunking c, const double* __restrict__ last, const unsigned char* __restrict__ endmap,
                                                               int* __restrict__ endstate, double* __restrict__ logprob) {
    if (threadIdx.x != 0) return;
    int best = 0;
    for (int j = 1; j < c.K; ++j)
        if (last[j] > last[best]) best = j;
    logprob[0] = last[best];
    int s = best;
    for (int64_t ch = c.nchunks - 1; ch >= 0; --ch) {
        endstate[ch] = s;                                                        // state at the chunk's last frame
        s = endmap[ch * c.K + s];
    }
}

__global__ __launch_bounds__(HMM_TPB) void hmm_viterbi_path_kernel(HmmChunking c, const unsigned char* __restrict__ psi, const int* __restrict__ endstate,
                                                                  int* __restrict__ path) {
    const int64_t ch = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x;
    if (ch >= c.nchunks) return;
    int64_t lo, hi;
    chunk_range(c, ch, lo, hi);
    int s = endstate[ch];
    for (int64_t t = hi - 1; t >= lo; --t) {
        path[t] = s;
        s = psi[t * c.K + s];
    }
    if (ch == 0) path[0] = s;
}

// ------------------------------------------------------------------------------------------- host side of the recursions
static int hmm_chunking(int64_t N, int K, int L, HmmChunking& c) {
    if (N < 1 || K < 1 || K > HMM_MAXK || L < 2) return 0;
    c.N = N; c.K = K; c.L = L; c.nchunks = cdiv64(N, L);
    return 1;
}
// workspace (doubles): P nchunks*KP*KP | S nchunks*KP | enter nchunks*KP
extern "C" int64_t vame_hmm_ws_doubles(int64_t N, int K, int L) {
    const int KP = K <= 16 ? 16 : 32;
    const int64_t nch = cdiv64(N, L);
    return nch * KP * KP + 2 * nch * KP + 64;
}

template <int KP>
static void hmm_forward_launch(const HmmChunking& c, const double* A, const double* start, const double* bexp, double* alpha, double* cnorm, double* ws, hipStream_t st) {
    double *P = ws, *S = P + c.nchunks * KP * KP, *enter = S + c.nchunks * KP;
    hipLaunchKernelGGL((hmm_summary_kernel<KP, false, false>), dim3((unsigned)cdiv64(c.nchunks * KP, HMM_TPB)), dim3(HMM_TPB), 0, st, c, A, bexp, P, S);
    // the scan starts from alpha-hat_0 = normalise(startprob . b_0): direction only, so the unnormalised product is enough
    hipLaunchKernelGGL((hmm_scan_kernel<KP, false, false>), dim3(1), dim3(64), 0, st, c, alpha, P, S, enter);
    hipLaunchKernelGGL((hmm_forward_apply_kernel<KP>), dim3((unsigned)cdiv64(c.nchunks, HMM_TPB)), dim3(HMM_TPB), 0, st, c, A, start, bexp, enter, alpha, cnorm);
}

__global__ void hmm_alpha0_kernel(const double* __restrict__ start, const double* __restrict__ bexp, int K, double* __restrict__ alpha) {
    const int j = threadIdx.x;
    if (j < K) alpha[j] = start[j] * bexp[j];
}
__global__ void hmm_ones_kernel(int K, double* __restrict__ v) {
    if ((int)threadIdx.x < K) v[threadIdx.x] = 1.0;
}
__global__ void hmm_vstart_kernel(const double* __restrict__ logstart, const double* __restrict__ logB, int K, double* __restrict__ v) {
    if ((int)threadIdx.x < K) v[threadIdx.x] = logstart[threadIdx.x] + logB[threadIdx.x];
}

extern "C" int vame_hmm_forward_f64(const double* bexp, int64_t N, int K, const double* startprob, const double* transmat, int L, double* alpha,
                                    double* cnorm, double* ws, void* stream) {
    VAME_CHECK_ARG(bexp && startprob && transmat && alpha && cnorm && ws, VAME_E_BADARG, "hmm_forward: null pointer");
    HmmChunking c;
    VAME_CHECK_ARG(hmm_chunking(N, K, L, c), VAME_E_SHAPE, "hmm_forward: N=%lld K=%d (<= %d) L=%d", (long long)N, K, HMM_MAXK, L);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hmm_alpha0_kernel, dim3(1), dim3(64), 0, st, startprob, bexp, K, alpha);       // scan seed (overwritten by pass 3)
    if (K <= 16) hmm_forward_launch<16>(c, transmat, startprob, bexp, alpha, cnorm, ws, st);
    else hmm_forward_launch<32>(c, transmat, startprob, bexp, alpha, cnorm, ws, st);
    VAME_LAUNCH_CHECK("hmm_forward");
    return VAME_OK;
}

template <int KP>
static void hmm_backward_launch(const HmmChunking& c, const double* A, const double* bexp, const double* alpha, double* gamma, double* R, double* ws, hipStream_t st) {
    double *P = ws, *S = P + c.nchunks * KP * KP, *enter = S + c.nchunks * KP, *ones = enter + c.nchunks * KP;
    hipLaunchKernelGGL((hmm_summary_kernel<KP, false, true>), dim3((unsigned)cdiv64(c.nchunks * KP, HMM_TPB)), dim3(HMM_TPB), 0, st, c, A, bexp, P, S);
    hipLaunchKernelGGL(hmm_ones_kernel, dim3(1), dim3(64), 0, st, c.K, ones);
    hipLaunchKernelGGL((hmm_scan_kernel<KP, false, true>), dim3(1), dim3(64), 0, st, c, ones, P, S, enter);
    hipLaunchKernelGGL((hmm_backward_apply_kernel<KP>), dim3((unsigned)cdiv64(c.nchunks, HMM_TPB)), dim3(HMM_TPB), 0, st, c, A, bexp, alpha, enter, gamma, R);
}

extern "C" int vame_hmm_backward_f64(const double* bexp, int64_t N, int K, const double* transmat, const double* alpha, int L, double* gamma, double* R,
                                     double* ws, void* stream) {
    VAME_CHECK_ARG(bexp && transmat && alpha && gamma && R && ws, VAME_E_BADARG, "hmm_backward: null pointer");
    HmmChunking c;
    VAME_CHECK_ARG(hmm_chunking(N, K, L, c), VAME_E_SHAPE, "hmm_backward: N=%lld K=%d L=%d", (long long)N, K, L);
    hipStream_t st = (hipStream_t)stream;
    if (K <= 16) hmm_backward_launch<16>(c, transmat, bexp, alpha, gamma, R, ws, st);
    else hmm_backward_launch<32>(c, transmat, bexp, alpha, gamma, R, ws, st);
    VAME_LAUNCH_CHECK("hmm_backward");
    return VAME_OK;
}

template <int KP>
static void hmm_viterbi_launch(const HmmChunking& c, const double* logA, const double* logstart, const double* logB, int* path, double* logprob, double* ws,
                               unsigned char* bws, hipStream_t st) {
    double *P = ws, *S = P + c.nchunks * KP * KP, *enter = S + c.nchunks * KP, *seed = enter + c.nchunks * KP, *last = seed + 32;
    unsigned char* psi = bws;
    unsigned char* endmap = psi + ((c.N * c.K + 15) / 16) * 16;
    int* endstate = reinterpret_cast<int*>(endmap + ((c.nchunks * c.K + 15) / 16) * 16);
    hipLaunchKernelGGL((hmm_summary_kernel<KP, true, false>), dim3((unsigned)cdiv64(c.nchunks * KP, HMM_TPB)), dim3(HMM_TPB), 0, st, c, logA, logB, P, S);
    hipLaunchKernelGGL(hmm_vstart_kernel, dim3(1), dim3(64), 0, st, logstart, logB, c.K, seed);
    hipLaunchKernelGGL((hmm_scan_kernel<KP, true, false>), dim3(1), dim3(64), 0, st, c, seed, P, S, enter);
    hipLaunchKernelGGL((hmm_viterbi_apply_kernel<KP>), dim3((unsigned)cdiv64(c.nchunks, HMM_TPB)), dim3(HMM_TPB), 0, st, c, logA, logstart, logB, enter, psi, last);
    hipLaunchKernelGGL(hmm_viterbi_endmap_kernel, dim3((unsigned)cdiv64(c.nchunks * c.K, HMM_TPB)), dim3(HMM_TPB), 0, st, c, psi, endmap);
    hipLaunchKernelGGL(hmm_viterbi_chain_kernel, dim3(1), dim3(64), 0, st, c, last, endmap, endstate, logprob);
    hipLaunchKernelGGL(hmm_viterbi_path_kernel, dim3((unsigned)cdiv64(c.nchunks, HMM_TPB)), dim3(HMM_TPB), 0, st, c, psi, endstate, path);
}

// byte workspace: psi N*K | endmap nchunks*K (padded to 16) | endstate nchunks ints
extern "C" int64_t vame_hmm_viterbi_ws_bytes(int64_t N, int K, int L) {
    const int64_t nch = cdiv64(N, L);
    return ((N * K + 15) / 16) * 16 + ((nch * K + 15) / 16) * 16 + nch * 4 + 64;
}

extern "C" int vame_hmm_viterbi_f64(const double* logB, int64_t N, int K, const double* log_startprob, const double* log_transmat, int L, int* path,
                                    double* logprob, double* ws, unsigned char* bws, void* stream) {
    VAME_CHECK_ARG(logB && log_startprob && log_transmat && path && logprob && ws && bws, VAME_E_BADARG, "hmm_viterbi: null pointer");
    VAME_CHECK_ARG((uintptr_t)bws % 16 == 0, VAME_E_SHAPE, "hmm_viterbi: byte workspace must be 16-byte aligned");
    HmmChunking c;
    VAME_CHECK_ARG(hmm_chunking(N, K, L, c), VAME_E_SHAPE, "hmm_viterbi: N=%lld K=%d L=%d", (long long)N, K, L);
    hipStream_t st = (hipStream_t)stream;
    if (K <= 16) hmm_viterbi_launch<16>(c, log_transmat, log_startprob, logB, path, logprob, ws, bws, st);
    else hmm_viterbi_launch<32>(c, log_transmat, log_startprob, logB, path, logprob, ws, bws, st);
    VAME_LAUNCH_CHECK("hmm_viterbi");
    return VAME_OK;
}

// ------------------------------------------------------------------------------------------- sufficient statistics
// Per frame-block partial sums in a fixed order (deterministic), finished by hmm_reduce_kernel:
//   small[fb][ post K | start K | trans K*K (= alpha^T R, without the factor A) | obs K*D | loglik 1 ]
//   big[fb][k][D*D] = sum_t gamma[t,k] x_t x_t^T
#define HMM_FB 256            /* frame blocks (partials per statistic) */
__global__ __launch_bounds__(HMM_TPB) void hmm_small_stats_kernel(const float* __restrict__ X, int64_t N, int D, int K, const double* __restrict__ alpha,
                                                                  const double* __restrict__ gamma, const double* __restrict__ R, const double* __restrict__ cnorm,
                                                                  const double* __restrict__ rowmax, double* __restrict__ small) {
    const int SS = 2 * K + K * K + K * D + 1;
    const int64_t per = (N + HMM_FB - 1) / HMM_FB, t0 = blockIdx.x * per, t1 = t0 + per < N ? t0 + per : N;
    double* out = small + (int64_t)blockIdx.x * SS;
    __shared__ double red[HMM_TPB];
    for (int o = threadIdx.x; o < SS - 1; o += HMM_TPB) {
        double s = 0.0;
        if (o < K) {
            for (int64_t t = t0; t < t1; ++t) s += gamma[t * K + o];
        } else if (o < 2 * K) {
            s = (blockIdx.x == 0 && N > 0) ? gamma[o - K] : 0.0;
        } else if (o < 2 * K + K * K) {
            const int i = (o - 2 * K) / K, j = (o - 2 * K) % K;
            for (int64_t t = t0; t < t1; ++t) s += alpha[t * K + i] * R[t * K + j];
        } else {
            const int k = (o - 2 * K - K * K) / D, d = (o - 2 * K - K * K) % D;
            for (int64_t t = t0; t < t1; ++t) s += gamma[t * K + k] * (double)X[t * D + d];
        }
        out[o] = s;
    }
    double ll = 0.0;
    for (int64_t t = t0 + threadIdx.x; t < t1; t += HMM_TPB) ll += log(cnorm[t]) + rowmax[t];
    red[threadIdx.x] = ll;
    __syncthreads();
    for (int s = HMM_TPB / 2; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[SS - 1] = red[0];
}

// grid (HMM_FB, K): second moments of state k over one frame block; thread (a, b) of 16 x 16 owns d1 = a + 16 i, d2 = b + 16 j
__global__ __launch_bounds__(HMM_TPB) void hmm_moment_kernel(const float* __restrict__ X, int64_t N, int D, int K, const double* __restrict__ gamma,
                                                             double* __restrict__ big) {
    constexpr int FT = 32;                                           // frames staged per tile
    __shared__ double xs[FT * 64];
    __shared__ double gs[FT];
    const int k = blockIdx.y, a = threadIdx.x >> 4, b = threadIdx.x & 15, nb = (D + 15) / 16;
    const int64_t per = (N + HMM_FB - 1) / HMM_FB, t0 = blockIdx.x * per, t1 = t0 + per < N ? t0 + per : N;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int64_t tb = t0; tb < t1; tb += FT) {
        const int nf = (int)(t1 - tb < FT ? t1 - tb : FT);
        __syncthreads();
        for (int i = threadIdx.x; i < FT * 64; i += HMM_TPB) {
            const int f = i >> 6, d = i & 63;
            xs[i] = (f < nf && d < D) ? (double)X[(tb + f) * D + d] : 0.0;
        }
        if (threadIdx.x < FT) gs[threadIdx.x] = (int)threadIdx.x < nf ? gamma[(tb + threadIdx.x) * K + k] : 0.0;
        __syncthreads();
        for (int f = 0; f < nf; ++f) {
            const double g = gs[f];
            double xa[4], xb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xa[i] = g * xs[f * 64 + a + 16 * i]; xb[i] = xs[f * 64 + b + 16 * i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nb) acc[i][j] += xa[i] * xb[j];
                }
        }
    }
    double* out = big + ((int64_t)blockIdx.x * K + k) * D * D;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d1 = a + 16 * i, d2 = b + 16 * j;
            if (d1 < D && d2 < D) out[d1 * D + d2] = acc[i][j];
        }
}

__global__ __launch_bounds__(HMM_TPB) void hmm_reduce_kernel(const double* __restrict__ parts, int nparts, int64_t n, double* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)HMM_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * HMM_TPB) {
        double s = 0.0;
        for (int p = 0; p < nparts; ++p) s += parts[(int64_t)p * n + i];
        out[i] = s;
    }
}

// stats: post K | start K | trans_raw K*K | obs K*D | loglik 1 | obsobs K*D*D ;  ws: HMM_FB * (small + K*D*D) doubles
extern "C" int64_t vame_hmm_stats_doubles(int K, int D) { return 2 * K + (int64_t)K * K + (int64_t)K * D + 1 + (int64_t)K * D * D; }
extern "C" int64_t vame_hmm_stats_ws_doubles(int K, int D) { return (int64_t)HMM_FB * vame_hmm_stats_doubles(K, D); }

extern "C" int vame_hmm_stats_f64(const float* X, int64_t N, int D, int K, const double* alpha, const double* gamma, const double* R, const double* cnorm,
                                  const double* rowmax, double* stats, double* ws, void* stream) {
    VAME_CHECK_ARG(X && alpha && gamma && R && cnorm && rowmax && stats && ws, VAME_E_BADARG, "hmm_stats: null pointer");
    VAME_CHECK_ARG(N >= 1 && D >= 1 && D <= 64 && K >= 1 && K <= HMM_MAXK, VAME_E_SHAPE, "hmm_stats: N=%lld D=%d K=%d", (long long)N, D, K);
    hipStream_t st = (hipStream_t)stream;
    const int64_t SS = 2 * K + (int64_t)K * K + (int64_t)K * D + 1, BS = (int64_t)K * D * D;
    double* small = ws;
    double* big = ws + HMM_FB * SS;
    hipLaunchKernelGGL(hmm_small_stats_kernel, dim3(HMM_FB), dim3(HMM_TPB), 0, st, X, N, D, K, alpha, gamma, R, cnorm, rowmax, small);
    hipLaunchKernelGGL(hmm_moment_kernel, dim3(HMM_FB, K), dim3(HMM_TPB), 0, st, X, N, D, K, gamma, big);
    hipLaunchKernelGGL(hmm_reduce_kernel, dim3((unsigned)cdiv64(SS, HMM_TPB)), dim3(HMM_TPB), 0, st, (const double*)small, HMM_FB, SS, stats);
    hipLaunchKernelGGL(hmm_reduce_kernel, dim3((unsigned)(cdiv64(BS, HMM_TPB) < 1024 ? cdiv64(BS, HMM_TPB) : 1024)), dim3(HMM_TPB), 0, st, (const double*)big,
                       HMM_FB, BS, stats + SS);
    VAME_LAUNCH_CHECK("hmm_stats");
    return VAME_OK;
}
