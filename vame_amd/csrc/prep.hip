// Training-set preparation kernels (float64, HBM-bound): the O(N*F) passes of vame/model/create_training.py
// (z-score + IQR outlier masking, the reference's two NaN-fill rules, per-feature std for the anchor search and the
// Savitzky-Golay smoothing along time).  Data layout everywhere: (F, N) feature-major with a leading dimension `ld`,
// i.e. the on-disk layout of `<file>-PE-seq.npy` / `train_seq.npy`; a file occupies a column range of the concatenated
// buffer.  Results are bit-identical to numpy / scipy: IEEE division, no fused multiply-add where the CPU code has a
// separate multiply and add (`#pragma clang fp contract(off)`), scipy's summation order in the filter.
#include "vame_common.h"
#include <math.h>

static inline int prep_blocks(int64_t n) {
    int64_t b = cdiv64(n, 256);
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---------------------------------------------------------------------------------------------- z-score + outlier mask
// z[f, off + n] = (x[f, n] - mean) / std; robust: z > cutoff or z < -cutoff -> NaN   (create_training.py:112-143, :217-234)
__global__ __launch_bounds__(256) void prep_zscore_kernel(const double* __restrict__ x, int F, int64_t N, int64_t ldx, double mean,
                                                          double sd, double cutoff, int robust, double* __restrict__ z,
                                                          int64_t ldz) {
    const int64_t total = (int64_t)F * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / N, n = i - f * N;
        double v = (x[f * ldx + n] - mean) / sd;
        if (robust && (v > cutoff || v < -cutoff)) v = __builtin_nan("");
        z[f * ldz + n] = v;
    }
}

extern "C" int vame_prep_zscore_mask_f64(const double* x, int F, int64_t N, int64_t ldx, double mean, double sd, double cutoff,
                                         int robust, double* z, int64_t ldz, void* stream) {
    VAME_CHECK_ARG(x && z, VAME_E_BADARG, "prep_zscore: null pointer");
    VAME_CHECK_ARG(F >= 1 && N >= 1 && ldx >= N && ldz >= N, VAME_E_SHAPE, "prep_zscore: bad shape F=%d N=%lld", F, (long long)N);
    hipLaunchKernelGGL(prep_zscore_kernel, dim3(prep_blocks((int64_t)F * N)), dim3(256), 0, (hipStream_t)stream, x, F, N, ldx, mean,
                       sd, cutoff, robust, z, ldz);
    VAME_LAUNCH_CHECK("prep_zscore");
    return VAME_OK;
}

// ---------------------------------------------------------------------------------------------- NaN fill, "aligned" rule
// The reference's interpol() on a whole (N, F) array interpolates over the FEATURE index (create_training.py:27-32,145):
// a NaN of feature f becomes the LAST valid sample (in time) of feature f.  Three launches over a (chunks, F) grid:
// reset the per-feature {first, last} valid frame indices, reduce them with 64-bit atomic min / max (order-independent),
// then overwrite the NaNs of every chunk with row[last].  first_last[f] = {row[first], row[last]} (NaN when the feature
// has no valid sample -- the host resolves that rare case exactly as np.interp does).
constexpr int PREP_CHUNK = 8192;           // frames per block

__global__ void prep_fill_init_kernel(long long* __restrict__ idx, int F, int64_t N) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) { idx[2 * f] = (long long)N; idx[2 * f + 1] = -1; }
}

__global__ __launch_bounds__(256) void prep_fill_scan_kernel(const double* __restrict__ z, int64_t N, int64_t ld,
                                                             long long* __restrict__ idx) {
    __shared__ long long s_first[256], s_last[256];
    const double* row = z + (int64_t)blockIdx.y * ld;
    const int64_t n0 = (int64_t)blockIdx.x * PREP_CHUNK, n1 = n0 + PREP_CHUNK < N ? n0 + PREP_CHUNK : N;
    long long first = N, last = -1;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += blockDim.x)
        if (!isnan(row[n])) { if (n < first) first = n; if (n > last) last = n; }
    s_first[threadIdx.x] = first; s_last[threadIdx.x] = last;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            if (s_first[threadIdx.x + s] < s_first[threadIdx.x]) s_first[threadIdx.x] = s_first[threadIdx.x + s];
            if (s_last[threadIdx.x + s] > s_last[threadIdx.x]) s_last[threadIdx.x] = s_last[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_last[0] >= 0) {
        atomicMin(&idx[2 * blockIdx.y], s_first[0]);
        atomicMax(&idx[2 * blockIdx.y + 1], s_last[0]);
    }
}

__global__ __launch_bounds__(256) void prep_fill_apply_kernel(double* __restrict__ z, int64_t N, int64_t ld,
                                                              const long long* __restrict__ idx, double* __restrict__ first_last) {
    double* row = z + (int64_t)blockIdx.y * ld;
    const long long first = idx[2 * blockIdx.y], last = idx[2 * blockIdx.y + 1];
    const double nanv = __builtin_nan("");
    const double fill = last >= 0 ? row[last] : nanv;       // row[last] / row[first] are valid: no block rewrites them
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        first_last[2 * blockIdx.y] = last >= 0 ? row[first] : nanv;
        first_last[2 * blockIdx.y + 1] = fill;
    }
    if (last < 0) return;
    const int64_t n0 = (int64_t)blockIdx.x * PREP_CHUNK, n1 = n0 + PREP_CHUNK < N ? n0 + PREP_CHUNK : N;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += blockDim.x)
        if (isnan(row[n])) row[n] = fill;
}

extern "C" int64_t vame_prep_ws_bytes(int F) { return (int64_t)F * 2 * 256 * 8; }

extern "C" int vame_prep_fill_last_valid_f64(double* z, int F, int64_t N, int64_t ld, double* first_last, void* ws,
                                             void* stream) {
    VAME_CHECK_ARG(z && first_last && ws, VAME_E_BADARG, "prep_fill_last_valid: null pointer");
    VAME_CHECK_ARG(F >= 1 && N >= 1 && ld >= N, VAME_E_SHAPE, "prep_fill_last_valid: bad shape F=%d N=%lld", F, (long long)N);
    hipStream_t st = (hipStream_t)stream;
    long long* idx = reinterpret_cast<long long*>(ws);
    const dim3 grid((unsigned)cdiv64(N, PREP_CHUNK), (unsigned)F);
    hipLaunchKernelGGL(prep_fill_init_kernel, dim3((F + 63) / 64), dim3(64), 0, st, idx, F, N);
    hipLaunchKernelGGL(prep_fill_scan_kernel, grid, dim3(256), 0, st, (const double*)z, N, ld, idx);
    hipLaunchKernelGGL(prep_fill_apply_kernel, grid, dim3(256), 0, st, z, N, ld, (const long long*)idx, first_last);
    VAME_LAUNCH_CHECK("prep_fill_last_valid");
    return VAME_OK;
}

// ---------------------------------------------------------------------------------------------- NaN fill, "fixed" rule
// interpol() on one frame's feature vector (create_training.py:236): np.interp across the feature index, clamped to the
// first / last valid feature.  One thread per frame; np.interp's arithmetic `slope * (x - xp[j]) + fp[j]` with
// slope = (fp[j+1] - fp[j]) / (xp[j+1] - xp[j]), unfused.  Frames without any valid feature stay NaN and are counted
// (np.interp raises on them in the reference).
__global__ __launch_bounds__(256) void prep_fill_features_kernel(double* __restrict__ z, int F, int64_t N, int64_t ld,
                                                                 int* __restrict__ n_empty) {
#pragma clang fp contract(off)
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        int prev = -1;                      // last valid feature index seen so far
        bool any_nan = false, any_ok = false;
        for (int f = 0; f < F; ++f) {
            const bool ok = !isnan(z[(int64_t)f * ld + n]);
            any_nan |= !ok; any_ok |= ok;
        }
        if (!any_nan) continue;
        if (!any_ok) { atomicAdd(n_empty, 1); continue; }
        for (int f = 0; f < F; ++f) {
            const double v = z[(int64_t)f * ld + n];
            if (!isnan(v)) { prev = f; continue; }
            int next = f + 1;
            while (next < F && isnan(z[(int64_t)next * ld + n])) ++next;
            double out;
            if (prev < 0) out = z[(int64_t)next * ld + n];                    // left of the first valid: fp[0]
            else if (next >= F) out = z[(int64_t)prev * ld + n];              // right of the last valid: fp[-1]
            else {
                const double y0 = z[(int64_t)prev * ld + n], y1 = z[(int64_t)next * ld + n];
                const double slope = (y1 - y0) / ((double)next - (double)prev);
                out = slope * ((double)f - (double)prev) + y0;
            }
            z[(int64_t)f * ld + n] = out;   // safe: later NaNs of this frame only look at `prev` (valid) and forward
        }
    }
}

extern "C" int vame_prep_fill_across_features_f64(double* z, int F, int64_t N, int64_t ld, int* n_empty, void* stream) {
    VAME_CHECK_ARG(z && n_empty, VAME_E_BADARG, "prep_fill_across_features: null pointer");
    VAME_CHECK_ARG(F >= 1 && N >= 1 && ld >= N, VAME_E_SHAPE, "prep_fill_across_features: bad shape F=%d N=%lld", F, (long long)N);
    hipLaunchKernelGGL(prep_fill_features_kernel, dim3(prep_blocks(N)), dim3(256), 0, (hipStream_t)stream, z, F, N, ld, n_empty);
    VAME_LAUNCH_CHECK("prep_fill_across_features");
    return VAME_OK;
}

// ---------------------------------------------------------------------------------------------- per-feature mean / std
// Population std over time per feature (np.std(X.T, axis=1), create_training.py:153).  Two passes (mean, then squared
// deviations), each a (<= 256 chunks, F) grid of fixed-tree block sums into a partial table plus one block per feature that adds
// the partials in index order: the summation tree depends only on N, so rows with identical contents give identical results
// (the anchor search relies on exact ties).
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void prep_rowpartial_kernel(const double* __restrict__ x, int64_t N, int64_t ld, int64_t per,
                                                              const double* __restrict__ centre, double* __restrict__ partial) {
    __shared__ double sh[256];
    const double* row = x + (int64_t)blockIdx.y * ld;
    const double c = centre ? centre[blockIdx.y] : 0.0;
    const int64_t n0 = (int64_t)blockIdx.x * per, n1 = n0 + per < N ? n0 + per : N;
    double s = 0.0;
    if (centre) for (int64_t n = n0 + threadIdx.x; n < n1; n += blockDim.x) { const double d = row[n] - c; s += d * d; }
    else for (int64_t n = n0 + threadIdx.x; n < n1; n += blockDim.x) s += row[n];
    s = block_sum_f64(s, sh);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ void prep_rowfinal_kernel(const double* __restrict__ partial, int nblk, int F, int64_t N, int take_sqrt,
                                     double* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)f * nblk + b];
    s /= (double)N;
    out[f] = take_sqrt ? sqrt(s) : s;
}

extern "C" int vame_prep_rowstats_f64(const double* x, int F, int64_t N, int64_t ld, double* mean_out, double* std_out, void* ws,
                                      void* stream) {
    VAME_CHECK_ARG(x && mean_out && std_out && ws, VAME_E_BADARG, "prep_rowstats: null pointer");
    VAME_CHECK_ARG(F >= 1 && N >= 1 && ld >= N, VAME_E_SHAPE, "prep_rowstats: bad shape F=%d N=%lld", F, (long long)N);
    hipStream_t st = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(ws);
    int64_t per = cdiv64(N, 256);
    per = cdiv64(per < 2048 ? 2048 : per, 256) * 256;
    const int nblk = (int)cdiv64(N, per);                     // <= 256
    const dim3 grid((unsigned)nblk, (unsigned)F);
    hipLaunchKernelGGL(prep_rowpartial_kernel, grid, dim3(256), 0, st, x, N, ld, per, (const double*)nullptr, partial);
    hipLaunchKernelGGL(prep_rowfinal_kernel, dim3((F + 63) / 64), dim3(64), 0, st, (const double*)partial, nblk, F, N, 0, mean_out);
    hipLaunchKernelGGL(prep_rowpartial_kernel, grid, dim3(256), 0, st, x, N, ld, per, (const double*)mean_out, partial);
    hipLaunchKernelGGL(prep_rowfinal_kernel, dim3((F + 63) / 64), dim3(64), 0, st, (const double*)partial, nblk, F, N, 1, std_out);
    VAME_LAUNCH_CHECK("prep_rowstats");
    return VAME_OK;
}

// ---------------------------------------------------------------------------------------------- Savitzky-Golay, interior
// y[f, n] for half <= n < N - half, half = L / 2, with scipy's symmetric correlate1d order (ni_filters.c):
//   tmp = x[n] * w[half];  for d = half .. 1:  tmp += (x[n - d] + x[n + d]) * w[half - d]
// where w is the REVERSED coefficient vector scipy.signal.savgol_filter hands to convolve1d (the caller passes it).  The
// first / last `half` columns are copied through; mode='interp' refits them from the edge windows on the host.
__global__ __launch_bounds__(256) void prep_savgol_kernel(const double* __restrict__ x, int F, int64_t N, int64_t ldx,
                                                          const double* __restrict__ w, int L, double* __restrict__ y,
                                                          int64_t ldy) {
#pragma clang fp contract(off)
    const int half = L / 2;
    const int64_t total = (int64_t)F * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / N, n = i - f * N;
        const double* row = x + f * ldx;
        double tmp;
        if (n < half || n >= N - half) tmp = row[n];
        else {
            tmp = row[n] * w[half];
            for (int d = half; d >= 1; --d) tmp += (row[n - d] + row[n + d]) * w[half - d];
        }
        y[f * ldy + n] = tmp;
    }
}

extern "C" int vame_prep_savgol_f64(const double* x, int F, int64_t N, int64_t ldx, const double* w, int L, double* y,
                                    int64_t ldy, void* stream) {
    VAME_CHECK_ARG(x && w && y && x != y, VAME_E_BADARG, "prep_savgol: null or aliased pointer");
    VAME_CHECK_ARG(F >= 1 && N >= L && L >= 1 && (L & 1) && ldx >= N && ldy >= N, VAME_E_SHAPE,
                   "prep_savgol: bad shape F=%d N=%lld L=%d (odd window <= N)", F, (long long)N, L);
    hipLaunchKernelGGL(prep_savgol_kernel, dim3(prep_blocks((int64_t)F * N)), dim3(256), 0, (hipStream_t)stream, x, F, N, ldx, w, L, y,
                       ldy);
    VAME_LAUNCH_CHECK("prep_savgol");
    return VAME_OK;
}
