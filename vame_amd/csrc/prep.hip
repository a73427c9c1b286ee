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
// a NaN of feature f becomes the LAST valid sample (in time) of feature f.  One block per feature: find the first and
// last valid frame, report their values (first_last[f] = {first, last}, NaN when the feature has no valid sample -- the
// host resolves that rare case exactly as np.interp does), then overwrite the NaNs.
__global__ __launch_bounds__(256) void prep_fill_last_kernel(double* __restrict__ z, int64_t N, int64_t ld,
                                                             double* __restrict__ first_last) {
    __shared__ long long s_first[256], s_last[256];
    double* row = z + (int64_t)blockIdx.x * ld;
    long long first = N, last = -1;
    bool any_nan = false;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {
        const bool ok = !isnan(row[n]);
        if (ok) { if (n < first) first = n; if (n > last) last = n; }
        else any_nan = true;
    }
    s_first[threadIdx.x] = first; s_last[threadIdx.x] = last;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            if (s_first[threadIdx.x + s] < s_first[threadIdx.x]) s_first[threadIdx.x] = s_first[threadIdx.x + s];
            if (s_last[threadIdx.x + s] > s_last[threadIdx.x]) s_last[threadIdx.x] = s_last[threadIdx.x + s];
        }
        __syncthreads();
    }
    first = s_first[0]; last = s_last[0];
    const double nanv = __builtin_nan("");
    const double fill = last >= 0 ? row[last] : nanv;
    if (threadIdx.x == 0) {
        first_last[2 * blockIdx.x] = last >= 0 ? row[first] : nanv;
        first_last[2 * blockIdx.x + 1] = fill;
    }
    __syncthreads();                        // row[last] / row[first] are never NaN, so nobody rewrites them below
    if (last >= 0 && any_nan)
        for (int64_t n = threadIdx.x; n < N; n += blockDim.x)
            if (isnan(row[n])) row[n] = fill;
}

extern "C" int vame_prep_fill_last_valid_f64(double* z, int F, int64_t N, int64_t ld, double* first_last, void* stream) {
    VAME_CHECK_ARG(z && first_last, VAME_E_BADARG, "prep_fill_last_valid: null pointer");
    VAME_CHECK_ARG(F >= 1 && N >= 1 && ld >= N, VAME_E_SHAPE, "prep_fill_last_valid: bad shape F=%d N=%lld", F, (long long)N);
    hipLaunchKernelGGL(prep_fill_last_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, z, N, ld, first_last);
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
// Population std over time per feature (np.std(X.T, axis=1), create_training.py:153), two passes, one block per feature,
// fixed summation tree: rows with identical contents give identical results (the anchor search relies on exact ties).
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

__global__ __launch_bounds__(256) void prep_rowstats_kernel(const double* __restrict__ x, int64_t N, int64_t ld,
                                                            double* __restrict__ mean_out, double* __restrict__ std_out) {
    __shared__ double sh[256];
    const double* row = x + (int64_t)blockIdx.x * ld;
    double s = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) s += row[n];
    const double mean = block_sum_f64(s, sh) / (double)N;
    double q = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) { const double d = row[n] - mean; q += d * d; }
    const double var = block_sum_f64(q, sh) / (double)N;
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = mean; std_out[blockIdx.x] = sqrt(var); }
}

extern "C" int vame_prep_rowstats_f64(const double* x, int F, int64_t N, int64_t ld, double* mean_out, double* std_out,
                                      void* stream) {
    VAME_CHECK_ARG(x && mean_out && std_out, VAME_E_BADARG, "prep_rowstats: null pointer");
    VAME_CHECK_ARG(F >= 1 && N >= 1 && ld >= N, VAME_E_SHAPE, "prep_rowstats: bad shape F=%d N=%lld", F, (long long)N);
    hipLaunchKernelGGL(prep_rowstats_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, x, N, ld, mean_out, std_out);
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
