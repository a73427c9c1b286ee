// Bandwidth-bound kernels of the path: sliding-window gather, reparameterisation + KL, MSE
// fwd/bwd, column sums, fused Adam-AMSGrad, and the (Z,Z) nuclear-norm ("kmeans") loss.
#include "vame_common.h"
#include <math.h>

static thread_local char g_err[512] = "";
void vame_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char* vame_last_error(void) { return g_err; }
extern "C" int vame_version(void) { return 100; }
#ifndef VAME_SRC_ID
#define VAME_SRC_ID "unidentified"          /* the emulator build, or a compile outside the Makefile */
#endif
extern "C" const char* vame_source_id(void) { return VAME_SRC_ID; }

// ---- clock stamps (measurement only; bench.py): the shader clock moves with the power budget (2.0-2.3 GHz under the MFMA kernels of
// this path, 2.4 GHz nominal), so a fraction of the nominal peak mixes kernel quality with the box.  One launch of this kernel before
// and one after a region give (s_memtime = shader-clock counter, s_memrealtime = constant 100 MHz counter) pairs per XCD; the ratio of
// the two differences x 100 MHz is the average shader clock over the region.
__global__ __launch_bounds__(64) void clock_stamp_kernel(long long* __restrict__ out) {
#ifdef VAME_EMU
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = 0; out[blockIdx.x * 4 + 1] = 0; out[blockIdx.x * 4 + 2] = blockIdx.x & 7; out[blockIdx.x * 4 + 3] = 0; }
#else
    const long long t = (long long)__builtin_amdgcn_s_memtime(), r = (long long)__builtin_amdgcn_s_memrealtime();
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) {
        long long* o = out + (long long)blockIdx.x * 4;
        o[0] = t; o[1] = r; o[2] = (long long)(xcc & 15u); o[3] = (long long)hwid;
    }
#endif
}
extern "C" int vame_clock_stamp(int64_t* out, int nblocks, void* stream) {
    VAME_CHECK_ARG(out && nblocks >= 1 && nblocks <= 1024, VAME_E_BADARG, "clock_stamp: out null or nblocks=%d not in 1..1024", nblocks);
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<long long*>(out));
    VAME_LAUNCH_CHECK("clock_stamp");
    return VAME_OK;
}

static inline int ew_blocks(int64_t n, int per_block = 256) {
    int64_t b = cdiv64(n, per_block);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));   // <= 256 CUs x 8 blocks, grid-stride the rest
}

__device__ __forceinline__ float block_sum_256(float v) {
    __shared__ float part[4];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) part[w] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

// --------------------------------------------------------------------------------- window gather
// out[b,l,f] = X[f*N + start_b + l].  One block per window: reads are contiguous along l for each
// feature row (coalesced 4*L bytes), transposed through LDS, written as one contiguous L*F run.
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ X, int64_t N, int F,
                                                            const int64_t* __restrict__ starts, int64_t start0, int B,
                                                            int L, float* __restrict__ out) {
    VAME_DYN_SMEM(smem_raw);
    float* tile = reinterpret_cast<float*>(smem_raw);   // [F][L+1]
    const int LP = L + 1;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int64_t s = starts ? starts[b] : start0 + b;
        for (int i = threadIdx.x; i < F * L; i += blockDim.x) {
            const int f = i / L, l = i % L;
            tile[f * LP + l] = X[(int64_t)f * N + s + l];
        }
        __syncthreads();
        float* o = out + (int64_t)b * L * F;
        for (int i = threadIdx.x; i < F * L; i += blockDim.x) {
            const int l = i / F, f = i % F;
            o[i] = tile[f * LP + l];
        }
        __syncthreads();
    }
}

extern "C" int vame_window_gather_f32(const float* X, int64_t N, int F, const int64_t* starts, int64_t start0, int B,
                                      int L, float* out, void* stream) {
    VAME_CHECK_ARG(X && out, VAME_E_BADARG, "window_gather: null pointer");
    VAME_CHECK_ARG(B >= 0 && F >= 1 && L >= 1 && N >= L, VAME_E_SHAPE, "window_gather: bad shape B=%d F=%d L=%d N=%lld", B, F,
                   L, (long long)N);
    if (B == 0) return VAME_OK;
    const size_t sh = (size_t)F * (L + 1) * sizeof(float);
    VAME_CHECK_ARG(sh <= 64 * 1024, VAME_E_SHAPE, "window_gather: F*L too large for the LDS transpose");
    hipLaunchKernelGGL(window_gather_kernel, dim3(B < 4096 ? B : 4096), dim3(256), sh, (hipStream_t)stream, X, N, F, starts,
                       start0, B, L, out);
    VAME_LAUNCH_CHECK("window_gather");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- latent fwd/bwd
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// Counter-based N(0,1) draw for the reparameterisation (rnn_model.py:71-74: `epsilon = torch.randn_like(...)`): Philox4x32-10 keyed by the
// caller's seed, counter = (element index, step), Box-Muller on two of its four words.  A draw depends on (seed, step, element) only --
// reproducible, independent of the launch geometry, nothing to carry between launches but the step number.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float philox_normal(uint64_t idx, uint64_t seed, uint64_t step) {
    uint32_t r[4];
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1): 24 bits, never 0 or 1
    const float u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// rng (device, 4 x uint64, or null): {seed, step, ticket, -}.  Non-null in training mode: eps is an OUTPUT -- the draw of this launch -- and the
// last workgroup to finish advances `step` (every workgroup has read it by then), so consecutive launches -- incl. replays of a captured
// graph, whose arguments are frozen -- draw fresh values with no host involvement.
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv_raw,
                                                         float* eps, int64_t n, int softplus,
                                                         int training, float* __restrict__ logvar, float* __restrict__ z,
                                                         float* __restrict__ kl_out, unsigned long long* rng) {
    float part = 0.f;
    const bool gen = rng != nullptr && training;
    const uint64_t seed = gen ? rng[0] : 0, step = gen ? rng[1] : 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mu[i];
        const float lv = softplus ? softplus_f(lv_raw[i]) : lv_raw[i];
        const float ev = expf(lv);
        logvar[i] = lv;
        float e = 0.f;
        if (gen) { e = philox_normal((uint64_t)i, seed, step); eps[i] = e; }
        else if (training) e = eps[i];
        z[i] = training ? e * expf(0.5f * lv) + m : m;
        part += 1.0f + lv - m * m - ev;
    }
    if (kl_out) {
        part = block_sum_256(part);
        if (threadIdx.x == 0) atomicAdd(kl_out, part);
    }
    if (gen) {
        __syncthreads();                                   // every thread of this workgroup has read `step`
        if (threadIdx.x == 0) {
            int* ticket = reinterpret_cast<int*>(rng + 2);
            if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) { *ticket = 0; rng[1] = step + 1; }
        }
    }
}

extern "C" int vame_latent_fwd_f32(const float* mu, const float* lv_raw, float* eps, int B, int Z, int softplus,
                                   int training, float* logvar, float* z, float* kl_out, uint64_t* rng, void* stream) {
    VAME_CHECK_ARG(mu && lv_raw && logvar && z, VAME_E_BADARG, "latent_fwd: null pointer");
    VAME_CHECK_ARG(!training || eps, VAME_E_BADARG, "latent_fwd: training mode needs eps (input, or output of the draw when rng is given)");
    VAME_CHECK_ARG(B >= 1 && Z >= 1, VAME_E_SHAPE, "latent_fwd: bad shape");
    const int64_t n = (int64_t)B * Z;
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, mu, lv_raw, eps, n,
                       softplus, training, logvar, z, kl_out, reinterpret_cast<unsigned long long*>(rng));
    VAME_LAUNCH_CHECK("latent_fwd");
    return VAME_OK;
}

// The step's loss bookkeeping (rnn_vae.py:129-150: loss = rec + fut + BETA kl_weight kl + kl_weight kmeans; train_loss += loss.item() ...)
// in one single-thread launch: raw = the four sums the loss kernels left ([rec, fut, sum(1 + logvar - mu^2 - exp logvar), kmeans]) -> the
// terms in the reference's units (x scale), the weighted total, the epoch accumulators (float64: total, rec, fut, kl, kmeans, total of
// the LAST step) -- and raw is zeroed for the next step's atomics.
__global__ void loss_finish_kernel(float* raw, float s0, float s1, float s2, float s3, float w0, float w1, float w2, float w3, int with_fut,
                                   float* out, double* acc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float t0 = raw[0] * s0, t1 = with_fut ? raw[1] * s1 : 0.f, t2 = raw[2] * s2, t3 = raw[3] * s3;
    const float total = ((w0 * t0 + w1 * t1) + w2 * t2) + w3 * t3;
    out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3; out[4] = total;
    if (acc) { acc[0] += total; acc[1] += t0; acc[2] += t1; acc[3] += t2; acc[4] += t3; acc[5] = total; }
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = 0.f;
}

extern "C" int vame_loss_finish_f32(float* raw, const float* scale, const float* weights, int with_fut, float* out, double* acc, void* stream) {
    VAME_CHECK_ARG(raw && scale && weights && out, VAME_E_BADARG, "loss_finish: null pointer");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, raw, scale[0], scale[1], scale[2], scale[3], weights[0], weights[1],
                       weights[2], weights[3], with_fut, out, acc);
    VAME_LAUNCH_CHECK("loss_finish");
    return VAME_OK;
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ mu,
                                                         const float* __restrict__ logvar, const float* __restrict__ lv_raw,
                                                         const float* __restrict__ eps, int64_t n, int softplus, float ckl,
                                                         float* __restrict__ dmu, float* __restrict__ dlv) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = dz[i], lv = logvar[i];
        dmu[i] = d + ckl * mu[i];
        float g = d * eps[i] * 0.5f * expf(0.5f * lv) + 0.5f * ckl * (expf(lv) - 1.0f);
        if (softplus) g *= 1.0f / (1.0f + expf(-lv_raw[i]));
        dlv[i] = g;
    }
}

extern "C" int vame_latent_bwd_f32(const float* dz, const float* mu, const float* logvar, const float* lv_raw,
                                   const float* eps, int B, int Z, int softplus, float ckl, float* dmu, float* dlv,
                                   void* stream) {
    VAME_CHECK_ARG(dz && mu && logvar && lv_raw && eps && dmu && dlv, VAME_E_BADARG, "latent_bwd: null pointer");
    const int64_t n = (int64_t)B * Z;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dz, mu, logvar, lv_raw, eps,
                       n, softplus, ckl, dmu, dlv);
    VAME_LAUNCH_CHECK("latent_bwd");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- MSE fwd + bwd
// A workgroup walks whole rows (pred is contiguous; the target row is a window inside a wider buffer): no per-element division,
// 16-byte accesses when every row start is 16-byte aligned (VEC).
template <bool VEC>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                  int64_t tgt_row, int B, int TF, float gscale, float* __restrict__ dpred,
                                                  float* __restrict__ loss_out) {
    float part = 0.f;
    if (VEC) {
        // four rows per pass: their eight loads are in flight together (one float atomic per workgroup ends the kernel, and atomics
        // on one address serialise at ~15 ns each -- hence few workgroups with deep passes rather than one row per workgroup)
        const int n4 = TF / 4;
        for (int b0 = blockIdx.x * 4; b0 < B; b0 += gridDim.x * 4) {
            for (int j = threadIdx.x; j < n4; j += blockDim.x) {
                float4 pv[4], tv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = b0 + r < B ? b0 + r : B - 1;
                    pv[r] = reinterpret_cast<const float4*>(pred + (int64_t)b * TF)[j];
                    tv[r] = reinterpret_cast<const float4*>(target + (int64_t)b * tgt_row)[j];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (b0 + r >= B) break;
                    const float4 e = make_float4(pv[r].x - tv[r].x, pv[r].y - tv[r].y, pv[r].z - tv[r].z, pv[r].w - tv[r].w);
                    part += (e.x * e.x + e.y * e.y) + (e.z * e.z + e.w * e.w);
                    if (dpred) reinterpret_cast<float4*>(dpred + (int64_t)(b0 + r) * TF)[j] = make_float4(gscale * e.x, gscale * e.y, gscale * e.z, gscale * e.w);
                }
            }
        }
    } else {
        for (int b = blockIdx.x; b < B; b += gridDim.x) {
            const float* p = pred + (int64_t)b * TF;
            const float* t = target + (int64_t)b * tgt_row;
            float* d = dpred ? dpred + (int64_t)b * TF : nullptr;
            for (int j = threadIdx.x; j < TF; j += blockDim.x) {
                const float e = p[j] - t[j];
                part += e * e;
                if (d) d[j] = gscale * e;
            }
        }
    }
    part = block_sum_256(part);
    if (threadIdx.x == 0 && loss_out) atomicAdd(loss_out, part);
}

extern "C" int vame_mse_fwd_bwd_f32(const float* pred, const float* target, int64_t tgt_row, int B, int TF, float gscale,
                                    float* dpred, float* loss_out, void* stream) {
    VAME_CHECK_ARG(pred && target, VAME_E_BADARG, "mse: null pointer");
    VAME_CHECK_ARG(B >= 1 && TF >= 1 && tgt_row >= TF, VAME_E_SHAPE, "mse: bad shape");
    const bool vec = TF % 4 == 0 && tgt_row % 4 == 0 && (uintptr_t)pred % 16 == 0 && (uintptr_t)target % 16 == 0 && (uintptr_t)dpred % 16 == 0;
    const dim3 grid(B < 2048 ? B : 2048), grid4((B + 3) / 4 < 512 ? (B + 3) / 4 : 512);
    if (vec) hipLaunchKernelGGL(mse_kernel<true>, grid4, dim3(256), 0, (hipStream_t)stream, pred, target, tgt_row, B, TF, gscale, dpred, loss_out);
    else hipLaunchKernelGGL(mse_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, pred, target, tgt_row, B, TF, gscale, dpred, loss_out);
    VAME_LAUNCH_CHECK("mse");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- column sums
// out[c] (+)= sum_r in[r*ld + c].  Row slabs are reduced by a 64-column x 4-row-lane workgroup each
// (coalesced 256 B row segments), slab partials land in `ws` and a second tiny pass adds them in a fixed
// order (deterministic, no float atomics).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ in, int64_t R, int C, int64_t ld,
                                                             int64_t rows_per_slab, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int64_t lo = (int64_t)blockIdx.y * rows_per_slab;
    const int64_t hi = lo + rows_per_slab < R ? lo + rows_per_slab : R;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int64_t r = lo + ty;
        for (; r + 4 < hi; r += 8) { s0 += in[r * ld + c]; s1 += in[(r + 4) * ld + c]; }
        if (r < hi) s0 += in[r * ld + c];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}
// The same for wide, 16-byte aligned matrices (C % 4 == 0, ld % 4 == 0): a thread owns four columns and the rows lo + ty, lo + ty + 4, ... of its slab,
// ALL of whose loads (<= 16 x 16 bytes) are issued before the first add -- the scalar form above walks its rows two loads at a time, one HBM round trip
// each (20 us for a 4096 x 512 matrix that is 8 MB).  Four interleaved partial sums per column, fixed order.
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ in, int64_t R, int C, int64_t ld,
                                                              int64_t rows_per_slab, float* __restrict__ part) {
    __shared__ float4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + tx) * 4;
    const int64_t lo = (int64_t)blockIdx.y * rows_per_slab;
    const int64_t hi = lo + rows_per_slab < R ? lo + rows_per_slab : R;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        for (int64_t r0 = lo + ty; r0 < hi; r0 += 64) {               // (slabs are <= 64 rows for R <= 8192: one trip)
            float4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t r = r0 + 4 * i;
                v[i] = r < hi ? *reinterpret_cast<const float4*>(in + r * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float4 t;
        t.x = (red[0][tx].x + red[1][tx].x) + (red[2][tx].x + red[3][tx].x); t.y = (red[0][tx].y + red[1][tx].y) + (red[2][tx].y + red[3][tx].y);
        t.z = (red[0][tx].z + red[1][tx].z) + (red[2][tx].z + red[3][tx].z); t.w = (red[0][tx].w + red[1][tx].w) + (red[2][tx].w + red[3][tx].w);
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * C + c) = t;
    }
}
// 16 row lanes per column: at most 8 dependent loads per thread over the <= 128 slab partials (with 4 lanes the single workgroup
// of a narrow sum spent 8 us on 32 dependent loads); fixed summation order.
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int nslabs, int C,
                                                            float* __restrict__ out, int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < C)
        for (int i = ty; i < nslabs; i += 16) s += part[(int64_t)i * C + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tx];
        out[c] = accumulate ? out[c] + t : t;
    }
}

extern "C" int64_t vame_colsum_ws_floats(int64_t R, int C) {
    const int64_t nslabs = R <= 64 ? 1 : (cdiv64(R, 64) < 128 ? cdiv64(R, 64) : 128);
    return nslabs * C;
}

// Narrow contiguous matrices (ld == C, C % 4 == 0, C <= 64: the (B*T, 24) dpred): the kernel above would use 24 of 64 lanes on 96-byte
// rows.  Here the matrix is a flat float4 stream; a thread whose float4 stride is a multiple of C/4 always sees the same four
// columns, so it sums 16-byte loads in registers and the workgroup folds its threads per column group through LDS.
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float4* __restrict__ in, int64_t n4, int C4, int threads,
                                                            float* __restrict__ part) {
    __shared__ float4 red[256];
    const int tid = threadIdx.x;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < threads) {
        const int64_t stride = (int64_t)gridDim.x * threads;
        int64_t i = (int64_t)blockIdx.x * threads + tid;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
            s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y); s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; i < n4; i += stride) { const float4 a = in[i]; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
    }
    red[tid] = s;
    __syncthreads();
    if (tid < C4) {                                    // column group g = tid: threads g, g + C4, ... in a fixed order
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = tid; k < threads; k += C4) { t.x += red[k].x; t.y += red[k].y; t.z += red[k].z; t.w += red[k].w; }
        float* o = part + (int64_t)blockIdx.x * C4 * 4 + tid * 4;
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
}

extern "C" int vame_colsum_f32(const float* in, int64_t R, int C, int64_t ld, float* out, int accumulate, float* ws,
                               void* stream) {
    VAME_CHECK_ARG(in && out && ws && R >= 1 && C >= 1, VAME_E_BADARG, "colsum: bad argument");
    const int64_t nslabs = vame_colsum_ws_floats(R, C) / C;
    if (ld == C && C % 4 == 0 && C <= 64 && R >= 4096 && (uintptr_t)in % 16 == 0) {
        const int C4 = C / 4, threads = 256 / C4 * C4;           // (the block's float4 offset blockIdx * threads keeps the column phase)
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3((unsigned)nslabs), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(in), R * C4, C4, threads, ws);
        VAME_LAUNCH_CHECK("colsum narrow");
    } else if (C % 4 == 0 && ld % 4 == 0 && C >= 64 && R >= 1024 && (uintptr_t)in % 16 == 0 && (uintptr_t)ws % 16 == 0) {
        const int64_t rps = cdiv64(R, nslabs);
        hipLaunchKernelGGL(colsum_partial4_kernel, dim3((unsigned)cdiv64(C, 256), (unsigned)nslabs), dim3(256), 0,
                           (hipStream_t)stream, in, R, C, ld, rps, ws);
        VAME_LAUNCH_CHECK("colsum partial4");
    } else {
        const int64_t rps = cdiv64(R, nslabs);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv64(C, 64), (unsigned)nslabs), dim3(256), 0,
                           (hipStream_t)stream, in, R, C, ld, rps, ws);
        VAME_LAUNCH_CHECK("colsum partial");
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv64(C, 64)), dim3(1024), 0, (hipStream_t)stream,
                       (const float*)ws, (int)nslabs, C, out, accumulate);
    VAME_LAUNCH_CHECK("colsum final");
    return VAME_OK;
}

// Batched single-pass column sums for many small jobs (the 24 bias-gradient reductions of a train step) in ONE launch.
struct ColsumJob { const float* in; int64_t R; int64_t C; int64_t ld; float* out; };
struct ColsumBatch { ColsumJob j[32]; };
__global__ __launch_bounds__(256) void colsum_batch_kernel(ColsumBatch P) {
    __shared__ float red[4][64];
    const ColsumJob& J = P.j[blockIdx.y];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    if ((int64_t)blockIdx.x * 64 >= J.C) return;
    float s = 0.f;
    if (c < J.C)
        for (int64_t r = ty; r < J.R; r += 4) s += J.in[r * J.ld + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < J.C) J.out[c] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

extern "C" int vame_colsum_batch_f32(const int64_t* desc, int njobs, void* stream) {
    VAME_CHECK_ARG(desc && njobs >= 1 && njobs <= 32, VAME_E_BADARG, "colsum_batch: njobs=%d not in 1..32", njobs);
    ColsumBatch P;
    int64_t maxc = 0;
    for (int i = 0; i < njobs; ++i) {
        const int64_t* d = desc + 5 * (int64_t)i;
        P.j[i] = {(const float*)d[0], d[1], d[2], d[3], (float*)d[4]};
        VAME_CHECK_ARG(P.j[i].in && P.j[i].out && d[1] >= 1 && d[2] >= 1, VAME_E_BADARG, "colsum_batch: job %d invalid", i);
        if (d[2] > maxc) maxc = d[2];
    }
    hipLaunchKernelGGL(colsum_batch_kernel, dim3((unsigned)cdiv64(maxc, 64), (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, P);
    VAME_LAUNCH_CHECK("colsum_batch");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- sum over time
// blockIdx.y = batch row, threads over the float4 columns; the time loop is unrolled by six loads that are in flight together (the
// rolled form waited for every load before issuing the next: 3.5 TB/s).  The sum runs in time order.
__global__ __launch_bounds__(256) void timesum_kernel(const float* __restrict__ in, int T, int C4, int64_t ld, float* __restrict__ out) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= C4) return;
    const float4* p = reinterpret_cast<const float4*>(in + (int64_t)blockIdx.y * T * ld) + c4;
    const int64_t ld4 = ld / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int t = 0;
    for (; t + 6 <= T; t += 6) {
        float4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = p[(int64_t)(t + k) * ld4];
#pragma unroll
        for (int k = 0; k < 6; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    }
    for (; t < T; ++t) {
        const float4 v = p[(int64_t)t * ld4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(out + (int64_t)blockIdx.y * (C4 * 4))[c4] = s;
}

extern "C" int vame_timesum_f32(const float* in, int B, int T, int C, int64_t ld, float* out, void* stream) {
    VAME_CHECK_ARG(in && out && B >= 1 && T >= 1, VAME_E_BADARG, "timesum: bad argument");
    VAME_CHECK_ARG(C >= 4 && C % 4 == 0 && ld % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0, VAME_E_SHAPE,
                   "timesum: C, ld must be multiples of 4 and pointers 16-byte aligned");
    const int C4 = C / 4, threads = C4 >= 256 ? 256 : (C4 + 63) / 64 * 64;
    for (int b0 = 0; b0 < B; b0 += 65535) {            // grid.y limit
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        hipLaunchKernelGGL(timesum_kernel, dim3((C4 + threads - 1) / threads, nb), dim3(threads), 0, (hipStream_t)stream,
                           in + (int64_t)b0 * T * ld, T, C4, ld, out + (int64_t)b0 * C);
    }
    VAME_LAUNCH_CHECK("timesum");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- Adam (AMSGrad)
// state (device, 4 words, or null): {lr as float, steps applied so far as int, ticket, -}.  Non-null: the learning rate and the step number of
// the bias corrections are read from the DEVICE and the last workgroup to finish counts the step -- nothing in the argument list changes from
// step to step, so a captured graph can replay the launch (and an LR scheduler writes one word); a dropped step (abort_flag) is not counted.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ vmax, int64_t n, float step_size,
                                                   float beta1, float beta2, float eps, float inv_sqrt_bc2, float gscale,
                                                   const int* __restrict__ abort_flag, int* __restrict__ dropped, int* state) {
    // a device-side failure upstream (a cooperative GRU launch that gave up waiting: gru_coop.hip) must not reach the weights:
    // the step is dropped here, on the device, and the host raises when it next looks at the same word; `dropped` counts the
    // launches that did nothing, so the host can take them out of its bias-correction step count
    if (abort_flag && *abort_flag != 0) {
        if (dropped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(dropped, 1);
        return;
    }
    int t = 0;
    if (state) {
        // the two double-precision pow() of the bias corrections once per workgroup (every thread doing them cost more than the update itself)
        __shared__ float sc[2];
        if (threadIdx.x == 0) {
            t = state[1] + 1;
            const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
            sc[0] = (float)((double)__builtin_bit_cast(float, state[0]) / bc1);
            sc[1] = (float)(1.0 / sqrt(bc2));
        }
        __syncthreads();
        step_size = sc[0];
        inv_sqrt_bc2 = sc[1];
    }
    auto upd = [&](float& pi, float gi, float& mi, float& vi, float& vmi) {
        gi *= gscale;
        mi = beta1 * mi + (1.0f - beta1) * gi;
        vi = beta2 * vi + (1.0f - beta2) * gi * gi;
        vmi = fmaxf(vmi, vi);
        const float denom = sqrtf(vmi) * inv_sqrt_bc2 + eps;
        pi -= step_size * (mi / denom);
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                       reinterpret_cast<uintptr_t>(vmax)) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t i = i0; i < n4; i += stride) {           // 16 bytes per lane and stream; the same arithmetic per element
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i],
               vx = reinterpret_cast<float4*>(vmax)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        upd(pp.x, gg.x, mm.x, vv.x, vx.x); upd(pp.y, gg.y, mm.y, vv.y, vx.y); upd(pp.z, gg.z, mm.z, vv.z, vx.z); upd(pp.w, gg.w, mm.w, vv.w, vx.w);
        reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv; reinterpret_cast<float4*>(vmax)[i] = vx;
        reinterpret_cast<float4*>(p)[i] = pp;
    }
    for (int64_t i = 4 * n4 + i0; i < n; i += stride) upd(p[i], g[i], m[i], v[i], vmax[i]);
    if (state) {
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(state + 2, 1) == (int)gridDim.x - 1) { state[2] = 0; state[1] = t; }
    }
}

extern "C" int vame_adam_amsgrad_f32(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr,
                                     float beta1, float beta2, float eps, int step, float gscale, const int* abort_flag, int* dropped,
                                     int* state, void* stream) {
    VAME_CHECK_ARG(p && g && m && v && vmax && n >= 1 && (state || step >= 1), VAME_E_BADARG, "adam: bad argument");
    const double bc1 = state ? 1.0 : 1.0 - pow((double)beta1, step), bc2 = state ? 1.0 : 1.0 - pow((double)beta2, step);
    // one pass of 16-byte accesses over the whole grid (a capped grid-stride grid left a second pass with a quarter of the threads: 34 us for 94 MB)
    const int64_t quads = (n + 3) / 4;
    const unsigned blocks = (unsigned)(quads / 256 + 1 > 65535 * 4 ? 65535 * 4 : quads / 256 + 1);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, n,
                       (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), gscale, abort_flag, dropped, state);
    VAME_LAUNCH_CHECK("adam");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- indexed copy (padded parameter images)
// dst[dst_idx[i]] = src[src_idx[i]]: moves the model's parameters into the zero-padded image the kernels run on when a hidden
// size is not a multiple of 32 (and the padded gradients back); both index lists are ascending, so accesses stay nearly coalesced.
__global__ __launch_bounds__(256) void index_copy_kernel(float* __restrict__ dst, const int64_t* __restrict__ dst_idx,
                                                         const float* __restrict__ src, const int64_t* __restrict__ src_idx, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[dst_idx[i]] = src[src_idx[i]];
}
extern "C" int vame_index_copy_f32(float* dst, const int64_t* dst_idx, const float* src, const int64_t* src_idx, int64_t n, void* stream) {
    VAME_CHECK_ARG(dst && dst_idx && src && src_idx && n >= 0, VAME_E_BADARG, "index_copy: bad argument");
    if (n == 0) return VAME_OK;
    hipLaunchKernelGGL(index_copy_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dst, dst_idx, src, src_idx, n);
    VAME_LAUNCH_CHECK("index_copy");
    return VAME_OK;
}

// Up to 8 Linear layers of ONE narrow input in one launch: C_g (M, N_g) = A (M, K) W_g^T (N_g, K) + bias_g, K <= 32.  The decoders' projections of z
// (rnn_model.py:103-106, 136-140: latent_to_hidden and the GRUs' W_ih applied to the time-constant input z -- six Linear layers of the same
// (B, zdims) matrix) were six launches of the MFMA GEMM whose k loop is a single short tile: 17-23 us each for 5-13 MB of output.  With K <= 32 an
// output needs <= 32 FMAs: this is a store stream.  Workgroup = 32 rows x 256 columns of one problem (A tile and W tile in LDS, k-major), thread =
// 8 rows x 4 columns; k-ordered fmaf chains from 0, bias added last.
struct LinGroupParams {
    const float* A; int64_t lda; int M, K, count;
    const float* W[8]; const float* bias[8]; float* C[8]; int64_t ldc[8]; int N[8];
    int tile0[9];                      // first 256-column tile of problem g in blockIdx.x
};

__global__ __launch_bounds__(256) void linear_group_kernel(LinGroupParams P) {
    __shared__ float As[32][32 + 4];           // [k][row]
    __shared__ float Ws[32][256 + 4];          // [k][column]
    const int tid = threadIdx.x;
    int g = 0;
    while (g + 1 < P.count && (int)blockIdx.x >= P.tile0[g + 1]) ++g;
    const int n0 = ((int)blockIdx.x - P.tile0[g]) * 256, m0 = blockIdx.y * 32, K = P.K, N = P.N[g];
    for (int i = tid; i < 32 * K; i += 256) {                       // A tile: consecutive threads walk a row
        const int r = i / K, k = i - r * K, m = m0 + r;
        As[k][r] = m < P.M ? P.A[(int64_t)m * P.lda + k] : 0.f;
    }
    {
        const int n = n0 + tid;                                     // W tile: one row (K contiguous floats) per thread
        const float* wr = P.W[g] + (int64_t)(n < N ? n : N - 1) * K;
        for (int k = 0; k < K; ++k) Ws[k][tid] = wr[k];
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    float acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(&Ws[k][4 * tx]);
        const float4 a0 = *reinterpret_cast<const float4*>(&As[k][8 * ty]), a1 = *reinterpret_cast<const float4*>(&As[k][8 * ty + 4]);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(av[r], wv[c], acc[r][c]);
    }
    const int n = n0 + 4 * tx;
    if (n >= N) return;
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.bias[g]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = n + c < N ? P.bias[g][n + c] : 0.f;
    }
    const int64_t ldc = P.ldc[g];
    const bool vec = n + 4 <= N && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(P.C[g]) & 15) == 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int m = m0 + 8 * ty + r;
        if (m >= P.M) break;
        float* out = P.C[g] + (int64_t)m * ldc + n;
        if (vec) *reinterpret_cast<float4*>(out) = make_float4(acc[r][0] + b[0], acc[r][1] + b[1], acc[r][2] + b[2], acc[r][3] + b[3]);
        else
            for (int c = 0; c < 4 && n + c < N; ++c) out[c] = acc[r][c] + b[c];
    }
}

extern "C" int vame_linear_group_f32(int count, int M, int K, const float* A, int64_t lda, const float* const* W, const float* const* bias,
                                     float* const* C, const int64_t* ldc, const int* N, void* stream) {
    VAME_CHECK_ARG(count >= 1 && count <= 8 && M >= 1 && K >= 1 && K <= 32 && A && W && bias && C && ldc && N && lda >= K, VAME_E_SHAPE,
                   "linear_group: count=%d (1..8) M=%d K=%d (1..32)", count, M, K);
    LinGroupParams P;
    P.A = A; P.lda = lda; P.M = M; P.K = K; P.count = count;
    int tiles = 0;
    for (int g = 0; g < 8; ++g) {
        const bool on = g < count;
        VAME_CHECK_ARG(!on || (W[g] && C[g] && N[g] >= 1 && ldc[g] >= N[g]), VAME_E_BADARG, "linear_group: problem %d: null operand or ldc < N", g);
        P.W[g] = on ? W[g] : nullptr; P.bias[g] = on ? bias[g] : nullptr; P.C[g] = on ? C[g] : nullptr; P.ldc[g] = on ? ldc[g] : 0; P.N[g] = on ? N[g] : 0;
        P.tile0[g] = tiles;
        if (on) tiles += (N[g] + 255) / 256;
    }
    P.tile0[8] = tiles;
    for (int g = count; g < 8; ++g) P.tile0[g] = tiles;
    hipLaunchKernelGGL(linear_group_kernel, dim3((unsigned)tiles, (unsigned)cdiv64(M, 32)), dim3(256), 0, (hipStream_t)stream, P);
    VAME_LAUNCH_CHECK("linear_group");
    return VAME_OK;
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float a, float* __restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}
extern "C" int vame_axpy_f32(const float* x, float a, float* y, int64_t n, void* stream) {
    VAME_CHECK_ARG(x && y && n >= 1, VAME_E_BADARG, "axpy: bad argument");
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, a, y, n);
    VAME_LAUNCH_CHECK("axpy");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- dropout mask (encoder inter-layer dropout)
// out[r][c] = x[row(r)][c] * mask[r][c] * scale over R x C (C % 4 == 0, 16-byte aligned rows); x rows are addressed in two
// levels (r / seg) * seg_stride + (r % seg) * ld + off like the GEMM operands (the padded (B, T+2, 2H) sequence layout); seg = 0:
// dense rows of ld elements.  In-place (out == x, dense) is allowed.  torch.nn.GRU(dropout=p) semantics (rnn_model.py:34-35):
// mask in {0,1}, scale = 1/(1-p).
__global__ __launch_bounds__(256) void mask_scale_kernel(const float* __restrict__ x, int64_t off, int64_t ld, int64_t seg, int64_t seg_stride,
                                                         const float* __restrict__ mask, float scale, float* __restrict__ out, int64_t R, int C) {
    const int cq = C / 4;
    const int64_t n = R * cq;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cq;
        const int c = (int)(i % cq) * 4;
        const int64_t xo = (seg ? (r / seg) * seg_stride + (r % seg) * ld : r * ld) + off + c;
        const float4 v = *reinterpret_cast<const float4*>(x + xo);
        const float4 m = *reinterpret_cast<const float4*>(mask + r * C + c);
        *reinterpret_cast<float4*>(out + r * C + c) = make_float4(v.x * m.x * scale, v.y * m.y * scale, v.z * m.z * scale, v.w * m.w * scale);
    }
}
extern "C" int vame_mask_scale_f32(const float* x, int64_t off, int64_t ld, int64_t seg, int64_t seg_stride, const float* mask, float scale,
                                   float* out, int64_t R, int C, void* stream) {
    VAME_CHECK_ARG(x && mask && out && R >= 1 && C >= 4, VAME_E_BADARG, "mask_scale: bad argument");
    VAME_CHECK_ARG(C % 4 == 0 && ld % 4 == 0 && off % 4 == 0 && seg_stride % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)mask % 16 == 0 &&
                   (uintptr_t)out % 16 == 0, VAME_E_SHAPE, "mask_scale: rows must be 16-byte aligned (C=%d)", C);
    hipLaunchKernelGGL(mask_scale_kernel, dim3(ew_blocks(R * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, off, ld, seg, seg_stride, mask,
                       scale, out, R, C);
    VAME_LAUNCH_CHECK("mask_scale");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- per-step GRU cell (large-H path)
// For hidden sizes beyond the persistent sequence kernels (H > 256) the recurrence runs step by step: the gate GEMM
// h_{t-1} W_hh^T (M = batch, a real dense contraction at those sizes) goes through vame_gemm_f32 and these two kernels
// do the gate math.  Same stash contents (cA, cB, u, r, gh_n) as the sequence kernels, row-major (B, 5H) per step.
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gi, int64_t gi_row, const float* __restrict__ gh,
                                                           const float* __restrict__ bhn, const float* __restrict__ hprev,
                                                           int64_t hp_row, float* __restrict__ hout, int64_t ho_row,
                                                           float* __restrict__ stash, int64_t st_row, int B, int H) {
    const int64_t n = (int64_t)B * H;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i % H);
        const float* g = gi + b * gi_row + j;
        const float* q = gh + b * 3 * H + j;
        const float ghn = q[2 * H] + bhn[j];
        const float r = fast_sigmoid(g[0] + q[0]);
        const float u = fast_sigmoid(g[H] + q[H]);
        const float nn = fast_tanh(g[2 * H] + r * ghn);
        const float hp = hprev ? hprev[b * hp_row + j] : 0.f;
        hout[b * ho_row + j] = nn + u * (hp - nn);
        if (stash) {
            float* s = stash + b * st_row + j;
            const float omu = 1.0f - u;
            s[0] = omu * (1.0f - nn * nn); s[H] = (hp - nn) * u * omu; s[2 * H] = u; s[3 * H] = r; s[4 * H] = ghn;
        }
    }
}

extern "C" int vame_gru_cell_fwd_f32(const float* gi, int64_t gi_row, const float* gh, const float* bhn, const float* hprev,
                                     int64_t hp_row, float* hout, int64_t ho_row, float* stash, int64_t st_row, int B, int H,
                                     void* stream) {
    VAME_CHECK_ARG(gi && gh && bhn && hout && B >= 1 && H >= 1, VAME_E_BADARG, "gru_cell_fwd: bad argument");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(ew_blocks((int64_t)B * H)), dim3(256), 0, (hipStream_t)stream, gi, gi_row, gh, bhn,
                       hprev, hp_row, hout, ho_row, stash, st_row, B, H);
    VAME_LAUNCH_CHECK("gru_cell_fwd");
    return VAME_OK;
}

// dh (B,H): in = gradient wrt h_t carried from the later steps, out = d*u (the part flowing through the update gate;
// the caller adds dgh W_hh with an accumulating GEMM).  dG row = [da_r | da_z | dgi_n | dgh_n], dgh (B,3H) = [da_r|da_z|dgh_n].
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ stash, int64_t st_row, float* __restrict__ dh,
                                                           const float* __restrict__ dy, int64_t dy_row, float* __restrict__ dG,
                                                           int64_t dg_row, float* __restrict__ dgh, int B, int H) {
    const int64_t n = (int64_t)B * H;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i % H);
        const float* s = stash + b * st_row + j;
        const float d = dh[i] + (dy ? dy[b * dy_row + j] : 0.f);
        const float r = s[3 * H];
        const float dan = d * s[0], dau = d * s[H], dghn = dan * r, dar = dghn * s[4 * H] * (1.0f - r);
        dh[i] = d * s[2 * H];
        float* o = dG + b * dg_row + j;
        o[0] = dar; o[H] = dau; o[2 * H] = dan; o[3 * H] = dghn;
        float* q = dgh + b * 3 * H + j;
        q[0] = dar; q[H] = dau; q[2 * H] = dghn;
    }
}

extern "C" int vame_gru_cell_bwd_f32(const float* stash, int64_t st_row, float* dh, const float* dy, int64_t dy_row, float* dG,
                                     int64_t dg_row, float* dgh, int B, int H, void* stream) {
    VAME_CHECK_ARG(stash && dh && dG && dgh && B >= 1 && H >= 1, VAME_E_BADARG, "gru_cell_bwd: bad argument");
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(ew_blocks((int64_t)B * H)), dim3(256), 0, (hipStream_t)stream, stash, st_row, dh, dy,
                       dy_row, dG, dg_row, dgh, B, H);
    VAME_LAUNCH_CHECK("gru_cell_bwd");
    return VAME_OK;
}

// The same step with the stash in the ACCUMULATOR-FRAGMENT order the persistent forward kernels write (gru_seq.hip / gru_wide.hip):
// float4 index ((((tile*T + t)*NB + wb)*5 + k)*4 + rq)*64 + lane holds rows 8*rq + 4*(lane>>5) + (0..3) of column 32*wb + (lane&31).
// One thread per float4 slot: five coalesced 16-byte loads, then four rows of dh / dy / dG / dgh (128-byte segments per half wave).
__global__ __launch_bounds__(256) void gru_cell_bwd_frag_kernel(const float4* __restrict__ stash, int T, int t, float* __restrict__ dh,
                                                                const float* __restrict__ dy, int64_t dy_row, float* __restrict__ dG,
                                                                int64_t dg_row, float* __restrict__ dgh, int B, int H) {
    const int NB = H / 32;
    const int64_t n = (int64_t)((B + 31) / 32) * NB * 4 * 64;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), rq = (int)((i >> 6) & 3);
        const int64_t tw = i >> 8;
        const int wb = (int)(tw % NB);
        const int64_t tile = tw / NB;
        const int j = 32 * wb + (lane & 31);
        const float4* sp = stash + ((((tile * T + t) * NB + wb) * 5) * 4 + rq) * 64 + lane;
        const float4 cA = sp[0], cB = sp[4 * 64], u = sp[8 * 64], r4 = sp[12 * 64], g4 = sp[16 * 64];
        const float a_[4] = {cA.x, cA.y, cA.z, cA.w}, b_[4] = {cB.x, cB.y, cB.z, cB.w}, u_[4] = {u.x, u.y, u.z, u.w},
                    r_[4] = {r4.x, r4.y, r4.z, r4.w}, g_[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t b = tile * 32 + 8 * rq + 4 * (lane >> 5) + e;
            if (b >= B) continue;
            const float d = dh[b * H + j] + (dy ? dy[b * dy_row + j] : 0.f);
            const float dan = d * a_[e], dau = d * b_[e], dghn = dan * r_[e], dar = dghn * g_[e] * (1.0f - r_[e]);
            dh[b * H + j] = d * u_[e];
            float* o = dG + b * dg_row + j;
            o[0] = dar; o[H] = dau; o[2 * H] = dan; o[3 * H] = dghn;
            float* q = dgh + b * 3 * H + j;
            q[0] = dar; q[H] = dau; q[2 * H] = dghn;
        }
    }
}

extern "C" int vame_gru_cell_bwd_frag_f32(const float* stash, int T, int t, float* dh, const float* dy, int64_t dy_row, float* dG, int64_t dg_row,
                                          float* dgh, int B, int H, void* stream) {
    VAME_CHECK_ARG(stash && dh && dG && dgh && B >= 1 && H >= 32 && H % 32 == 0 && t >= 0 && t < T, VAME_E_BADARG, "gru_cell_bwd_frag: bad argument");
    VAME_CHECK_ARG((uintptr_t)stash % 16 == 0, VAME_E_SHAPE, "gru_cell_bwd_frag: stash must be 16-byte aligned");
    const int64_t n = (int64_t)((B + 31) / 32) * (H / 32) * 4 * 64;
    hipLaunchKernelGGL(gru_cell_bwd_frag_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(stash), T, t, dh,
                       dy, dy_row, dG, dg_row, dgh, B, H);
    VAME_LAUNCH_CHECK("gru_cell_bwd_frag");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- k-means E-step
// Nearest-centre assignment over the (N, D) latent vectors (vame/analysis/pose_segmentation.py:141,179: sklearn KMeans on the
// embedding).  HBM-bound scan: 4*D bytes in per row; rows are staged through LDS with coalesced loads, the K centres live in
// LDS, each thread scores one row.  Writes label, squared distance to it and (optionally) a one-hot row so the M-step is a
// deterministic split-K MFMA GEMM  sums = onehot^T X  (vame_gemm_f32) instead of float atomics.
#define KM_ROWS 256
// DP = feature count padded to a multiple of 4 (<= 64): the row lives in registers, centres are read from LDS as
// broadcast float4's -> ~1 LDS instruction and 8 VALU per (centre, 4 features) per wave
template <int DP>
__global__ __launch_bounds__(KM_ROWS) void kmeans_assign_kernel(const float* __restrict__ X, int64_t N, int D,
                                                                const float* __restrict__ C, int K, int Kp,
                                                                int* __restrict__ labels, float* __restrict__ mind2,
                                                                float* __restrict__ onehot) {
    VAME_DYN_SMEM(smem_raw);
    float* cs = reinterpret_cast<float*>(smem_raw);          // [K][DP], zero padded
    float* xs = cs + K * DP;                                   // [KM_ROWS][DP + 1]
    constexpr int LX = DP + 1;
    for (int i = threadIdx.x; i < K * DP; i += KM_ROWS) { const int k = i / DP, j = i % DP; cs[i] = j < D ? C[k * D + j] : 0.f; }
    for (int64_t r0 = (int64_t)blockIdx.x * KM_ROWS; r0 < N; r0 += (int64_t)gridDim.x * KM_ROWS) {
        const int nr = (int)(N - r0 < KM_ROWS ? N - r0 : KM_ROWS);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * D; i += KM_ROWS) xs[(i / D) * LX + i % D] = X[r0 * D + i];      // contiguous block, coalesced
        __syncthreads();
        if ((int)threadIdx.x < nr) {
            float x[DP];
#pragma unroll
            for (int j = 0; j < DP; ++j) x[j] = j < D ? xs[threadIdx.x * LX + j] : 0.f;
            float best = 3.4e38f;
            int bk = 0;
            for (int k = 0; k < K; ++k) {
                const float4* c4 = reinterpret_cast<const float4*>(&cs[k * DP]);
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int q = 0; q < DP / 4; ++q) {
                    const float4 c = c4[q];
                    const float t0 = x[4 * q] - c.x, t1 = x[4 * q + 1] - c.y, t2 = x[4 * q + 2] - c.z, t3 = x[4 * q + 3] - c.w;
                    d0 = fmaf(t0, t0, d0); d1 = fmaf(t1, t1, d1); d0 = fmaf(t2, t2, d0); d1 = fmaf(t3, t3, d1);
                }
                const float d2 = d0 + d1;
                if (d2 < best) { best = d2; bk = k; }            // ties: lowest index, like argmin
            }
            const int64_t r = r0 + threadIdx.x;
            labels[r] = bk;
            if (mind2) mind2[r] = best;
            if (onehot) {
                float4* o = reinterpret_cast<float4*>(onehot + r * Kp);
                for (int k4 = 0; k4 < Kp / 4; ++k4)
                    o[k4] = make_float4(4 * k4 == bk ? 1.f : 0.f, 4 * k4 + 1 == bk ? 1.f : 0.f, 4 * k4 + 2 == bk ? 1.f : 0.f, 4 * k4 + 3 == bk ? 1.f : 0.f);
            }
        }
    }
}

extern "C" int vame_kmeans_assign_f32(const float* X, int64_t N, int D, const float* C, int K, int* labels, float* mind2,
                                      float* onehot, int Kp, void* stream) {
    VAME_CHECK_ARG(X && C && labels && N >= 1 && D >= 1 && K >= 1, VAME_E_BADARG, "kmeans_assign: bad argument");
    VAME_CHECK_ARG(!onehot || (Kp >= K && Kp % 4 == 0 && (uintptr_t)onehot % 16 == 0), VAME_E_SHAPE,
                   "kmeans_assign: one-hot width %d must be a multiple of 4 >= K=%d, 16-byte aligned", Kp, K);
    VAME_CHECK_ARG(D <= 64, VAME_E_UNSUPPORTED, "kmeans_assign: D=%d > 64 features", D);
    const int DP = D <= 32 ? 32 : 64;
    const size_t sh = ((size_t)K * DP + (size_t)KM_ROWS * (DP + 1)) * 4;
    VAME_CHECK_ARG(sh <= 150 * 1024, VAME_E_SHAPE, "kmeans_assign: K*D too large for LDS");
    const int64_t nb = cdiv64(N, KM_ROWS);
    const dim3 grid((unsigned)(nb < 4096 ? nb : 4096));
    if (DP == 32) hipLaunchKernelGGL(kmeans_assign_kernel<32>, grid, dim3(KM_ROWS), sh, (hipStream_t)stream, X, N, D, C, K, Kp, labels, mind2, onehot);
    else hipLaunchKernelGGL(kmeans_assign_kernel<64>, grid, dim3(KM_ROWS), sh, (hipStream_t)stream, X, N, D, C, K, Kp, labels, mind2, onehot);
    VAME_LAUNCH_CHECK("kmeans_assign");
    return VAME_OK;
}

// --------------------------------------------------------------------------------- nuclear norm
// Symmetric eigen-decomposition of G/bsize (Z<=64) by parallel-ordered cyclic Jacobi in fp64: a
// round-robin schedule gives Z/2 disjoint rotations per round, applied as a column pass and a row
// pass.  One 256-thread workgroup; latency ~0.1 ms, run beside the decoder kernels.
#define NUC_MAXZ 64
#define NUC_BLK ((NUC_MAXZ / 2) * (NUC_MAXZ / 2) / 256)   /* 2x2 rotation blocks per thread at Z = NUC_MAXZ */
#ifndef NUC_TOL
#define NUC_TOL 1e-13   /* off-diagonal mass / diagonal mass at which the sweeps stop: eigenvalues to ~1e-13, eigenvectors to ~3e-7 relative -- below the fp32 Gram it starts from; 1e-20 costs one more sweep (+35 us) for the same loss */
#endif
// vstate (optional, Z'xZ' doubles with Z' = Z rounded up to even, zero-initialised by the caller): eigenvectors of the
// previous call.  G changes little between optimizer steps, so rotating into the previous eigenbasis first (A = V^T G V)
// leaves an almost diagonal matrix and the Jacobi iteration converges in 1-2 sweeps instead of 6-8.
__global__ __launch_bounds__(256) void nuclear_kernel(const float* __restrict__ G, int Z, int kloss, int nrows, float lmbda,
                                                      float bsize, float gscale, float* __restrict__ loss_out,
                                                      float* __restrict__ Minv, double* __restrict__ vstate) {
    __shared__ double A[NUC_MAXZ * NUC_MAXZ];
    __shared__ double V[NUC_MAXZ * NUC_MAXZ];
    __shared__ double cs[NUC_MAXZ];        // c at [k], s at [k + 32]
    __shared__ int pq[NUC_MAXZ];           // p at [k], q at [k + 32]
    __shared__ double wsel[NUC_MAXZ];
    const int tid = threadIdx.x, n = Z + (Z & 1), np = n / 2;
    __shared__ int warm;
    if (tid == 0) warm = (vstate != nullptr && vstate[0] != 0.0);      // a used state has a non-zero (0,0) entry w.p. 1
    for (int i = tid; i < n * n; i += 256) {
        const int r = i / n, c = i % n;
        A[i] = (r < Z && c < Z) ? 0.5 * ((double)G[r * Z + c] + (double)G[c * Z + r]) / (double)bsize : 0.0;
        V[i] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (warm) {
        __shared__ double Tm[NUC_MAXZ * NUC_MAXZ];
        for (int i = tid; i < n * n; i += 256) V[i] = vstate[i];
        __syncthreads();
        for (int i = tid; i < n * n; i += 256) {          // T = A V
            const int r = i / n, c = i % n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[r * n + k] * V[k * n + c];
            Tm[i] = s;
        }
        __syncthreads();
        for (int i = tid; i < n * n; i += 256) {          // A = V^T T (symmetrised)
            const int r = i / n, c = i % n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += V[k * n + r] * Tm[k * n + c];
            A[i] = s;
        }
        __syncthreads();
        for (int i = tid; i < n * n; i += 256) {
            const int r = i / n, c = i % n;
            if (r < c) { const double m = 0.5 * (A[r * n + c] + A[c * n + r]); Tm[i] = m; } else Tm[i] = A[i];
        }
        __syncthreads();
        for (int i = tid; i < n * n; i += 256) { const int r = i / n, c = i % n; A[i] = r <= c ? Tm[r * n + c] : Tm[c * n + r]; }
        __syncthreads();
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = tid; i < n * n; i += 256) {
            const int r = i / n, c = i % n;
            const double a = A[i];
            if (r == c) dg += a * a; else off += a * a;
        }
        __shared__ double ro[256], rd[256];   // block reduce (fp64) through LDS
        ro[tid] = off; rd[tid] = dg;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (tid < s) { ro[tid] += ro[tid + s]; rd[tid] += rd[tid + s]; }
            __syncthreads();
        }
        const bool done = ro[0] <= NUC_TOL * rd[0] || rd[0] == 0.0;      // |off| <= sqrt(NUC_TOL) |diag|: eigenvalue error ~ off^2 / gap
        __syncthreads();
        if (done) break;
        for (int round = 0; round < n - 1; ++round) {
            if (tid < np) {
                int p, q;
                if (tid == 0) { p = n - 1; q = round; }
                else { p = (round + tid) % (n - 1); q = (round - tid + (n - 1)) % (n - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double app = A[p * n + p], aqq = A[q * n + q], apq = A[p * n + q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-300) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                }
                cs[tid] = c; cs[tid + 32] = s; pq[tid] = p; pq[tid + 32] = q;
            }
            __syncthreads();
            // A <- J^T A J in ONE pass: the pairs are disjoint, so the 2x2 block (pair P, pair Q) of the result depends only on
            // the same block of A: thread (P,Q) loads it, rotates rows by J_P and columns by J_Q, and writes it back in place
            // (np * np <= 1024 blocks for Z <= 64: up to NUC_BLK per thread, all read before the barrier, all written after it)
            double blk[NUC_BLK][4];
            int bidx[NUC_BLK][4];
#pragma unroll
            for (int u = 0; u < NUC_BLK; ++u) {
                const int b = tid + u * 256;
                if (b < np * np) {
                    const int P = b / np, Q = b % np;
                    const int bp0 = pq[P], bp1 = pq[P + 32], bq0 = pq[Q], bq1 = pq[Q + 32];
                    const double cp = cs[P], sp = cs[P + 32], cq = cs[Q], sq = cs[Q + 32];
                    bidx[u][0] = bp0 * n + bq0; bidx[u][1] = bp0 * n + bq1; bidx[u][2] = bp1 * n + bq0; bidx[u][3] = bp1 * n + bq1;
                    const double a00 = A[bidx[u][0]], a01 = A[bidx[u][1]], a10 = A[bidx[u][2]], a11 = A[bidx[u][3]];
                    const double r00 = cp * a00 - sp * a10, r01 = cp * a01 - sp * a11;       // rows: J_P^T
                    const double r10 = sp * a00 + cp * a10, r11 = sp * a01 + cp * a11;
                    blk[u][0] = cq * r00 - sq * r01; blk[u][1] = sq * r00 + cq * r01;         // columns: J_Q
                    blk[u][2] = cq * r10 - sq * r11; blk[u][3] = sq * r10 + cq * r11;
                }
            }
            for (int i = tid; i < np * n; i += 256) {       // V <- V J (columns; disjoint pairs, in place)
                const int k = i / n, r = i % n;
                const double c = cs[k], s = cs[k + 32];
                const int p = pq[k], q = pq[k + 32];
                const double vp = V[r * n + p], vq = V[r * n + q];
                V[r * n + p] = c * vp - s * vq; V[r * n + q] = s * vp + c * vq;
            }
            __syncthreads();                                  // every block has been read before any is overwritten
#pragma unroll
            for (int u = 0; u < NUC_BLK; ++u)
                if (tid + u * 256 < np * np) {
                    A[bidx[u][0]] = blk[u][0]; A[bidx[u][1]] = blk[u][1]; A[bidx[u][2]] = blk[u][2]; A[bidx[u][3]] = blk[u][3];
                }
            __syncthreads();
        }
    }
    if (vstate) {
        for (int i = tid; i < n * n; i += 256) vstate[i] = V[i];
    }
    // select the top k_eff eigenvalues: the (B,B) Gram of the reference has min(B,Z) non-zero ones
    int keff = kloss < Z ? kloss : Z;
    if (nrows < keff) keff = nrows;
    if (tid < Z) {
        const double w = A[tid * n + tid];
        int rank = 0;
        for (int j = 0; j < Z; ++j) {
            const double wj = A[j * n + j];
            rank += (wj > w) || (wj == w && j < tid);
        }
        const double sv = (rank < keff && w > 0.0) ? sqrt(w) : 0.0;
        wsel[tid] = sv;
        cs[tid] = sv > 0.0 ? 1.0 / sv : 0.0;                 // one division per eigenvalue, not one per term of the Minv sums below
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int j = 0; j < Z; ++j) s += wsel[j];
        loss_out[0] = (float)(lmbda * s);
    }
    if (Minv) {
        const double scale = (double)gscale * (double)lmbda / (double)bsize;
        for (int i = tid; i < Z * Z; i += 256) {
            const int a = i / Z, b = i % Z;
            double s = 0.0;
            for (int j = 0; j < Z; ++j) s += V[a * n + j] * V[b * n + j] * cs[j];
            Minv[i] = (float)(scale * s);
        }
    }
}

// Same decomposition for 64 < Z <= 512 (configurations far from VAME's zdims = 30 default): the matrices no longer fit the
// LDS, so A, V and a scratch image live in a caller-provided state buffer st = [V | A | T] (3 Z'^2 doubles, L2-resident; V
// doubles as the warm-start state exactly as above and must be zero on first use).  One 1024-thread workgroup; global
// traffic between threads of the workgroup is ordered by the barriers.  Latency is milliseconds, not 0.1 ms.
#define NUCB_MAXZ 512
#define NUCB_THREADS 1024
__global__ __launch_bounds__(NUCB_THREADS) void nuclear_big_kernel(const float* __restrict__ G, int Z, int kloss, int nrows, float lmbda,
                                                                   float bsize, float gscale, float* __restrict__ loss_out,
                                                                   float* __restrict__ Minv, double* st) {
    __shared__ double cs[NUCB_MAXZ];       // c at [k], s at [k + NUCB_MAXZ / 2]
    __shared__ int pq[NUCB_MAXZ];
    __shared__ double wsel[NUCB_MAXZ];
    __shared__ double ro[NUCB_THREADS], rd[NUCB_THREADS];
    __shared__ int warm;
    constexpr int HB = NUCB_MAXZ / 2, NT = NUCB_THREADS;
    const int tid = threadIdx.x, n = Z + (Z & 1), np = n / 2, nn = n * n;
    double* V = st;
    double* A = st + nn;
    double* Tm = st + 2 * nn;
    if (tid == 0) warm = st[0] != 0.0;
    __syncthreads();
    for (int i = tid; i < nn; i += NT) {
        const int r = i / n, c = i % n;
        A[i] = (r < Z && c < Z) ? 0.5 * ((double)G[r * Z + c] + (double)G[c * Z + r]) / (double)bsize : 0.0;
        if (!warm) V[i] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (warm) {
        for (int i = tid; i < nn; i += NT) {              // T = A V
            const int r = i / n, c = i % n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[r * n + k] * V[k * n + c];
            Tm[i] = s;
        }
        __syncthreads();
        for (int i = tid; i < nn; i += NT) {              // A = V^T T
            const int r = i / n, c = i % n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += V[k * n + r] * Tm[k * n + c];
            A[i] = s;
        }
        __syncthreads();
        for (int i = tid; i < nn; i += NT) {              // symmetrise through T
            const int r = i / n, c = i % n;
            Tm[i] = 0.5 * (A[r * n + c] + A[c * n + r]);
        }
        __syncthreads();
        for (int i = tid; i < nn; i += NT) A[i] = Tm[i];
        __syncthreads();
    }
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = tid; i < nn; i += NT) {
            const int r = i / n, c = i % n;
            const double a = A[i];
            if (r == c) dg += a * a; else off += a * a;
        }
        ro[tid] = off; rd[tid] = dg;
        __syncthreads();
        for (int s = NT / 2; s >= 1; s >>= 1) {
            if (tid < s) { ro[tid] += ro[tid + s]; rd[tid] += rd[tid + s]; }
            __syncthreads();
        }
        const bool done = ro[0] <= NUC_TOL * rd[0] || rd[0] == 0.0;
        __syncthreads();
        if (done) break;
        for (int round = 0; round < n - 1; ++round) {
            if (tid < np) {
                int p, q;
                if (tid == 0) { p = n - 1; q = round; }
                else { p = (round + tid) % (n - 1); q = (round - tid + (n - 1)) % (n - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double app = A[p * n + p], aqq = A[q * n + q], apq = A[p * n + q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-300) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                }
                cs[tid] = c; cs[tid + HB] = s; pq[tid] = p; pq[tid + HB] = q;
            }
            __syncthreads();
            for (int b = tid; b < np * np; b += NT) {          // A <- J^T A J, one 2x2 block (pair P, pair Q) at a time, out of place
                const int P = b / np, Q = b % np;
                const int bp0 = pq[P], bp1 = pq[P + HB], bq0 = pq[Q], bq1 = pq[Q + HB];
                const double cp = cs[P], sp = cs[P + HB], cq = cs[Q], sq = cs[Q + HB];
                const int i00 = bp0 * n + bq0, i01 = bp0 * n + bq1, i10 = bp1 * n + bq0, i11 = bp1 * n + bq1;
                const double a00 = A[i00], a01 = A[i01], a10 = A[i10], a11 = A[i11];
                const double r00 = cp * a00 - sp * a10, r01 = cp * a01 - sp * a11;
                const double r10 = sp * a00 + cp * a10, r11 = sp * a01 + cp * a11;
                Tm[i00] = cq * r00 - sq * r01; Tm[i01] = sq * r00 + cq * r01;
                Tm[i10] = cq * r10 - sq * r11; Tm[i11] = sq * r10 + cq * r11;
            }
            for (int i = tid; i < np * n; i += NT) {           // V <- V J (columns; disjoint pairs, in place)
                const int k = i / n, r = i % n;
                const double c = cs[k], s = cs[k + HB];
                const int p = pq[k], q = pq[k + HB];
                const double vp = V[r * n + p], vq = V[r * n + q];
                V[r * n + p] = c * vp - s * vq; V[r * n + q] = s * vp + c * vq;
            }
            __syncthreads();
            { double* t = A; A = Tm; Tm = t; }                 // every entry belongs to exactly one block: the scratch image is complete
        }
    }
    int keff = kloss < Z ? kloss : Z;
    if (nrows < keff) keff = nrows;
    for (int i = tid; i < Z; i += NT) {
        const double w = A[i * n + i];
        int rank = 0;
        for (int j = 0; j < Z; ++j) {
            const double wj = A[j * n + j];
            rank += (wj > w) || (wj == w && j < i);
        }
        wsel[i] = (rank < keff && w > 0.0) ? sqrt(w) : 0.0;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int j = 0; j < Z; ++j) s += wsel[j];
        loss_out[0] = (float)(lmbda * s);
    }
    if (Minv) {
        for (int i = tid; i < Z * Z; i += NT) {
            const int a = i / Z, b = i % Z;
            double s = 0.0;
            for (int j = 0; j < Z; ++j) {
                const double sv = wsel[j];
                if (sv > 0.0) s += V[a * n + j] * V[b * n + j] / sv;
            }
            Minv[i] = (float)((double)gscale * (double)lmbda / (double)bsize * s);
        }
    }
}

// doubles of state `vame_nuclear_f32` needs for a latent width: Z'^2 (optional warm start) up to Z = 64, 3 Z'^2 (required) above
extern "C" int64_t vame_nuclear_state_doubles(int Z) {
    const int64_t n = Z + (Z & 1);
    return Z <= NUC_MAXZ ? n * n : 3 * n * n;
}

extern "C" int vame_nuclear_f32(const float* G, int Z, int kloss, int nrows, float lmbda, float bsize, float gscale,
                                float* loss_out, float* Minv, double* vstate, void* stream) {
    VAME_CHECK_ARG(G && loss_out, VAME_E_BADARG, "nuclear: null pointer");
    VAME_CHECK_ARG(Z >= 1 && Z <= NUCB_MAXZ && kloss >= 1 && nrows >= 1 && bsize > 0, VAME_E_SHAPE, "nuclear: Z=%d must be in 1..%d",
                   Z, NUCB_MAXZ);
    if (Z <= NUC_MAXZ) {
        hipLaunchKernelGGL(nuclear_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, G, Z, kloss, nrows, lmbda, bsize, gscale,
                           loss_out, Minv, vstate);
    } else {
        VAME_CHECK_ARG(vstate, VAME_E_BADARG, "nuclear: Z=%d > %d needs the state buffer (vame_nuclear_state_doubles)", Z, NUC_MAXZ);
        hipLaunchKernelGGL(nuclear_big_kernel, dim3(1), dim3(NUCB_THREADS), 0, (hipStream_t)stream, G, Z, kloss, nrows, lmbda, bsize,
                           gscale, loss_out, Minv, vstate);
    }
    VAME_LAUNCH_CHECK("nuclear");
    return VAME_OK;
}
