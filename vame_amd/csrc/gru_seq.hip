// GRU sequence kernels (forward + BPTT) for gfx950.
//
// Mapping (MI355X-first, not a cuDNN-style per-step GEMM chain): a workgroup owns 32 batch rows
// of ONE (layer,direction) stream for all T steps -- batch rows are independent, so there is no
// grid-wide synchronisation.  Wave w owns hidden columns [32w, 32w+32): its three gate tiles
// (r,z,n) are 32x32 fp32 MFMA accumulators whose C-layout puts r,z,n and h of the same
// (row, column) in the same lane, so the sigmoid/tanh/blend epilogue is lane-local.  h_t lives in
// LDS (double-buffered A operand) and in registers (blend operand); W_hh is streamed from L2
// every step in a pre-packed B-fragment order (one coalesced 1 KiB dwordx4 load per 4 MFMAs).
// Streams (fwd/bwd directions, decoder + future decoder) are spread over XCDs so each XCD's L2
// holds one stream's weights.
//
// Reference semantics: torch.nn.GRU as instantiated at vame/model/rnn_model.py:34-35,91-92,125-126.
#include "vame_common.h"
#include "gru_desc.h"

#ifdef VAME_PROBE   // tuning build (make probe): per-workgroup begin/end stamps, s_memtime (shader clock) vs s_memrealtime (100 MHz)
__device__ long long* g_gru_probe;
extern "C" int vame_probe_set_gru(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gru_probe), &p, sizeof(p)); }
#define GRU_PROBE_BEGIN() const long long pk_t0 = (long long)__builtin_amdgcn_s_memtime(), pk_r0 = (long long)__builtin_amdgcn_s_memrealtime()
#define GRU_PROBE_END()                                                                                      \
    if (threadIdx.x == 0 && g_gru_probe) {                                                                    \
        long long* o_ = g_gru_probe + (long long)blockIdx.x * 4;                                              \
        o_[0] = (long long)__builtin_amdgcn_s_memtime() - pk_t0; o_[2] = pk_r0;                               \
        o_[3] = (long long)__builtin_amdgcn_s_memrealtime(); o_[1] = o_[3] - pk_r0;                           \
    }
// phase timers of one wave (wave 0 of each workgroup reports): 8 accumulators after the 4 stamp slots of all workgroups
#define GRU_PHASE_DECL() long long pp_[24] = {0}, pa_ = (long long)__builtin_amdgcn_s_memtime()
#define GRU_PHASE(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); pp_[i] += t_ - pa_; pa_ = t_; } while (0)
#define GRU_PHASE_DYN(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); const long long d_ = t_ - pa_; pa_ = t_; \
        _Pragma("unroll") for (int k_ = 8; k_ < 24; ++k_) if (k_ == (i)) pp_[k_] += d_; } while (0)
#define GRU_PHASE_END()                                                                                      \
    if (threadIdx.x == 0 && g_gru_probe) {                                                                    \
        long long* o_ = g_gru_probe + (1 << 16) + (long long)blockIdx.x * 24;                                 \
        for (int i_ = 0; i_ < 24; ++i_) o_[i_] = pp_[i_];                                                     \
    }                                                                                                         \
    if ((threadIdx.x & 63) == 0 && g_gru_probe) {      /* every wave: its first 8 phase sums, after wave 0's region */ \
        long long* o_ = g_gru_probe + (1 << 16) + (1 << 14) * 24 + ((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8; \
        for (int i_ = 0; i_ < 7; ++i_) o_[i_] = pp_[i_];                                                      \
        long long g_ = 0; for (int i_ = 8; i_ < 24; ++i_) g_ += pp_[i_];                                      \
        o_[7] = g_;                                     /* the MFMA loop's chunk groups together */           \
    }
#else
#define GRU_PROBE_BEGIN()
#define GRU_PROBE_END()
#define GRU_PHASE_DECL()
#define GRU_PHASE(i)
#define GRU_PHASE_DYN(i)
#define GRU_PHASE_END()
#endif

// blockIdx -> (stream, tile).  Workgroup b is dispatched to XCD b%8 (observed, speed only).  Streams are dealt to
// XCD parity classes so that one XCD's 4 MiB L2 keeps at most two streams' W_hh, and -- for 4 streams ordered
// (long, long, short, short), i.e. decoder f/b + future-decoder f/b -- every XCD gets the same amount of work:
// XCD x serves stream x&1 first (its tiles x>>1, x>>1 + 4, ...) and then stream 2 + (x&1).
__device__ __forceinline__ bool map_block(int nstreams, int ntiles, int& s, int& tile) {
    const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
    if (nstreams == 1) { s = 0; tile = q * 8 + xcd; }
    else if (nstreams == 2) { s = xcd & 1; tile = q * 4 + (xcd >> 1); }
    else if (nstreams == 4) {
        const int per = (ntiles + 3) >> 2;
        if (q < per) { s = xcd & 1; tile = q * 4 + (xcd >> 1); }
        else { s = 2 + (xcd & 1); tile = (q - per) * 4 + (xcd >> 1); }
    } else { s = bid % nstreams; tile = bid / nstreams; }
    return tile < ntiles;
}
static int grid_blocks(int nstreams, int ntiles) {
    if (nstreams == 1) return (int)cdiv64(ntiles, 8) * 8;
    if (nstreams == 2) return (int)cdiv64(ntiles, 4) * 8;
    if (nstreams == 4) return 2 * (int)cdiv64(ntiles, 4) * 8;
    return nstreams * ntiles;
}

// ------------------------------------------------------------------------------------------- pack
// wp_fwd[(((w*(H/8)+c)*3+g)*64+l)*4+e] = W_hh[(g*H+32w+(l&31))*H + 8c+4(l>>5)+e]
// wp_bwd[((w*(3H/8)+c)*64+l)*4+e]       = W_hh[(8c+4(l>>5)+e)*H + 32w+(l&31)]
__device__ __forceinline__ void gru_pack_body(const float* __restrict__ W, const float* __restrict__ b_ih,
                                              const float* __restrict__ b_hh, int H, float* __restrict__ wpf,
                                              float* __restrict__ wpb, float* __restrict__ bias_gi, float* __restrict__ bhn) {
    const int64_t n = (int64_t)3 * H * H;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        {
            const int e = i & 3, l = (i >> 2) & 63;
            int64_t r = i >> 8;
            const int g = r % 3; r /= 3;
            const int c = r % (H / 8), w = r / (H / 8);
            wpf[i] = W[(int64_t)(g * H + 32 * w + (l & 31)) * H + 8 * c + 4 * (l >> 5) + e];
        }
        {
            const int e = i & 3, l = (i >> 2) & 63;
            const int64_t r = i >> 8;
            const int c = r % (3 * H / 8), w = r / (3 * H / 8);
            wpb[i] = W[(int64_t)(8 * c + 4 * (l >> 5) + e) * H + 32 * w + (l & 31)];
        }
        if (i < 3 * H) bias_gi[i] = b_ih[i] + (i < 2 * H ? b_hh[i] : 0.0f);
        if (i < H) bhn[i] = b_hh[2 * H + i];
    }
}
__global__ __launch_bounds__(256) void gru_pack_kernel(const float* __restrict__ W, const float* __restrict__ b_ih,
                                                       const float* __restrict__ b_hh, int H, float* __restrict__ wpf,
                                                       float* __restrict__ wpb, float* __restrict__ bias_gi,
                                                       float* __restrict__ bhn) {
    gru_pack_body(W, b_ih, b_hh, H, wpf, wpb, bias_gi, bhn);
}

extern "C" int vame_gru_pack_f32(const float* W_hh, const float* b_ih, const float* b_hh, int H, float* wp_fwd,
                                 float* wp_bwd, float* bias_gi, float* b_hn, void* stream) {
    VAME_CHECK_ARG(H >= 32 && H % 32 == 0, VAME_E_SHAPE, "gru_pack: H=%d must be a multiple of 32", H);
    VAME_CHECK_ARG(W_hh && b_ih && b_hh && wp_fwd && wp_bwd && bias_gi && b_hn, VAME_E_BADARG, "gru_pack: null pointer");
    const int64_t n = (int64_t)3 * H * H;
    const int blocks = (int)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024);
    hipLaunchKernelGGL(gru_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W_hh, b_ih, b_hh, H, wp_fwd,
                       wp_bwd, bias_gi, b_hn);
    VAME_LAUNCH_CHECK("gru_pack");
    return VAME_OK;
}

// wpx[(((w*4+c)*3+g)*64+l)*4+e] = W_ih[(g*H+32w+(l&31))*F + k], k = 8c+4(l>>5)+e (< F, else 0): the K = 32 (zero padded)
// input projection of a layer whose input has F <= 32 features, in the same B-fragment order as wp_fwd
__device__ __forceinline__ void gru_pack_x_body(const float* __restrict__ W, int F, int H, float* __restrict__ wpx) {
    const int64_t n = (int64_t)3 * H * 32;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = i & 3, l = (i >> 2) & 63;
        int64_t r = i >> 8;
        const int g = r % 3; r /= 3;
        const int c = r % 4, w = r / 4;
        const int k = 8 * c + 4 * (l >> 5) + e;
        wpx[i] = k < F ? W[(int64_t)(g * H + 32 * w + (l & 31)) * F + k] : 0.0f;
    }
}
__global__ __launch_bounds__(256) void gru_pack_x_kernel(const float* __restrict__ W, int F, int H, float* __restrict__ wpx) {
    gru_pack_x_body(W, F, H, wpx);
}

// All (layer, direction) packs of a model in ONE launch (the weights change every optimizer step: eight 3-us kernels and their
// dispatch gaps otherwise open every train step).  blockIdx.y = item.
struct GruPackItem { const float* W_hh; const float* b_ih; const float* b_hh; int H; float* wpf; float* wpb; float* bias_gi; float* bhn;
                     const float* W_ih; int F; float* wpx; };
struct GruPackBatch { GruPackItem it[VAME_GRU_PACK_MAX]; };
__global__ __launch_bounds__(256) void gru_pack_batch_kernel(GruPackBatch P) {
    const GruPackItem& I = P.it[blockIdx.y];
    gru_pack_body(I.W_hh, I.b_ih, I.b_hh, I.H, I.wpf, I.wpb, I.bias_gi, I.bhn);
    if (I.W_ih) gru_pack_x_body(I.W_ih, I.F, I.H, I.wpx);
}

extern "C" int vame_gru_pack_batch_f32(const int64_t* items, int n, void* stream) {
    VAME_CHECK_ARG(items && n >= 1 && n <= VAME_GRU_PACK_MAX, VAME_E_BADARG, "gru_pack_batch: n=%d not in 1..%d", n, VAME_GRU_PACK_MAX);
    GruPackBatch P;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t* r = items + (int64_t)i * VAME_GRU_PACK_FIELDS;
        GruPackItem& I = P.it[i];
        I.W_hh = (const float*)r[GP_W_HH]; I.b_ih = (const float*)r[GP_B_IH]; I.b_hh = (const float*)r[GP_B_HH]; I.H = (int)r[GP_H];
        I.wpf = (float*)r[GP_WP_FWD]; I.wpb = (float*)r[GP_WP_BWD]; I.bias_gi = (float*)r[GP_BIAS_GI]; I.bhn = (float*)r[GP_B_HN];
        I.W_ih = (const float*)r[GP_W_IH]; I.F = (int)r[GP_F]; I.wpx = (float*)r[GP_WPX];
        VAME_CHECK_ARG(I.H >= 32 && I.H % 32 == 0, VAME_E_SHAPE, "gru_pack_batch: item %d: H=%d must be a multiple of 32", i, I.H);
        VAME_CHECK_ARG(I.W_hh && I.b_ih && I.b_hh && I.wpf && I.wpb && I.bias_gi && I.bhn, VAME_E_BADARG, "gru_pack_batch: item %d: null pointer", i);
        VAME_CHECK_ARG(!I.W_ih || (I.wpx && I.F >= 1 && I.F <= 32), VAME_E_SHAPE, "gru_pack_batch: item %d: fused input needs wpx and F <= 32 (F=%d)", i, I.F);
        const int64_t e = (int64_t)3 * I.H * I.H;
        most = e > most ? e : most;
    }
    const int blocks = (int)(cdiv64(most, 256) < 256 ? cdiv64(most, 256) : 256);
    hipLaunchKernelGGL(gru_pack_batch_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, P);
    VAME_LAUNCH_CHECK("gru_pack_batch");
    return VAME_OK;
}

extern "C" int vame_gru_pack_x_f32(const float* W_ih, int F, int H, float* wpx, void* stream) {
    VAME_CHECK_ARG(H >= 32 && H % 32 == 0 && F >= 1 && F <= 32, VAME_E_SHAPE, "gru_pack_x: H=%d F=%d (need F <= 32)", H, F);
    VAME_CHECK_ARG(W_ih && wpx, VAME_E_BADARG, "gru_pack_x: null pointer");
    hipLaunchKernelGGL(gru_pack_x_kernel, dim3((unsigned)cdiv64(3 * H * 32, 256)), dim3(256), 0, (hipStream_t)stream, W_ih, F, H, wpx);
    VAME_LAUNCH_CHECK("gru_pack_x");
    return VAME_OK;
}

extern "C" int64_t vame_gru_stash_floats(int B, int T, int H) {
    return cdiv64(B, 32) * 32 * (int64_t)T * 5 * H;
}

// ------------------------------------------------------------------------------------------- forward
// Addressing: in the 32x32 accumulator layout register r of lane l holds row CR(r) + 4*(l>>5), column l&31.
// Every row-major operand is therefore addressed as  uniform_row_pointer(r) [ lane_offset ]  with
// lane_offset = 4*(l>>5)*row_stride + (l&31): one VGPR per array, row pointers stay in SGPRs.
// Backward stash, written in the accumulator-fragment order it is read back in (opaque to the host):
//   float4 index ((((tile*T + t)*NW + w)*5 + k)*4 + rq)*64 + lane,  k = cA, cB, u, r, gh_n  with
//   cA = (1-u)(1-n^2) (d a_n / d h'),  cB = (h_prev - n) u (1-u) (d a_z / d h'):  every BPTT gate gradient is
//   d * {cA, cB, u} or a product with r / gh_n, so the backward kernel needs neither n nor h_prev.
// ABL (ablation mask, 0 in production; tools/microbench.py VAME_ABL_FWD): 1 no stash stores, 2 no gi loads, 256 gi loads mid-loop, 512 gi as
// twelve wide loads (timing only),
// 4 no y stores, 8 no gate transcendental math, 16 W fragments not re-streamed, 32 no per-step barrier
// XIN: the input projection x_t W_ih^T (F <= 32 features, zero padded to K = 32) is computed in-kernel as four extra
// MFMA chunks per step from an LDS-staged (32 x F) tile of x_t; no gi tensor exists (encoder layer 0).
template <int H, int ABL = 0, bool XIN = false>
__global__ __launch_bounds__(H / 32 * 64) void gru_seq_fwd_kernel(GruFwdParams P) {
    constexpr int NW = H / 32, LDH = H + 4, KC = H / 8, LDX = 36;
    __shared__ float hs[2][32 * LDH];
    __shared__ float xs[XIN ? 2 : 1][XIN ? 32 * LDX : 4];
    int sidx, tile;
    if (!map_block(P.nstreams, P.ntiles, sidx, tile)) return;
    GRU_PROBE_BEGIN();
    const GruFwdStream& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    const int row0 = tile * 32, col0 = 32 * w;
    const int nvalid = B - row0;                        // rows of this tile inside the batch (>= 32: full tile)
    const bool full = nvalid >= 32;
    const int lrow = 4 * hh;                            // lane part of the fragment row
    const int lo_gi = lrow * (int)S.gi_row + li;
    const float* gi_base = S.gi + (int64_t)row0 * S.gi_row + col0;
    float* y_tile = S.y ? S.y + (int64_t)row0 * S.y_row : nullptr;
    // h_t leaves through LDS: after the step's barrier the whole 32 x H tile is copied row-major with 16-byte
    // stores (4 per thread) instead of 16 dword stores per lane from the accumulator layout
    const int crow = tid / (H / 4), cc4 = tid % (H / 4);          // copy pass: 8 rows x (H/4) float4 per pass, 4 passes
    auto store_h = [&](const float* hbuf, int t) {
        float* yt = y_tile + (int64_t)t * S.y_t + (int64_t)crow * S.y_row + 4 * cc4;
        const float* src = hbuf + crow * LDH + 4 * cc4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (full || crow + 8 * i < nvalid)
                *reinterpret_cast<float4*>(yt + (int64_t)(8 * i) * S.y_row) = *reinterpret_cast<const float4*>(src + 8 * i * LDH);
    };
    f32x16 hprev;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = CR(r) + lrow, grow = row0 + row;
        float v = 0.0f;
        if (S.h0 && grow < B) v = S.h0[(int64_t)grow * S.h0_row + col0 + li];
        hprev[r] = v;
        hs[0][row * LDH + col0 + li] = v;
    }
    // fused input: thread -> (row, float4 column) of the (32 x F) x_t tile; columns F..31 of the LDS tile stay zero
    constexpr int XI = XIN ? (32 * 8 + NW * 64 - 1) / (NW * 64) : 1;      // x-tile items per thread (1 for H >= 128)
    const int nq = XIN ? (int)S.xf / 4 : 1;
    float4 xv[XI];
    int xgo[XI], xlo[XI];                     // per item: global element offset (-1 = none) and LDS offset, resolved once
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int idx = tid + i * NW * 64, xr = idx / nq, xq = idx % nq;
        xlo[i] = idx < 32 * nq ? xr * LDX + 4 * xq : -1;
        xgo[i] = (XIN && idx < 32 * nq && xr < nvalid) ? (row0 + xr) * (int)S.gi_row + 4 * xq : -1;
    }
    auto load_x = [&](int t) {
        const float* xt = S.gi + (int64_t)t * S.gi_t;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            xv[i] = xgo[i] >= 0 ? *reinterpret_cast<const float4*>(xt + xgo[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_x = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (xlo[i] >= 0) *reinterpret_cast<float4*>(&buf[xlo[i]]) = xv[i];
    };
    if (XIN) {
        for (int i = tid; i < 2 * 32 * LDX; i += NW * 64) (&xs[0][0])[i] = 0.f;
        load_x(S.reverse ? T - 1 : 0);
    }
    __syncthreads();
    if (XIN) store_x(xs[0]);
    if (y_tile && S.pad) store_h(hs[0], S.reverse ? T : -1);
    const float bhn = S.bhn[col0 + li];
    const float bgr = XIN ? S.bgi[col0 + li] : 0.f, bgu = XIN ? S.bgi[H + col0 + li] : 0.f, bgn = XIN ? S.bgi[2 * H + col0 + li] : 0.f;
    // uniform ring bases (SGPR pairs) + one per-lane byte offset (RING_LOAD_U: no vector-ALU address arithmetic in the K loop)
    const float4* __restrict__ wpx = XIN ? reinterpret_cast<const float4*>(S.wpx) + (int64_t)w * 4 * 3 * 64 : nullptr;
    const float4* __restrict__ wp = reinterpret_cast<const float4*>(S.wp) + (int64_t)w * KC * 3 * 64;
    const unsigned lane16 = (unsigned)lane * 16u;
    float4* stash = S.stash ? reinterpret_cast<float4*>(S.stash) : nullptr;
    int cur = 0;
    // gi of the first step; later steps are prefetched during the previous step's MFMA loop
    f32x16 gr, gu, gn;
    auto load_gi = [&](int t) {
        const float* gt = gi_base + (int64_t)t * S.gi_t;
        if (ABL & 512) {        // timing experiment only (wrong values): the same 12 KB as twelve contiguous 16-byte loads per lane
            const float4* gf = reinterpret_cast<const float4*>(S.gi + (int64_t)(row0 + (t % 16) * 2) * S.gi_row + w * 3072) + lane;   // 12 KB of its own per (tile, step, wave)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = gf[(0 * 4 + q) * 64], b = gf[(1 * 4 + q) * 64], c = gf[(2 * 4 + q) * 64];
                gr[4 * q] = a.x; gr[4 * q + 1] = a.y; gr[4 * q + 2] = a.z; gr[4 * q + 3] = a.w;
                gu[4 * q] = b.x; gu[4 * q + 1] = b.y; gu[4 * q + 2] = b.z; gu[4 * q + 3] = b.w;
                gn[4 * q] = c.x; gn[4 * q + 1] = c.y; gn[4 * q + 2] = c.z; gn[4 * q + 3] = c.w;
            }
        } else
        if (!(ABL & 2) && full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* g = gt + (int64_t)CR(r) * S.gi_row;
                gr[r] = g[lo_gi]; gu[r] = g[lo_gi + H]; gn[r] = g[lo_gi + 2 * H];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* g = gt + (int64_t)CR(r) * S.gi_row;
                if (!(ABL & 2) && CR(r) + lrow < nvalid) { gr[r] = g[lo_gi]; gu[r] = g[lo_gi + H]; gn[r] = g[lo_gi + 2 * H]; }
                else { gr[r] = 0.f; gu[r] = 0.f; gn[r] = 0.f; }
            }
        }
    };
    if (!XIN) load_gi(S.reverse ? T - 1 : 0);
    constexpr int PD = 4;                       // KC % PD == 0 for every supported H; the fused input has exactly PD chunks
    f32x4 wq[PD][3];
    {
        const float4* w0 = XIN ? wpx : wp;
#pragma unroll
        for (int c = 0; c < PD; ++c) { RING_LOAD_U(wq[c][0], w0 + (c) * 3 * 64, lane16, 0); RING_LOAD_U(wq[c][1], w0 + (c) * 3 * 64, lane16, 1024); RING_LOAD_U(wq[c][2], w0 + (c) * 3 * 64, lane16, 2048); }
    }
    if (XIN) __syncthreads();
    GRU_PHASE_DECL();
    for (int step = 0; step < T; ++step) {
        const int t = S.reverse ? T - 1 - step : step;
        const bool skip_first = step == 0 && S.h0 == nullptr && (XIN || S.gi_t == 0) && !(ABL & 63);   // (per-step gi streams keep
                                                                                                  // the loop: it carries their gi prefetch)
        f32x16 ar, au, ani, anh;
        GRU_PHASE(0);
        if (XIN) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { ar[r] = bgr; au[r] = bgu; ani[r] = bgn; anh[r] = bhn; }
            if (step + 1 < T) load_x(S.reverse ? t - 1 : t + 1);
            // input projection: K = 32 from the LDS x tile; the ring is refilled with the first W_hh chunks
            const float* xrow = &xs[cur][li * LDX + 4 * hh];
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * j);
                RING_WAIT3(3 * (PD - 1), wq[j][0], wq[j][1], wq[j][2]);
                const f32x4 b0 = wq[j][0], b1 = wq[j][1], b2 = wq[j][2];
                ar = MFMA_32x32x2(a.x, b0[0], ar); au = MFMA_32x32x2(a.x, b1[0], au); ani = MFMA_32x32x2(a.x, b2[0], ani);
                ar = MFMA_32x32x2(a.y, b0[1], ar); au = MFMA_32x32x2(a.y, b1[1], au); ani = MFMA_32x32x2(a.y, b2[1], ani);
                ar = MFMA_32x32x2(a.z, b0[2], ar); au = MFMA_32x32x2(a.z, b1[2], au); ani = MFMA_32x32x2(a.z, b2[2], ani);
                ar = MFMA_32x32x2(a.w, b0[3], ar); au = MFMA_32x32x2(a.w, b1[3], au); ani = MFMA_32x32x2(a.w, b2[3], ani);
                RING_FENCE();                  // refill only after the slot's last use: see the note at the main loop
                const float4* nsrc = skip_first ? wpx : wp;      // no recurrent loop in a zero-state first step: keep the input chunks
                RING_LOAD_U(wq[j][0], nsrc + (j) * 3 * 64, lane16, 0); RING_LOAD_U(wq[j][1], nsrc + (j) * 3 * 64, lane16, 1024); RING_LOAD_U(wq[j][2], nsrc + (j) * 3 * 64, lane16, 2048);
            }
        } else {
            ar = gr; au = gu; ani = gn;
#pragma unroll
            for (int r = 0; r < 16; ++r) anh[r] = bhn;
        }
        const float* hrow = &hs[cur][li * LDH + 4 * hh];
        if (skip_first) {
            // h_{-1} = 0 (no initial state given): h W_hh^T contributes nothing to the first step -- skip its 3H x H MACs; the
            // ring keeps the chunks it holds for the next step
        } else
        // software pipeline, distance PD chunks: W_hh fragments (L2) are requested PD x 12 MFMAs ahead of use; the
        // first PD chunks of a step were requested before the previous step's epilogue (they do not depend on h)
#pragma unroll 1
        for (int c0 = 0; c0 < KC; c0 += PD) {
        if ((ABL & 256) && !XIN && c0 == KC / 2 / PD * PD && step + 1 < T && S.gi_t != 0) load_gi(S.reverse ? t - 1 : t + 1);   // (ablation: the round-1 place)
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int c = c0 + j;
            const float4 a = *reinterpret_cast<const float4*>(hrow + 8 * c);
            RING_WAIT3(3 * (PD - 1), wq[j][0], wq[j][1], wq[j][2]);
            const f32x4 b0 = wq[j][0], b1 = wq[j][1], b2 = wq[j][2];
            ar = MFMA_32x32x2(a.x, b0[0], ar); au = MFMA_32x32x2(a.x, b1[0], au); anh = MFMA_32x32x2(a.x, b2[0], anh);
            ar = MFMA_32x32x2(a.y, b0[1], ar); au = MFMA_32x32x2(a.y, b1[1], au); anh = MFMA_32x32x2(a.y, b2[1], anh);
            ar = MFMA_32x32x2(a.z, b0[2], ar); au = MFMA_32x32x2(a.z, b1[2], au); anh = MFMA_32x32x2(a.z, b2[2], anh);
            ar = MFMA_32x32x2(a.w, b0[3], ar); au = MFMA_32x32x2(a.w, b1[3], au); anh = MFMA_32x32x2(a.w, b2[3], anh);
            RING_FENCE();
            {
                // ring refill for chunk c + PD, issued AFTER the slot's last use so the load lands in the same registers
                // (a refill placed before the MFMAs makes hipcc load into fresh registers and rotate the ring with v_mov
                // at every back edge).  The last group wraps into the next step's first chunks (the input-projection
                // chunks when XIN): pointer select, no branch
                const bool wrap = c0 + PD == KC;
                const float4* src = (XIN && wrap) ? wpx : wp;
                const int cn = (ABL & 16) ? 0 : (wrap ? j : c + PD);
                RING_LOAD_U(wq[j][0], src + (cn) * 3 * 64, lane16, 0); RING_LOAD_U(wq[j][1], src + (cn) * 3 * 64, lane16, 1024); RING_LOAD_U(wq[j][2], src + (cn) * 3 * 64, lane16, 2048);
            }
        }
        GRU_PHASE_DYN(8 + c0 / PD);
        }
        // gi of the next step: requested AFTER the loop's last weight-fragment wait.  The ring waits are vmcnt(9) -- "at most nine
        // younger operations outstanding" -- so 48 gi loads issued inside the loop turn the next such wait into a wait for the gi
        // loads themselves (probe: 6500 of a step's 80000 cycles); from here their latency runs under the epilogue and the barrier.
        if (!(ABL & 256) && !XIN && !skip_first && step + 1 < T && S.gi_t != 0) load_gi(S.reverse ? t - 1 : t + 1);
        GRU_PHASE(1);                 // input projection + recurrent MFMA loop
        float* hnext = &hs[cur ^ 1][lrow * LDH + col0 + li];
        f32x16 ust;
        float4* sp = stash ? stash + ((((int64_t)tile * T + t) * NW + w) * 20) * 64 + lane : nullptr;
        // four groups of 4 fragment rows: gate math of a group, then its 5 stash stores, so that the stores' trip through the
        // vector-memory path overlaps the next group's VALU work instead of following all of it
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * q + j;
                const float rr = (ABL & 8) ? ar[r] * 0.01f : fast_sigmoid(ar[r]);
                const float uu = (ABL & 8) ? au[r] * 0.01f : fast_sigmoid(au[r]);
                const float nn = (ABL & 8) ? (ani[r] + rr * anh[r]) * 0.01f : fast_tanh(ani[r] + rr * anh[r]);
                const float hp = hprev[r];
                const float hv = nn + uu * (hp - nn);
                const float omu = 1.0f - uu;
                ani[r] = omu * (1.0f - nn * nn);          // cA
                au[r] = (hp - nn) * uu * omu;             // cB
                ar[r] = rr;
                ust[r] = uu;
                hprev[r] = hv;
                hnext[CR(r) * LDH] = hv;
            }
            if (!(ABL & 1) && stash) {
                const float4 v0 = make_float4(ani[4 * q], ani[4 * q + 1], ani[4 * q + 2], ani[4 * q + 3]);
                const float4 v1 = make_float4(au[4 * q], au[4 * q + 1], au[4 * q + 2], au[4 * q + 3]);
                const float4 v2 = make_float4(ust[4 * q], ust[4 * q + 1], ust[4 * q + 2], ust[4 * q + 3]);
                const float4 v3 = make_float4(ar[4 * q], ar[4 * q + 1], ar[4 * q + 2], ar[4 * q + 3]);
                const float4 v4 = make_float4(anh[4 * q], anh[4 * q + 1], anh[4 * q + 2], anh[4 * q + 3]);
                sp[(0 * 4 + q) * 64] = v0; sp[(1 * 4 + q) * 64] = v1; sp[(2 * 4 + q) * 64] = v2; sp[(3 * 4 + q) * 64] = v3; sp[(4 * 4 + q) * 64] = v4;
            }
            SCHED_FENCE();
        }
        GRU_PHASE(2);                 // gate math + h -> LDS + stash stores
        if (XIN) store_x(xs[cur ^ 1]);
        GRU_PHASE(3);                 // stash stores (issue)
        if (!(ABL & 32)) __syncthreads();
        GRU_PHASE(4);                 // barrier
        cur ^= 1;
        if (!(ABL & 4) && y_tile) store_h(hs[cur], t);
        GRU_PHASE(5);                 // y stores through LDS
    }
    GRU_PHASE_END();
    if (S.hn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int grow = row0 + CR(r) + lrow;
            if (grow < B) S.hn[(int64_t)grow * S.hn_row + col0 + li] = hprev[r];
        }
    }
    GRU_PROBE_END();
}

// ------------------------------------------------------------------------------------------- forward, two wave groups half a step apart (round 4)
// Same tile, same LDS images, same stash / Y formats, same arithmetic in the same order as gru_seq_fwd_kernel (bit-identical outputs) -- but
// the two waves of every SIMD no longer do the same thing at the same time.  The lock-step kernel serialises, per step and per SIMD,
// [contraction of both waves: 49 k cycles] -> [gate math of both waves: ~7 k cycles of VALU + transcendentals with the matrix pipe idle]
// -> [barrier, with the stash stores' and the next gi tile's round trips in it]; nothing of a tile's step t+1 can start before ALL of h_t
// exists.  But h_t is only needed COLUMN BLOCK BY COLUMN BLOCK by the K loop of step t+1.  So the waves form two groups -- X = the first
// half of the waves (hidden columns [0, H/2)), Y = the second half ([H/2, H)); wave w and wave w + NW/2 share a SIMD -- and every wave
// runs its step in two parts with a workgroup barrier after each:
//     part 0:  accumulators <- gi;  K loop over k in [0, H/2)         (needs h_{t-1} of X's columns)
//     part 1:  K loop over k in [H/2, H);  gate math, h_t -> LDS, stash (needs h_{t-1} of Y's columns)
// with Y HALF A STEP BEHIND X: in interval J(2s) X runs part 0 of step s while Y runs part 1 of step s-1, in J(2s+1) X runs part 1 of s
// while Y runs part 0 of s.  Every dependency is one barrier old: X.part0(s) reads the columns X wrote in J(2s-1), X.part1(s) those Y
// wrote in J(2s), Y.part0(s) reads X's of J(2s-1), Y.part1(s) Y's own of J(2s); a half is overwritten (other LDS image) three intervals
// after its last reader.  Per accumulator the k order is still ascending, hence the identical bits.  What it buys: on every SIMD one wave
// is in part 1 while the other is in part 0, so gate math, the wait for the gi tile and the drain of the stash stores of one wave run
// beside the other wave's MFMAs; the wave in part 1 raises its priority so that its MFMAs go first and its VALU tail is covered.
// 2 T + 1 intervals for T steps (half an interval of fill and of drain).  Barriers order LDS only (LDS_BARRIER): global stores and
// the weight ring stay in flight across them.
// prio (P.pace_cp; tuning): 0 = no priority change, 1 (default) = s_setprio 1 during part 1.
template <int H, bool XIN>
__global__ __launch_bounds__(H / 32 * 64) void gru_skew_fwd_kernel(GruFwdParams P) {
    constexpr int NW = H / 32, GW = NW / 2, GT = GW * 64, LDH = H + 4, KC = H / 8, KH = KC / 2, LDX = 36, PD = 4;
    static_assert(NW % 2 == 0 && KH % PD == 0, "skewed forward: H a multiple of 64");
    __shared__ float hs[2][32 * LDH];                          // h_{s-1} lives in hs[s & 1]
    __shared__ float xs[XIN ? 2 : 1][XIN ? 32 * LDX : 4];      // x_s lives in xs[s & 1]
    int sidx, tile;
    if (!map_block(P.nstreams, P.ntiles, sidx, tile)) return;
    GRU_PROBE_BEGIN();
    const GruFwdStream& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    const int grp = UNIFORM(w >= GW ? 1 : 0);
    const int gtid = tid - grp * GT;                           // thread index inside the group
    const int row0 = tile * 32, col0 = 32 * w, lrow = 4 * hh;
    const int nvalid = B - row0;
    const bool full = nvalid >= 32;
    const int prio = P.pace_cp < 0 ? 1 : P.pace_cp;
    const int delay = P.pace_ld < 0 ? 0 : P.pace_ld;          // part 0 starts `delay` x ~256 cycles late (see the step loop)
    const int lo_gi = lrow * (int)S.gi_row + li;
    const float* gi_base = S.gi + (int64_t)row0 * S.gi_row + col0;
    float* y_tile = S.y ? S.y + (int64_t)row0 * S.y_row : nullptr;
    // h_t leaves through LDS as 16-byte row stores; a group copies the half it wrote: GT = H threads cover 8 rows x (H/8) float4 per pass
    const int crow = gtid / (H / 8), cc = grp * (H / 2) + 4 * (gtid % (H / 8));
    auto store_h = [&](const float* hbuf, int t) {
        float* yt = y_tile + (int64_t)t * S.y_t + (int64_t)crow * S.y_row + cc;
        const float* src = hbuf + crow * LDH + cc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (full || crow + 8 * i < nvalid)
                *reinterpret_cast<float4*>(yt + (int64_t)(8 * i) * S.y_row) = *reinterpret_cast<const float4*>(src + 8 * i * LDH);
    };
    auto time_of = [&](int s) { return S.reverse ? T - 1 - s : s; };
    f32x16 hprev;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = CR(r) + lrow, grow = row0 + row;
        float v = 0.0f;
        if (S.h0 && grow < B) v = S.h0[(int64_t)grow * S.h0_row + col0 + li];
        hprev[r] = v;
        hs[0][row * LDH + col0 + li] = v;
    }
    // fused input (XIN): the (32 x F) tile of x_s is staged by group X's threads alone -- fetched during X's part 0 of step s-1, written
    // to xs[s & 1] at the end of its part 1 (read by X one barrier later, by Y two; its previous content x_{s-2} was last read by Y two
    // intervals earlier); columns F..31 stay zero
    constexpr int XI = XIN ? (32 * 8 + GT - 1) / GT : 1;
    const int nq = XIN ? (int)S.xf / 4 : 1;
    float4 xv[XI];
    int xgo[XI], xlo[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int idx = gtid + i * GT, xr = idx / nq, xq = idx % nq;
        const bool mine = XIN && grp == 0 && idx < 32 * nq;
        xlo[i] = mine ? xr * LDX + 4 * xq : -1;
        xgo[i] = (mine && xr < nvalid) ? (row0 + xr) * (int)S.gi_row + 4 * xq : -1;
    }
    auto load_x = [&](int t) {
        const float* xt = S.gi + (int64_t)t * S.gi_t;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            xv[i] = xgo[i] >= 0 ? *reinterpret_cast<const float4*>(xt + xgo[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_x = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (xlo[i] >= 0) *reinterpret_cast<float4*>(&buf[xlo[i]]) = xv[i];
    };
    if (XIN) {
        for (int i = tid; i < 2 * 32 * LDX; i += NW * 64) (&xs[0][0])[i] = 0.f;
        load_x(time_of(0));
    }
    __syncthreads();
    if (XIN) store_x(xs[0]);
    if (y_tile && S.pad) store_h(hs[0], S.reverse ? T : -1);
    const float bhn = S.bhn[col0 + li];
    const float bgr = XIN ? S.bgi[col0 + li] : 0.f, bgu = XIN ? S.bgi[H + col0 + li] : 0.f, bgn = XIN ? S.bgi[2 * H + col0 + li] : 0.f;
    // uniform ring bases (SGPR pairs) + one per-lane byte offset (RING_LOAD_U: no vector-ALU address arithmetic in the K loop)
    const float4* __restrict__ wpx = XIN ? reinterpret_cast<const float4*>(S.wpx) + (int64_t)w * 4 * 3 * 64 : nullptr;
    const float4* __restrict__ wp = reinterpret_cast<const float4*>(S.wp) + (int64_t)w * KC * 3 * 64;
    const unsigned lane16 = (unsigned)lane * 16u;
    float4* stash = S.stash ? reinterpret_cast<float4*>(S.stash) : nullptr;
    f32x16 gr, gu, gn;                       // gi of the wave's next step (per-step gi streams: fetched behind part 1's K loop)
    auto load_gi = [&](int t) {
        const float* gt = gi_base + (int64_t)t * S.gi_t;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* g = gt + (int64_t)CR(r) * S.gi_row;
                gr[r] = g[lo_gi]; gu[r] = g[lo_gi + H]; gn[r] = g[lo_gi + 2 * H];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* g = gt + (int64_t)CR(r) * S.gi_row;
                if (CR(r) + lrow < nvalid) { gr[r] = g[lo_gi]; gu[r] = g[lo_gi + H]; gn[r] = g[lo_gi + 2 * H]; }
                else { gr[r] = 0.f; gu[r] = 0.f; gn[r] = 0.f; }
            }
        }
    };
    if (!XIN) load_gi(time_of(0));
    f32x4 wq[PD][3];
    {
        const float4* w0 = XIN ? wpx : wp;
#pragma unroll
        for (int c = 0; c < PD; ++c) { RING_LOAD_U(wq[c][0], w0 + (c) * 3 * 64, lane16, 0); RING_LOAD_U(wq[c][1], w0 + (c) * 3 * 64, lane16, 1024); RING_LOAD_U(wq[c][2], w0 + (c) * 3 * 64, lane16, 2048); }
    }
    // a zero initial state contributes nothing to step 0: its K loops are skipped (acc + 0 * w = acc), the ring keeps its first chunks
    const bool skip0 = S.h0 == nullptr;
    if (XIN) __syncthreads();                // xs[0] is complete
    GRU_PHASE_DECL();
    // The skew is ONE barrier: group Y passes an extra barrier before its first step, group X one after its last.  Both then run the same
    // loop -- part 0, barrier, part 1, barrier -- and X's k-th barrier is Y's (k+1)-th, so Y's part 0 of step s runs beside X's part 1 of
    // step s, Y's part 1 beside X's part 0 of step s + 1 (2 T + 1 barriers for every wave).
    if (grp == 1) LDS_BARRIER();
    for (int s = 0; s < T; ++s) {
        const int t = time_of(s);
        const bool skip = s == 0 && skip0;
        const float* hrow = &hs[s & 1][li * LDH + 4 * hh];
        if (s >= 1 && y_tile) store_h(hs[s & 1], time_of(s - 1));      // h_{s-1} of this group's columns (complete since the last barrier)
        // The partner wave on this SIMD is entering part 1: K loop, then gate math.  Two K loops side by side advance at the same rate and
        // end together (the matrix pipe alternates between the waves whatever their priority: measured, profiles/r04_skew_probe.txt) --
        // and the partner's gate math would then run with the pipe idle.  Starting part 0 late by about the length of that gate math lets
        // the partner's loop run alone first and end early; this wave's loop then has the pipe to itself while the partner does its VALU work.
        for (int z = 0; z < delay; ++z) VAME_SLEEP4();
        GRU_PHASE(0);
        // ---------------------------------------------------------------- part 0
        f32x16 ar, au, ani, anh;                 // the gate accumulators live across the barrier between the two parts
        if (XIN) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { ar[r] = bgr; au[r] = bgu; ani[r] = bgn; anh[r] = bhn; }
            if (grp == 0 && s + 1 < T) load_x(time_of(s + 1));
            const float* xrow = &xs[s & 1][li * LDX + 4 * hh];
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * j);
                RING_WAIT3(3 * (PD - 1), wq[j][0], wq[j][1], wq[j][2]);
                const f32x4 b0 = wq[j][0], b1 = wq[j][1], b2 = wq[j][2];
                ar = MFMA_32x32x2(a.x, b0[0], ar); au = MFMA_32x32x2(a.x, b1[0], au); ani = MFMA_32x32x2(a.x, b2[0], ani);
                ar = MFMA_32x32x2(a.y, b0[1], ar); au = MFMA_32x32x2(a.y, b1[1], au); ani = MFMA_32x32x2(a.y, b2[1], ani);
                ar = MFMA_32x32x2(a.z, b0[2], ar); au = MFMA_32x32x2(a.z, b1[2], au); ani = MFMA_32x32x2(a.z, b2[2], ani);
                ar = MFMA_32x32x2(a.w, b0[3], ar); au = MFMA_32x32x2(a.w, b1[3], au); ani = MFMA_32x32x2(a.w, b2[3], ani);
                RING_FENCE();
                const float4* nsrc = skip ? wpx : wp;          // no recurrent loop in a zero-state first step: keep the input chunks
                RING_LOAD_U(wq[j][0], nsrc + (j) * 3 * 64, lane16, 0); RING_LOAD_U(wq[j][1], nsrc + (j) * 3 * 64, lane16, 1024); RING_LOAD_U(wq[j][2], nsrc + (j) * 3 * 64, lane16, 2048);
            }
        } else {
            ar = gr; au = gu; ani = gn;
#pragma unroll
            for (int r = 0; r < 16; ++r) anh[r] = bhn;
        }
        // one group of PD chunks of the recurrent K loop: chunk c = c0 + j from ring slot j, refilled behind its last use with chunk
        // c + PD (the last group of a step wraps into the next step's first chunks: the input-projection chunks when XIN)
        auto k_group = [&](int c0) {
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const int c = c0 + j;
                const float4 a = *reinterpret_cast<const float4*>(hrow + 8 * c);
                RING_WAIT3(3 * (PD - 1), wq[j][0], wq[j][1], wq[j][2]);
                const f32x4 b0 = wq[j][0], b1 = wq[j][1], b2 = wq[j][2];
                ar = MFMA_32x32x2(a.x, b0[0], ar); au = MFMA_32x32x2(a.x, b1[0], au); anh = MFMA_32x32x2(a.x, b2[0], anh);
                ar = MFMA_32x32x2(a.y, b0[1], ar); au = MFMA_32x32x2(a.y, b1[1], au); anh = MFMA_32x32x2(a.y, b2[1], anh);
                ar = MFMA_32x32x2(a.z, b0[2], ar); au = MFMA_32x32x2(a.z, b1[2], au); anh = MFMA_32x32x2(a.z, b2[2], anh);
                ar = MFMA_32x32x2(a.w, b0[3], ar); au = MFMA_32x32x2(a.w, b1[3], au); anh = MFMA_32x32x2(a.w, b2[3], anh);
                RING_FENCE();
                {
                    const bool wrap = c0 + PD == KC;
                    const float4* src = (XIN && wrap) ? wpx : wp;
                    const int cn = wrap ? j : c + PD;
                    RING_LOAD_U(wq[j][0], src + (cn) * 3 * 64, lane16, 0); RING_LOAD_U(wq[j][1], src + (cn) * 3 * 64, lane16, 1024); RING_LOAD_U(wq[j][2], src + (cn) * 3 * 64, lane16, 2048);
                }
            }
        };
        if (!skip) {
#pragma unroll 1
            for (int c0 = 0; c0 < KH; c0 += PD) k_group(c0);
        }
        GRU_PHASE(1);
        LDS_BARRIER();
        GRU_PHASE(4);
        // ---------------------------------------------------------------- part 1
        if (prio) SETPRIO(1);
        if (!skip) {
#pragma unroll 1
            for (int c0 = KH; c0 < KC; c0 += PD) k_group(c0);
        }
        // gi of this wave's next step: requested behind the loop's last ring wait (in front of it, the ring's vmcnt(9) waits would
        // wait for these loads too); it arrives beside the gate math, the barrier and the partner wave's MFMAs
        if (!XIN && s + 1 < T && S.gi_t != 0) load_gi(time_of(s + 1));
        GRU_PHASE(2);
        float* hnext = &hs[(s + 1) & 1][lrow * LDH + col0 + li];
        f32x16 ust;
        float4* sp = stash ? stash + ((((int64_t)tile * T + t) * NW + w) * 20) * 64 + lane : nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * q + j;
                const float rr = fast_sigmoid(ar[r]);
                const float uu = fast_sigmoid(au[r]);
                const float nn = fast_tanh(ani[r] + rr * anh[r]);
                const float hp = hprev[r];
                const float hv = nn + uu * (hp - nn);
                const float omu = 1.0f - uu;
                ani[r] = omu * (1.0f - nn * nn);          // cA
                au[r] = (hp - nn) * uu * omu;             // cB
                ar[r] = rr;
                ust[r] = uu;
                hprev[r] = hv;
                hnext[CR(r) * LDH] = hv;
            }
            if (stash) {
                const float4 v0 = make_float4(ani[4 * q], ani[4 * q + 1], ani[4 * q + 2], ani[4 * q + 3]);
                const float4 v1 = make_float4(au[4 * q], au[4 * q + 1], au[4 * q + 2], au[4 * q + 3]);
                const float4 v2 = make_float4(ust[4 * q], ust[4 * q + 1], ust[4 * q + 2], ust[4 * q + 3]);
                const float4 v3 = make_float4(ar[4 * q], ar[4 * q + 1], ar[4 * q + 2], ar[4 * q + 3]);
                const float4 v4 = make_float4(anh[4 * q], anh[4 * q + 1], anh[4 * q + 2], anh[4 * q + 3]);
                sp[(0 * 4 + q) * 64] = v0; sp[(1 * 4 + q) * 64] = v1; sp[(2 * 4 + q) * 64] = v2; sp[(3 * 4 + q) * 64] = v3; sp[(4 * 4 + q) * 64] = v4;
            }
            SCHED_FENCE();
        }
        if (XIN && grp == 0 && s + 1 < T) store_x(xs[(s + 1) & 1]);
        if (prio) SETPRIO(0);
        GRU_PHASE(3);
        LDS_BARRIER();
        GRU_PHASE(5);
    }
    if (grp == 0) LDS_BARRIER();
    if (y_tile) store_h(hs[T & 1], time_of(T - 1));                  // the last step's half (complete since this wave's last in-loop barrier)
    GRU_PHASE_END();
    if (S.hn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int grow = row0 + CR(r) + lrow;
            if (grow < B) S.hn[(int64_t)grow * S.hn_row + col0 + li] = hprev[r];
        }
    }
    GRU_PROBE_END();
}

// ------------------------------------------------------------------------------------------- backward
// ABL: 1 no dG stores, 2 no stash/dy loads (first step's reused), 16 W fragments not re-streamed, 32 no barriers, 256 / 1024 dG copy-out placement (see the step loop)
template <int H, int ABL = 0>
__global__ __launch_bounds__(H / 32 * 64) void gru_seq_bwd_kernel(GruBwdParams P) {
    constexpr int NW = H / 32, K3 = 3 * H, LDG = 4 * H + 4, KC = K3 / 8;
    __shared__ float gs[32 * LDG];      // per row [da_r | da_z | dgh_n | dgi_n]: first 3H = MFMA A operand
    int sidx, tile;
    if (!map_block(P.nstreams, P.ntiles, sidx, tile)) return;
    GRU_PROBE_BEGIN();
    const GruBwdStream& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    const int row0 = tile * 32, col0 = 32 * w, lrow = 4 * hh;
    const int nvalid = B - row0;
    const bool full = nvalid >= 32;
    const int lo_dy = lrow * (int)S.dy_row + li;
    const float* dy_base = S.dy ? S.dy + (int64_t)row0 * S.dy_row + col0 : nullptr;
    // dG copy pass: thread -> (row = tid / H + 2*i, float4 column tid % H), i = 0..15; LDS blocks (r,z,gh_n,gi_n) -> global (r,z,gi_n,gh_n)
    const int crow = tid / H, cc = 4 * (tid % H), cblk = cc / H;
    float* dg_copy = S.dg + ((int64_t)(row0 + crow) * T) * 4 * H + (cblk == 2 ? 3 * H : cblk == 3 ? 2 * H : cblk * H) + cc % H;
    const float* gs_copy = &gs[crow * LDG + cc];

    f32x16 dh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int grow = row0 + CR(r) + lrow;
        dh[r] = (S.dhn && grow < B) ? S.dhn[(int64_t)grow * S.dhn_row + col0 + li] : 0.0f;
    }
    float dbs0 = 0.f, dbs1 = 0.f, dbs2 = 0.f, dbs3 = 0.f;
    const float4* __restrict__ wpt = reinterpret_cast<const float4*>(S.wpt) + (int64_t)w * KC * 64;      // uniform ring base (RING_LOAD_U)
    const unsigned lane16 = (unsigned)lane * 16u;
    const float4* stash = reinterpret_cast<const float4*>(S.stash);
    const float* grow_a = &gs[li * LDG + 4 * hh];
    // LDS is > 64 KiB: two lane bases keep every ds_write inside the 16-bit immediate offset range
    float* gw_lo = &gs[lrow * LDG + col0 + li];
    float* gw_hi = gw_lo + 16 * LDG;

    // per-step operands (coefficient stash cA,cB,u,r,gh_n and dy_t), loaded one step ahead: the loads for step
    // s+1 are issued right before the MFMA loop of step s into registers that are dead during it
    float4 sa[4], sb[4], su[4], sr[4], sg[4];
    f32x16 dyv;
    auto load_step = [&](int step) {
        const int fstep = T - 1 - step;
        const int t = S.reverse ? T - 1 - fstep : fstep;
        const float4* sp = stash + ((((int64_t)tile * T + t) * NW + w) * 20) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sa[q] = sp[(0 * 4 + q) * 64]; sb[q] = sp[(1 * 4 + q) * 64]; su[q] = sp[(2 * 4 + q) * 64];
            sr[q] = sp[(3 * 4 + q) * 64]; sg[q] = sp[(4 * 4 + q) * 64];
        }
        const float* dyt = dy_base ? dy_base + (int64_t)t * S.dy_t : nullptr;
        if (dyt && full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dyv[r] = (dyt + (int64_t)CR(r) * S.dy_row)[lo_dy];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dyv[r] = (dyt && CR(r) + lrow < nvalid) ? (dyt + (int64_t)CR(r) * S.dy_row)[lo_dy] : 0.f;
        }
    };
    load_step(0);
    constexpr int PD = 3;                       // W_hh fragment prefetch distance in chunk pairs ((KC/2) % PD == 0 for H = 32..256; 4 measured slower)
    f32x4 wq[PD][2];
#pragma unroll
    for (int c = 0; c < PD; ++c) { RING_LOAD_U(wq[c][0], wpt + (2 * c) * 64, lane16, 0); RING_LOAD_U(wq[c][1], wpt + (2 * c) * 64, lane16, 1024); }
    // dG[b][t][da_r | da_z | dgi_n | dgh_n] leaves through LDS: 16 coalesced 16-byte stores per thread
    auto dg_copy_out = [&](int t) {
        float* dgt = dg_copy + (int64_t)t * 4 * H;
        if (full) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i)
                *reinterpret_cast<float4*>(dgt + (int64_t)(2 * i) * T * 4 * H) =
                    *reinterpret_cast<const float4*>((i < 8 ? gs_copy : gs_copy + 16 * LDG) + 2 * (i & 7) * LDG);
        } else {
#pragma unroll 4
            for (int i = 0; i < 16; ++i)
                if (crow + 2 * i < nvalid)
                    *reinterpret_cast<float4*>(dgt + (int64_t)(2 * i) * T * 4 * H) =
                        *reinterpret_cast<const float4*>((i < 8 ? gs_copy : gs_copy + 16 * LDG) + 2 * (i & 7) * LDG);
        }
    };
    // (ablation 256) the same copy by the first H threads alone (= the older wave of every SIMD pair, NW/2 waves): one full 4H-float row
    // per pass, 32 passes.  Those waves win the MFMA arbitration, leave the loop ~4 k cycles before the others and then idle at barrier 2
    // (per-wave probes, profiles/r02_gru_ablation.txt); letting them copy in that slack was measured 3 % slower than the default.
    const int ecol = 4 * (tid % H), eblk = ecol / H;
    float* dg_early = S.dg + ((int64_t)row0 * T) * 4 * H + (eblk == 2 ? 3 * H : eblk == 3 ? 2 * H : eblk * H) + ecol % H;
    auto dg_copy_out_early = [&](int t) {
        if (tid >= H) return;
        float* dgt = dg_early + (int64_t)t * 4 * H;
#pragma unroll 8
        for (int i = 0; i < 32; ++i)
            if (full || i < nvalid)
                *reinterpret_cast<float4*>(dgt + (int64_t)i * T * 4 * H) = *reinterpret_cast<const float4*>(&gs[i * LDG + ecol]);
    };
    GRU_PHASE_DECL();
    for (int step = 0; step < T; ++step) {
        const int fstep = T - 1 - step;
        const int t = S.reverse ? T - 1 - fstep : fstep;
        f32x16 acc0;
        GRU_PHASE(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float av[4] = {sa[q].x, sa[q].y, sa[q].z, sa[q].w}, bv[4] = {sb[q].x, sb[q].y, sb[q].z, sb[q].w},
                        uv[4] = {su[q].x, su[q].y, su[q].z, su[q].w}, rv[4] = {sr[q].x, sr[q].y, sr[q].z, sr[q].w},
                        gv[4] = {sg[q].x, sg[q].y, sg[q].z, sg[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * q + j;
                const float d = dh[r] + dyv[r];
                const float dan = d * av[j];
                const float dau = d * bv[j];
                const float dgh = dan * rv[j];
                const float dar = dgh * gv[j] * (1.0f - rv[j]);
                acc0[r] = d * uv[j];                               // dh carried through the update gate
                float* gw = (r < 8 ? gw_lo : gw_hi) + (CR(r) & 15) * LDG;
                gw[0] = dar; gw[H] = dau; gw[2 * H] = dgh; gw[3 * H] = dan;
                dbs0 += dar; dbs1 += dau; dbs2 += dan; dbs3 += dgh;
            }
        }
        GRU_PHASE(1);                 // coefficient math + LDS tile writes (incl. the wait for the stash loads)
        if (!(ABL & 32)) __syncthreads();
        GRU_PHASE(2);                 // barrier 1
        // dG copy-out: BEFORE the MFMA loop by all waves -- the loop covers the stores.  Measured alternatives (ablations; a last step
        // that skips the loop copies here in every form): 1024 = after the loop by all waves, 2-3 % slower (the stores delay barrier 2 for
        // the waves that arrive last); 256 = after the loop by the early half of the waves only (dg_copy_out_early), also 3 % slower
        // although those waves idle at barrier 2 for 8 k cycles: their stores then share the memory path with the late waves' loads.
        if (!(ABL & 1) && !(ABL & (256 | 1024))) dg_copy_out(t);
        else if (!(ABL & 1) && step + 1 == T && S.dh0 == nullptr) { if (ABL & 1024) dg_copy_out(t); else dg_copy_out_early(t); }
        GRU_PHASE(3);                 // dG copy-out
        if (!(ABL & 2) && (ABL & 128) && step + 1 < T) load_step(step + 1);      // (ablation: the old place, before the MFMA loop)
        GRU_PHASE(4);                 // next step's stash / dy loads (issue)
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        if (step + 1 == T && S.dh0 == nullptr && !(ABL & 63)) break;      // dh before the first step is not asked for: its 3H x H MACs are skipped
#pragma unroll 1
        for (int c0 = 0; c0 < KC / 2; c0 += PD)
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            if ((ABL & 64) && j == 0 && c0 == (KC / 2 / PD * 3 / 4) * PD && step + 1 < T) load_step(step + 1);
            const int c = 2 * (c0 + j);
            const float4 a0 = *reinterpret_cast<const float4*>(grow_a + 8 * c);
            const float4 a1 = *reinterpret_cast<const float4*>(grow_a + 8 * c + 8);
            RING_WAIT2(2 * (PD - 1), wq[j][0], wq[j][1]);
            const f32x4 b0 = wq[j][0], b1 = wq[j][1];
            acc0 = MFMA_32x32x2(a0.x, b0[0], acc0); acc1 = MFMA_32x32x2(a1.x, b1[0], acc1);
            acc0 = MFMA_32x32x2(a0.y, b0[1], acc0); acc1 = MFMA_32x32x2(a1.y, b1[1], acc1);
            acc0 = MFMA_32x32x2(a0.z, b0[2], acc0); acc1 = MFMA_32x32x2(a1.z, b1[2], acc1);
            acc0 = MFMA_32x32x2(a0.w, b0[3], acc0); acc1 = MFMA_32x32x2(a1.w, b1[3], acc1);
            RING_FENCE();
            {   // refill after the slot's last use (see the forward kernel); wraps into the next step
                const int cn = (ABL & 16) ? 0 : (c0 + j + PD == KC / 2 + j ? 2 * j : c + 2 * PD);
                RING_LOAD_U(wq[j][0], wpt + cn * 64, lane16, 0); RING_LOAD_U(wq[j][1], wpt + cn * 64, lane16, 1024);
            }
            if (j == PD - 1) GRU_PHASE_DYN(8 + c0 / PD);
        }
        // next step's stash / dy loads: issued AFTER the MFMA loop -- before it they sit in front of every weight-fragment load of the
        // loop in the in-order vmcnt queue (measured: 2-3 % slower); their latency overlaps barrier 2 and the other wave's loop tail
        if (!(ABL & 2) && !(ABL & 192) && step + 1 < T) load_step(step + 1);
        if (!(ABL & 1) && (ABL & (256 | 1024))) { if (ABL & 1024) dg_copy_out(t); else dg_copy_out_early(t); }     // (gs is intact until barrier 2)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = acc0[r] + acc1[r];
        GRU_PHASE(5);                 // MFMA loop
        if (!(ABL & 32)) __syncthreads();
        GRU_PHASE(6);                 // barrier 2
    }
    GRU_PHASE_END();
    if (S.dh0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int grow = row0 + CR(r) + lrow;
            if (grow < B) S.dh0[(int64_t)grow * S.dh0_row + col0 + li] = dh[r];
        }
    }
    if (S.dbias) {
        dbs0 += __shfl_xor(dbs0, 32); dbs1 += __shfl_xor(dbs1, 32);
        dbs2 += __shfl_xor(dbs2, 32); dbs3 += __shfl_xor(dbs3, 32);
        if (hh == 0) {
            float* o = S.dbias + (int64_t)tile * 4 * H + col0 + li;
            o[0] = dbs0; o[H] = dbs1; o[2 * H] = dbs2; o[3 * H] = dbs3;
        }
    }
    GRU_PROBE_END();
}


// ------------------------------------------------------------------------------------------- backward, wave-specialised (round 3)
// Same tile (32 batch rows of one stream, all T steps), same stash / dG / dbias formats and the same arithmetic order as
// gru_seq_bwd_kernel -- but the two waves of every SIMD no longer do the same thing in lock step:
//   * waves 0 .. NW/2-1 ("MFMA waves", one per SIMD) own 64 hidden columns each and do nothing but the contraction
//     dh_{t-1} = dh*u + [da_r | da_z | dgh_n] W_hh (K = 3H): their vector-memory queue holds ONLY the L2 weight ring, so no HBM
//     round trip ever sits in front of a weight fragment (the coupling that capped the old kernel, profiles/r02_gru_ablation.txt);
//   * waves NW/2 .. NW-1 ("memory waves", the other wave of each SIMD) own every HBM stream: during the MFMA loop of step s they
//     copy dG of step s out and fetch the coefficient stash + dy of step s+1 into registers -- a whole contraction (~50 k cycles)
//     of cover instead of one barrier's worth -- and between two loops they do the coefficient math for their 64 columns and write
//     the A operand tile.
// Hand-offs: dh (MFMA -> memory) and the update-gate carry dh*u (memory -> MFMA) go through `xd`, an LDS image in accumulator-
// fragment order (each lane re-reads exactly the 16-byte slots its partner lane wrote); the A tile through `gs`.  Two barriers per
// step, as before.  dgi_n no longer passes through LDS (the tile is 3H wide): the memory waves keep it in registers across barrier 1
// and store it beside the contraction.  The coefficient phase is written on packed fp32 pairs (v_pk_mul_f32 / v_pk_add_f32): it
// runs on ONE wave per SIMD while the MFMA waves wait, so its instruction count is step time.  The memory waves' requests are paced
// (s_sleep between small groups): issued as a burst they fill the CU's vector-memory FIFO and the contraction stalls behind them.
// Cycles per step at H = 256 (profiles/r03_ws_probe.txt): coefficient phase 3.4 k, contraction 54.5 k (floor 49.2 k), hand-offs 1 k
// = 58.9 k, against 72 k for the lock-step kernel.
// LDS: 32 x (3H + 4) + 32 x H floats = 131,584 B at H = 256.
// ABL (timing only, wrong results): 1 no dG copy-out, 2 no next-step loads, 4 no MFMA loop, 8 no dgi_n stores
template <int H, int ABL = 0>
__global__ __launch_bounds__(H / 32 * 64) void gru_ws_bwd_kernel(GruBwdParams P) {
    constexpr int NW = H / 32, MW = NW / 2, K3 = 3 * H, LDG = K3 + 4, KC = K3 / 8;
    static_assert(NW % 2 == 0, "wave-specialised BPTT: H a multiple of 64 (an even number of waves)");
    __shared__ float gs[32 * LDG];                          // per row [da_r | da_z | dgh_n]: the MFMA A operand
    __shared__ float4 xd[NW * 4 * 64];                      // [col-block][q][lane]: dh (after barrier 2) / carry (after barrier 1)
    int sidx, tile;
    if (!map_block(P.nstreams, P.ntiles, sidx, tile)) return;
    GRU_PROBE_BEGIN();
    const GruBwdStream& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    const int row0 = tile * 32, lrow = 4 * hh;
    const int nvalid = B - row0;
    const bool full = nvalid >= 32;
    const bool skip_last = S.dh0 == nullptr;               // dh before the first step is not asked for: the last contraction is skipped

    if (w < MW) {
        // ================================================================ MFMA waves: columns [64 w, 64 w + 64)
        const int cbA = 2 * w, cbB = 2 * w + 1;
        // uniform ring bases (SGPR pairs) + one per-lane byte offset: no vector-ALU address arithmetic in the contraction
        const float4* __restrict__ uA = reinterpret_cast<const float4*>(S.wpt) + (int64_t)cbA * KC * 64;
        const float4* __restrict__ uB = uA + (int64_t)KC * 64;
        const unsigned lane16 = (unsigned)lane * 16u;
        const float* grow_a = &gs[li * LDG + 4 * hh];
        constexpr int PD = 3;                               // ring depth in chunk pairs ((KC / 2) % PD == 0 for every H % 64 == 0)
        f32x4 wq[PD][4];
#pragma unroll
        for (int c = 0; c < PD; ++c) {
            RING_LOAD_U(wq[c][0], uA + (2 * c) * 64, lane16, 0); RING_LOAD_U(wq[c][1], uA + (2 * c) * 64, lane16, 1024);
            RING_LOAD_U(wq[c][2], uB + (2 * c) * 64, lane16, 0); RING_LOAD_U(wq[c][3], uB + (2 * c) * 64, lane16, 1024);
        }
        f32x16 a0, b0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; b0[r] = 0.f; }
        __syncthreads();                                    // prologue: xd holds dh_T (the memory waves' load of dhn)
        GRU_PHASE_DECL();
        for (int step = 0; step < T; ++step) {
            __syncthreads();                                // barrier 1: A tile and carry of this step are in LDS
            GRU_PHASE(2);                 // (probe build) MFMA waves: wait for the memory waves' coefficient phase
            if (step + 1 == T && skip_last) break;
            f32x16 a1, b1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 va = xd[(cbA * 4 + q) * 64 + lane], vb = xd[(cbB * 4 + q) * 64 + lane];
                a0[4 * q] = va.x; a0[4 * q + 1] = va.y; a0[4 * q + 2] = va.z; a0[4 * q + 3] = va.w;
                b0[4 * q] = vb.x; b0[4 * q + 1] = vb.y; b0[4 * q + 2] = vb.z; b0[4 * q + 3] = vb.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { a1[r] = 0.f; b1[r] = 0.f; }
            float4 fa0 = *reinterpret_cast<const float4*>(grow_a), fa1 = *reinterpret_cast<const float4*>(grow_a + 8);
            GRU_PHASE(1);                 // carry read, accumulator init
#pragma unroll 1
            for (int c0 = 0; c0 < ((ABL & 4) ? PD : KC / 2); c0 += PD)
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const int c = 2 * (c0 + j);
                const int cnx = (c + 2 < KC) ? c + 2 : 0;   // A fragments of the next chunk pair, requested before this pair's MFMAs
                const float4 na0 = *reinterpret_cast<const float4*>(grow_a + 8 * cnx);
                const float4 na1 = *reinterpret_cast<const float4*>(grow_a + 8 * cnx + 8);
                RING_WAIT4(4 * (PD - 1), wq[j][0], wq[j][1], wq[j][2], wq[j][3]);
                const f32x4 wa0 = wq[j][0], wa1 = wq[j][1], wb0 = wq[j][2], wb1 = wq[j][3];
                a0 = MFMA_32x32x2(fa0.x, wa0[0], a0); a1 = MFMA_32x32x2(fa1.x, wa1[0], a1);
                b0 = MFMA_32x32x2(fa0.x, wb0[0], b0); b1 = MFMA_32x32x2(fa1.x, wb1[0], b1);
                a0 = MFMA_32x32x2(fa0.y, wa0[1], a0); a1 = MFMA_32x32x2(fa1.y, wa1[1], a1);
                b0 = MFMA_32x32x2(fa0.y, wb0[1], b0); b1 = MFMA_32x32x2(fa1.y, wb1[1], b1);
                a0 = MFMA_32x32x2(fa0.z, wa0[2], a0); a1 = MFMA_32x32x2(fa1.z, wa1[2], a1);
                b0 = MFMA_32x32x2(fa0.z, wb0[2], b0); b1 = MFMA_32x32x2(fa1.z, wb1[2], b1);
                a0 = MFMA_32x32x2(fa0.w, wa0[3], a0); a1 = MFMA_32x32x2(fa1.w, wa1[3], a1);
                b0 = MFMA_32x32x2(fa0.w, wb0[3], b0); b1 = MFMA_32x32x2(fa1.w, wb1[3], b1);
                RING_FENCE();
                {   // refill behind the slot's last use; wraps into the next step's first chunks
                    const int cn = (c0 + j + PD == KC / 2 + j) ? 2 * j : c + 2 * PD;
                    RING_LOAD_U(wq[j][0], uA + cn * 64, lane16, 0); RING_LOAD_U(wq[j][1], uA + cn * 64, lane16, 1024);
                    RING_LOAD_U(wq[j][2], uB + cn * 64, lane16, 0); RING_LOAD_U(wq[j][3], uB + cn * 64, lane16, 1024);
                }
                fa0 = na0; fa1 = na1;
            }
            GRU_PHASE(5);                 // contraction
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = a0[r] + a1[r]; b0[r] = b0[r] + b1[r]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xd[(cbA * 4 + q) * 64 + lane] = make_float4(a0[4 * q], a0[4 * q + 1], a0[4 * q + 2], a0[4 * q + 3]);
                xd[(cbB * 4 + q) * 64 + lane] = make_float4(b0[4 * q], b0[4 * q + 1], b0[4 * q + 2], b0[4 * q + 3]);
            }
            GRU_PHASE(4);                 // dh -> xd
            __syncthreads();                                // barrier 2: dh_{t-1} is in xd, the A tile may be overwritten
            GRU_PHASE(6);
        }
        GRU_PHASE_END();
        if (S.dh0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int grow = row0 + CR(r) + lrow;
                if (grow < B) {
                    S.dh0[(int64_t)grow * S.dh0_row + 32 * cbA + li] = a0[r];
                    S.dh0[(int64_t)grow * S.dh0_row + 32 * cbB + li] = b0[r];
                }
            }
        }
    } else {
        // ================================================================ memory waves: all HBM streams + coefficient math
        // Every stream is addressed as a tile-relative BUFFER RANGE (uniform base + scalar offset + one 32-bit lane offset): the flat
        // form costs a 64-bit address pair per row and access, which hipcc keeps live across the step loop (1800 spilled registers
        // in the first version of this kernel).  Rows past the batch fall outside the ranges: their loads return 0, their stores are
        // dropped; an absent dy is a 0-byte range.
        const int k = w - MW;
        const int rows_here = full ? 32 : nvalid;
        const BufRange r_st = buf_range(S.stash + (int64_t)tile * T * NW * 20 * 64 * 4, (uint64_t)T * NW * 20 * 64 * 16);
        const BufRange r_dy = buf_range(S.dy ? S.dy + (int64_t)row0 * S.dy_row : nullptr, (uint64_t)rows_here * (uint64_t)S.dy_row * 4);
        const BufRange r_dg = buf_range(S.dg + (int64_t)row0 * T * 4 * H, (uint64_t)rows_here * T * 4 * H * 4);
        const uint32_t v_dy = ((uint32_t)lrow * (uint32_t)S.dy_row + (uint32_t)li) * 4u;         // dy / dgi_n: accumulator layout
        const uint32_t dy_row_b = UNIFORM((uint32_t)S.dy_row * 4u), dy_t_b = UNIFORM((uint32_t)S.dy_t * 4u);
        const uint32_t v_st = (uint32_t)lane * 16u;                                              // stash: float4 per lane
        const uint32_t v_dn = ((uint32_t)lrow * (uint32_t)T * 4u * H + (uint32_t)li) * 4u;
        const uint32_t dg_row_b = UNIFORM((uint32_t)T * 4u * H * 4u);
        float4 st[2][20];                                    // [col-block][coefficient * 4 + q]: cA, cB, u, r, gh_n
        f32x16 dyv[2];
        // PACING: the memory waves have a whole contraction (~50 k cycles) for 40 loads and 24 + stores per lane.  Issued as one burst
        // they fill the CU's vector-memory FIFO and the LDS queues, and the MFMA waves' weight-ring loads and A-fragment reads wait
        // behind them (measured: contraction 60.7 k cycles per step instead of 49.2 k, profiles/r03_ws_probe.txt); so they go out in
        // small groups with s_sleep in between (pace_* x 256 cycles after each group; 0 in the first call, where nothing overlaps).
        const int pace_cp = P.pace_cp;
        auto load_step = [&](int step, int pace) {
            const int fstep = T - 1 - step;
            const uint32_t t = (uint32_t)(S.reverse ? T - 1 - fstep : fstep);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const uint32_t cb = 2 * k + c2;
                const uint32_t s_st = UNIFORM((t * NW + cb) * 20u * 1024u);
#pragma unroll
                for (int i = 0; i < 20; ++i) {
                    st[c2][i] = buf_load_f32x4(r_st, v_st, s_st + i * 1024u);
                    if (i % 4 == 3) { SCHED_FENCE(); for (int z = 0; z < pace; ++z) VAME_SLEEP4(); }
                }
                const uint32_t s_dy = UNIFORM(t * dy_t_b + cb * 128u);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dyv[c2][r] = buf_load_f32(r_dy, v_dy, s_dy + (uint32_t)CR(r) * dy_row_b);
                    if (r % 8 == 7) { SCHED_FENCE(); for (int z = 0; z < pace; ++z) VAME_SLEEP4(); }
                }
            }
        };
        // dh_T -> xd
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int cb = 2 * k + c2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int grow = row0 + CR(4 * q + j) + lrow;
                    v[j] = (S.dhn && grow < B) ? S.dhn[(int64_t)grow * S.dhn_row + 32 * cb + li] : 0.0f;
                }
                xd[(cb * 4 + q) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        load_step(0, 0);
        float dbs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // LDS -> global copy of the three tile blocks: this wave's rows are k, k + MW, ...; a lane moves float4 number i = lane + 64 it
        // of a row (3H / 4 per row), which lands in global block (0, 1, 3)[i / (H/4)]
        uint32_t v_cp[(K3 / 4 + 63) / 64];
#pragma unroll
        for (int it = 0; it < (K3 / 4 + 63) / 64; ++it) {
            const int col = 4 * (lane + 64 * it), blk = col / H;
            v_cp[it] = col < K3 ? (uint32_t)((blk == 2 ? 3 * H : blk * H) + col % H) * 4u : 0xffffffffu;      // (poisoned: dropped)
        }
        __syncthreads();                                    // prologue
        GRU_PHASE_DECL();
        for (int step = 0; step < T; ++step) {
            const int fstep = T - 1 - step;
            const uint32_t t = (uint32_t)(S.reverse ? T - 1 - fstep : fstep);
            const uint32_t s_t = UNIFORM(t * 4u * H * 4u);
            // ---- coefficient math of this wave's 64 columns: A tile -> gs, carry -> xd, dgi_n straight to dG (the A tile has no
            // room for it and 32 more live registers do not fit next to the 192 of the prefetched operands)
            // Packed fp32 pairs (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction, same results): the phase is bound by
            // the instruction issue of ONE wave per SIMD while the MFMA waves wait, so halving its VALU count shortens the step.
            float dan_keep[2][16];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int cb = 2 * k + c2;
                float* gw_lo = &gs[lrow * LDG + 32 * cb + li];
                float* gw_hi = gw_lo + 16 * LDG;
                f32x2 bs_ar = {0.f, 0.f}, bs_au = {0.f, 0.f}, bs_an = {0.f, 0.f}, bs_gh = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 dhq = xd[(cb * 4 + q) * 64 + lane];
                    const float4 sa = st[c2][0 * 4 + q], sb = st[c2][1 * 4 + q], su = st[c2][2 * 4 + q], sr = st[c2][3 * 4 + q], sg = st[c2][4 * 4 + q];
                    float cy[4];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const f32x2 d = (h2 ? f32x2{dhq.z, dhq.w} : f32x2{dhq.x, dhq.y})
                                        + f32x2{dyv[c2][4 * q + 2 * h2], dyv[c2][4 * q + 2 * h2 + 1]};
                        const f32x2 av = h2 ? f32x2{sa.z, sa.w} : f32x2{sa.x, sa.y}, bv = h2 ? f32x2{sb.z, sb.w} : f32x2{sb.x, sb.y};
                        const f32x2 uv = h2 ? f32x2{su.z, su.w} : f32x2{su.x, su.y}, rv = h2 ? f32x2{sr.z, sr.w} : f32x2{sr.x, sr.y};
                        const f32x2 gv = h2 ? f32x2{sg.z, sg.w} : f32x2{sg.x, sg.y};
                        const f32x2 one = {1.0f, 1.0f};
                        const f32x2 dan = d * av;
                        const f32x2 dau = d * bv;
                        const f32x2 dgh = dan * rv;
                        const f32x2 dar = dgh * gv * (one - rv);
                        const f32x2 c = d * uv;                    // dh carried through the update gate
                        bs_ar += dar; bs_au += dau; bs_an += dan; bs_gh += dgh;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 4 * q + 2 * h2 + e;
                            float* gw = (r < 8 ? gw_lo : gw_hi) + (CR(r) & 15) * LDG;
                            gw[0] = dar[e]; gw[H] = dau[e]; gw[2 * H] = dgh[e];
                            dan_keep[c2][r] = dan[e];
                            cy[2 * h2 + e] = c[e];
                        }
                    }
                    xd[(cb * 4 + q) * 64 + lane] = make_float4(cy[0], cy[1], cy[2], cy[3]);
                    if (q & 1) SCHED_FENCE();               // two row groups at a time (their LDS round trips overlap); their stash registers die here
                    if (c2 == 0 && q == 0) GRU_PHASE(0);    // (probe build) the first row group, i.e. the wait for the prefetched operands
                }
                // bias partials: the lock-step kernel adds element by element in register order; pairs first and then the two halves is a
                // different (equally valid) fp32 summation order of the same 16 x T terms -> dbias agrees to rounding, not bit for bit
                dbs[c2][0] += bs_ar[0] + bs_ar[1]; dbs[c2][1] += bs_au[0] + bs_au[1]; dbs[c2][2] += bs_an[0] + bs_an[1]; dbs[c2][3] += bs_gh[0] + bs_gh[1];
            }
            GRU_PHASE(1);                 // (probe build) memory waves: coefficient phase
            __syncthreads();                                // barrier 1
            GRU_PHASE(2);
            // ---- beside the MFMA loop: dgi_n out of the registers, the three tile blocks of dG out of LDS ...
            if (!(ABL & 8)) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const uint32_t s_dn = UNIFORM(s_t + (2u * H + 32u * (2 * k + c2)) * 4u);
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf_store_f32(r_dg, v_dn, s_dn + (uint32_t)CR(r) * dg_row_b, dan_keep[c2][r]);
                }
                SCHED_FENCE();
            }
            if (!(ABL & 1)) {
#pragma unroll 2
                for (int row = k; row < 32; row += MW) {
                    const uint32_t s_row = UNIFORM(s_t + (uint32_t)row * dg_row_b);
#pragma unroll
                    for (int it = 0; it < (K3 / 4 + 63) / 64; ++it)
                        buf_store_f32x4(r_dg, v_cp[it], s_row, *reinterpret_cast<const float4*>(&gs[row * LDG + (4 * (lane + 64 * it)) % K3]));
                    SCHED_FENCE();
                    for (int z = 0; z < pace_cp; ++z) VAME_SLEEP4();      // pacing: see below
                }
            }
            GRU_PHASE(3);                 // dG copy-out
            if (step + 1 == T && skip_last) break;
            // ---- ... and the stash / dy of the next step in: a whole contraction of cover
            SCHED_FENCE();
            // (unconditional: a conditional load makes every stash register a loop-carried phi of old and new value -- both sets live
            // during the loads, 384 registers; the last step re-reads its own operands instead)
            if (!(ABL & 2)) load_step(step + 1 < T ? step + 1 : step, P.pace_ld);
            GRU_PHASE(4);                 // next step's stash request (issue)
            __syncthreads();                                // barrier 2
            GRU_PHASE(6);
        }
        GRU_PHASE_END();
        if (S.dbias) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                float d0 = dbs[c2][0], d1 = dbs[c2][1], d2 = dbs[c2][2], d3 = dbs[c2][3];
                d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32); d3 += __shfl_xor(d3, 32);
                if (hh == 0) {
                    float* o = S.dbias + (int64_t)tile * 4 * H + 32 * (2 * k + c2) + li;
                    o[0] = d0; o[H] = d1; o[2 * H] = d2; o[3 * H] = d3;
                }
            }
        }
    }
    GRU_PROBE_END();
}

// ------------------------------------------------------------------------------------------- host
#if !defined(VAME_EMU) && (defined(VAME_GEMM_AB) || defined(VAME_PROBE))
// TUNING BUILDS ONLY (make ab / make probe; never the shipped library): ablation masks and kernel / pacing overrides from the environment,
// so that tools/ws_probe.py, ws_abl.py can sweep them under an unchanged Python layer
#include <stdlib.h>
[[maybe_unused]] static int abl_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
static void tuning_overrides(GruBwdParams& Q) {
    if (const char* e = getenv("VAME_GRU_WS")) Q.kernel = atoi(e) == 0 ? VAME_GRU_KERNEL_LOCKSTEP : atoi(e) >= 2 ? VAME_GRU_KERNEL_WS : VAME_GRU_KERNEL_AUTO;
    if (const char* e = getenv("VAME_WS_PACE_CP")) Q.pace_cp = atoi(e);
    if (const char* e = getenv("VAME_WS_PACE_LD")) Q.pace_ld = atoi(e);
}
static void tuning_overrides(GruFwdParams& Q) {
    if (const char* e = getenv("VAME_GRU_FWD")) Q.kernel = atoi(e);
    if (const char* e = getenv("VAME_GRU_FWD_PRIO")) Q.pace_cp = atoi(e);
    if (const char* e = getenv("VAME_GRU_FWD_DELAY")) Q.pace_ld = atoi(e);
}
#define GRU_TUNING_OVERRIDES(Q) tuning_overrides(Q)
#else
#define GRU_TUNING_OVERRIDES(Q)
#endif
#define ABL_CASE(K, H, A, P, st) case A: hipLaunchKernelGGL((K<H, A>), dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P); return;

// Kernel choice for BPTT (GB_OPT of stream 0, include/vame_hip.h): AUTO = the wave-specialised kernel at H = 256 (102 -> 120 TF at batch
// 4096, faster down to batch 100) and H = 192 (+18 %), the lock-step kernel elsewhere (H = 128 / 64: a step's contraction is 4x / 16x
// shorter and the serial coefficient phase weighs more -- measured 12-16 % slower; instantiated for every multiple of 64 for the tests and
// for callers that ask).  tools/bwd_table.py, profiles/r04_bwd_table.txt.
template <int H> static constexpr bool gru_ws_instantiated() { return H % 64 == 0; }
static bool gru_ws_auto(int H) { return H == 256 || H == 192; }      // measured per hidden size: profiles/r04_bwd_table.txt
// forward: AUTO = the skewed kernel where it measured faster (tools/fwd_table.py, profiles/r04_fwd_table.txt): H = 256 and 192 for streams
// that read something every step (a per-step gi tile: +6-7 %, the fused input projection: +3-4 %); streams with a time-constant gi (the
// decoders) gain nothing (-0.5 %) and H <= 128 loses 1-4 % (a step's K loop is too short for the second barrier) -> lock-step there
static bool gru_skew_auto(int H, const GruFwdParams& P) {
    if (H != 256 && H != 192) return false;
    for (int i = 0; i < P.nstreams; ++i)
        if (P.s[i].xf == 0 && P.s[i].gi_t == 0) return false;
    return true;
}

template <int H>
static void launch_fwd(const GruFwdParams& P_, hipStream_t st) {
    GruFwdParams P = P_;
    GRU_TUNING_OVERRIDES(P);
#if !defined(VAME_EMU) && defined(VAME_GEMM_AB)
    if (H == 256) switch (abl_env("VAME_ABL_FWD")) {      // profiling-only ablations, tuning build (make ab) only
        ABL_CASE(gru_seq_fwd_kernel, 256, 1, P, st) ABL_CASE(gru_seq_fwd_kernel, 256, 2, P, st) ABL_CASE(gru_seq_fwd_kernel, 256, 4, P, st)
        ABL_CASE(gru_seq_fwd_kernel, 256, 8, P, st) ABL_CASE(gru_seq_fwd_kernel, 256, 16, P, st) ABL_CASE(gru_seq_fwd_kernel, 256, 32, P, st)
        ABL_CASE(gru_seq_fwd_kernel, 256, 7, P, st) ABL_CASE(gru_seq_fwd_kernel, 256, 63, P, st) ABL_CASE(gru_seq_fwd_kernel, 256, 256, P, st) ABL_CASE(gru_seq_fwd_kernel, 256, 512, P, st)
        default: break;
    }
#endif
    if constexpr (H % 64 == 0) {
        // two wave groups half a step apart (gru_skew_fwd_kernel): bit-identical to the lock-step kernel
        if (P.kernel == VAME_GRU_KERNEL_SKEWED || (P.kernel == VAME_GRU_KERNEL_AUTO && gru_skew_auto(H, P))) {
            if (P.s[0].xf > 0) hipLaunchKernelGGL((gru_skew_fwd_kernel<H, true>), dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
            else hipLaunchKernelGGL((gru_skew_fwd_kernel<H, false>), dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
            return;
        }
    }
    if (P.s[0].xf > 0) hipLaunchKernelGGL((gru_seq_fwd_kernel<H, 0, true>), dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
    else hipLaunchKernelGGL((gru_seq_fwd_kernel<H, 0, false>), dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
}
template <int H>
static void launch_bwd(const GruBwdParams& P_, hipStream_t st) {
    GruBwdParams P = P_;
    GRU_TUNING_OVERRIDES(P);
#if !defined(VAME_EMU) && defined(VAME_GEMM_AB)
    if (H == 256) switch (abl_env("VAME_ABL_BWD")) {
        ABL_CASE(gru_seq_bwd_kernel, 256, 1, P, st) ABL_CASE(gru_seq_bwd_kernel, 256, 2, P, st) ABL_CASE(gru_seq_bwd_kernel, 256, 3, P, st)
        ABL_CASE(gru_seq_bwd_kernel, 256, 16, P, st) ABL_CASE(gru_seq_bwd_kernel, 256, 32, P, st) ABL_CASE(gru_seq_bwd_kernel, 256, 51, P, st)
        ABL_CASE(gru_seq_bwd_kernel, 256, 64, P, st) ABL_CASE(gru_seq_bwd_kernel, 256, 128, P, st) ABL_CASE(gru_seq_bwd_kernel, 256, 256, P, st) ABL_CASE(gru_seq_bwd_kernel, 256, 1024, P, st)
        default: break;
    }
#endif
    if constexpr (gru_ws_instantiated<H>()) {
        // wave-specialised BPTT (gru_ws_bwd_kernel) for the hidden sizes it is instantiated for
        if (P.kernel == VAME_GRU_KERNEL_WS || (P.kernel == VAME_GRU_KERNEL_AUTO && gru_ws_auto(H))) {
            GruBwdParams Q = P;
            // pacing of the memory waves (units of 256 cycles per request group), sized so that their 8 copy + 14 load groups span
            // about 90 % of one contraction (54 k cycles at H = 256, a quarter of that at H = 128) and never outlast it: measured
            // optimum 2 / 8 at H = 256 (profiles/r03_ws_probe.txt: 7 .. 9 within noise, 10 and more make the MFMA waves wait)
            if (Q.pace_cp < 0) Q.pace_cp = 2 * H * H / 65536;
            if (Q.pace_ld < 0) Q.pace_ld = 8 * H * H / 65536;
            const GruBwdParams& P = Q;
#if !defined(VAME_EMU) && defined(VAME_GEMM_AB)
            if (H == 256) switch (abl_env("VAME_WS_ABL")) {
                ABL_CASE(gru_ws_bwd_kernel, 256, 1, P, st) ABL_CASE(gru_ws_bwd_kernel, 256, 2, P, st) ABL_CASE(gru_ws_bwd_kernel, 256, 3, P, st)
                ABL_CASE(gru_ws_bwd_kernel, 256, 4, P, st) ABL_CASE(gru_ws_bwd_kernel, 256, 8, P, st) ABL_CASE(gru_ws_bwd_kernel, 256, 11, P, st) ABL_CASE(gru_ws_bwd_kernel, 256, 7, P, st)
                default: break;
            }
#endif
            hipLaunchKernelGGL((gru_ws_bwd_kernel<H, 0>), dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
            return;
        }
    }
    hipLaunchKernelGGL((gru_seq_bwd_kernel<H, 0>), dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
}

extern "C" int vame_gru_seq_fwd_has_kernel(int H, int kernel) {
    if (H < 32 || H > 256 || H % 32) return 0;
    if (kernel == VAME_GRU_KERNEL_SKEWED) return H % 64 == 0;
    return kernel == VAME_GRU_KERNEL_AUTO || kernel == VAME_GRU_KERNEL_LOCKSTEP;
}

extern "C" int vame_gru_seq_fwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream) {
    VAME_CHECK_ARG(desc && nstreams >= 1 && nstreams <= 8, VAME_E_BADARG, "gru_seq_fwd: nstreams=%d not in 1..8", nstreams);
    VAME_CHECK_ARG(B >= 1, VAME_E_SHAPE, "gru_seq_fwd: empty batch");
    GruFwdParams P;
    if (int rc = gru_parse_fwd(desc, nstreams, B, P)) return rc;
    VAME_CHECK_ARG(vame_gru_seq_fwd_has_kernel(H, P.kernel) || H > 256 || H % 32, VAME_E_UNSUPPORTED,
                   "gru_seq_fwd: kernel option %d is not instantiated for H=%d", P.kernel, H);
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 32: launch_fwd<32>(P, st); break;
        case 64: launch_fwd<64>(P, st); break;
        case 96: launch_fwd<96>(P, st); break;
        case 128: launch_fwd<128>(P, st); break;
        case 160: launch_fwd<160>(P, st); break;
        case 192: launch_fwd<192>(P, st); break;
        case 224: launch_fwd<224>(P, st); break;
        case 256: launch_fwd<256>(P, st); break;
        default: VAME_CHECK_ARG(false, VAME_E_UNSUPPORTED, "gru_seq_fwd: H=%d unsupported (multiples of 32 up to 256)", H);
    }
    VAME_LAUNCH_CHECK("gru_seq_fwd");
    return VAME_OK;
}

extern "C" int vame_gru_seq_bwd_has_kernel(int H, int kernel) {
    if (H < 32 || H > 256 || H % 32) return 0;
    if (kernel == VAME_GRU_KERNEL_WS) return H % 64 == 0;
    return kernel == VAME_GRU_KERNEL_AUTO || kernel == VAME_GRU_KERNEL_LOCKSTEP;
}

extern "C" int vame_gru_seq_bwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream) {
    VAME_CHECK_ARG(desc && nstreams >= 1 && nstreams <= 8, VAME_E_BADARG, "gru_seq_bwd: nstreams=%d not in 1..8", nstreams);
    VAME_CHECK_ARG(B >= 1, VAME_E_SHAPE, "gru_seq_bwd: empty batch");
    GruBwdParams P;
    if (int rc = gru_parse_bwd(desc, nstreams, B, P)) return rc;
    VAME_CHECK_ARG(vame_gru_seq_bwd_has_kernel(H, P.kernel) || H > 256 || H % 32, VAME_E_UNSUPPORTED,
                   "gru_seq_bwd: kernel option %d is not instantiated for H=%d", P.kernel, H);
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 32: launch_bwd<32>(P, st); break;
        case 64: launch_bwd<64>(P, st); break;
        case 96: launch_bwd<96>(P, st); break;
        case 128: launch_bwd<128>(P, st); break;
        case 160: launch_bwd<160>(P, st); break;
        case 192: launch_bwd<192>(P, st); break;
        case 224: launch_bwd<224>(P, st); break;
        case 256: launch_bwd<256>(P, st); break;
        default: VAME_CHECK_ARG(false, VAME_E_UNSUPPORTED, "gru_seq_bwd: H=%d unsupported (multiples of 32 up to 256)", H);
    }
    VAME_LAUNCH_CHECK("gru_seq_bwd");
    return VAME_OK;
}
