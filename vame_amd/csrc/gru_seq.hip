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

struct GruFwdStream {
    const float* gi; int64_t gi_row, gi_t;
    const float* wp; const float* bhn;
    const float* h0; int64_t h0_row;
    float* y; int64_t y_row, y_t;
    float* hn; int64_t hn_row;
    float* stash;
    int64_t T, reverse, pad;
};
struct GruFwdParams { GruFwdStream s[8]; int nstreams; int B; int ntiles; };

struct GruBwdStream {
    const float* stash; const float* y; int64_t y_row, y_t;
    const float* h0; int64_t h0_row;
    const float* wpt;
    const float* dy; int64_t dy_row, dy_t;
    const float* dhn; int64_t dhn_row;
    float* dg;
    float* dh0; int64_t dh0_row;
    float* dbias;
    int64_t T, reverse, pad;
};
struct GruBwdParams { GruBwdStream s[8]; int nstreams; int B; int ntiles; };

// blockIdx -> (stream, tile).  Workgroup b is dispatched to XCD b%8 (observed, speed only): when the
// stream count divides 8 each XCD serves a single stream so its 4 MiB L2 keeps that stream's W_hh.
__device__ __forceinline__ bool map_block(int nstreams, int ntiles, int& s, int& tile) {
    const int bid = blockIdx.x;
    if (8 % nstreams == 0) {
        const int xcd = bid & 7, q = bid >> 3, per = 8 / nstreams;
        s = xcd % nstreams;
        tile = q * per + xcd / nstreams;
    } else {
        s = bid % nstreams;
        tile = bid / nstreams;
    }
    return tile < ntiles;
}
static int grid_blocks(int nstreams, int ntiles) {
    if (8 % nstreams == 0) { const int per = 8 / nstreams; return (int)cdiv64(ntiles, per) * 8; }
    return nstreams * ntiles;
}

// ------------------------------------------------------------------------------------------- pack
// wp_fwd[(((w*(H/8)+c)*3+g)*64+l)*4+e] = W_hh[(g*H+32w+(l&31))*H + 8c+4(l>>5)+e]
// wp_bwd[((w*(3H/8)+c)*64+l)*4+e]       = W_hh[(8c+4(l>>5)+e)*H + 32w+(l&31)]
__global__ __launch_bounds__(256) void gru_pack_kernel(const float* __restrict__ W, const float* __restrict__ b_ih,
                                                       const float* __restrict__ b_hh, int H, float* __restrict__ wpf,
                                                       float* __restrict__ wpb, float* __restrict__ bias_gi,
                                                       float* __restrict__ bhn) {
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

extern "C" int64_t vame_gru_stash_floats(int B, int T, int H) {
    return cdiv64(B, 32) * 32 * (int64_t)T * 4 * H;
}

// ------------------------------------------------------------------------------------------- forward
// Addressing: in the 32x32 accumulator layout register r of lane l holds row CR(r) + 4*(l>>5), column l&31.
// Every row-major operand is therefore addressed as  uniform_row_pointer(r) [ lane_offset ]  with
// lane_offset = 4*(l>>5)*row_stride + (l&31): one VGPR per array, row pointers stay in SGPRs.
#define CR(r) (((r) & 3) + 8 * ((r) >> 2))

#ifdef VAME_EMU
#define UNIFORM(x) (x)
#else
#define UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// stash index: ((((tile*T + t)*NW + w)*4 + gate)*4 + rq)*64 + lane   (float4 units), gate = r,u,n,gh_n
template <int H>
__global__ __launch_bounds__(H / 32 * 64) void gru_seq_fwd_kernel(GruFwdParams P) {
    constexpr int NW = H / 32, LDH = H + 4, KC = H / 8;
    __shared__ float hs[2][32 * LDH];
    int sidx, tile;
    if (!map_block(P.nstreams, P.ntiles, sidx, tile)) return;
    const GruFwdStream& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    const int row0 = tile * 32, col0 = 32 * w;
    const int nvalid = B - row0;                        // rows of this tile inside the batch (>= 32: full tile)
    const bool full = nvalid >= 32;
    const int lrow = 4 * hh;                            // lane part of the fragment row
    const int lo_gi = lrow * (int)S.gi_row + li, lo_y = lrow * (int)S.y_row + li;
    const float* gi_base = S.gi + (int64_t)row0 * S.gi_row + col0;
    float* y_base = S.y ? S.y + (int64_t)row0 * S.y_row + col0 : nullptr;

    f32x16 hprev;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = CR(r) + lrow, grow = row0 + row;
        float v = 0.0f;
        if (S.h0 && grow < B) v = S.h0[(int64_t)grow * S.h0_row + col0 + li];
        hprev[r] = v;
        hs[0][row * LDH + col0 + li] = v;
        if (y_base && S.pad && grow < B) (y_base + (int64_t)CR(r) * S.y_row + (int64_t)(S.reverse ? T : -1) * S.y_t)[lo_y] = v;
    }
    __syncthreads();
    const float bhn = S.bhn[col0 + li];
    const float4* __restrict__ wp = reinterpret_cast<const float4*>(S.wp) + (int64_t)w * KC * 3 * 64 + lane;
    float4* stash = S.stash ? reinterpret_cast<float4*>(S.stash) : nullptr;
    int cur = 0;
    // gi of the first step; later steps are prefetched during the previous step's MFMA loop
    f32x16 gr, gu, gn;
    auto load_gi = [&](int t) {
        const float* gt = gi_base + (int64_t)t * S.gi_t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* g = gt + (int64_t)CR(r) * S.gi_row;
            if (full || CR(r) + lrow < nvalid) { gr[r] = g[lo_gi]; gu[r] = g[lo_gi + H]; gn[r] = g[lo_gi + 2 * H]; }
            else { gr[r] = 0.f; gu[r] = 0.f; gn[r] = 0.f; }
        }
    };
    load_gi(S.reverse ? T - 1 : 0);
    for (int step = 0; step < T; ++step) {
        const int t = S.reverse ? T - 1 - step : step;
        f32x16 ar = gr, au = gu, ani = gn, anh;
#pragma unroll
        for (int r = 0; r < 16; ++r) anh[r] = bhn;
        if (step + 1 < T && S.gi_t != 0) load_gi(S.reverse ? t - 1 : t + 1);     // in flight during the k-loop
        const float* hrow = &hs[cur][li * LDH + 4 * hh];
        // software pipeline: fragments of chunk c+1 are requested before the 12 MFMAs of chunk c
        float4 a = *reinterpret_cast<const float4*>(hrow);
        float4 b0 = wp[0], b1 = wp[64], b2 = wp[128];
#pragma unroll 2
        for (int c = 0; c < KC; ++c) {
            const int cn = c + 1 < KC ? c + 1 : c;
            const float4 an = *reinterpret_cast<const float4*>(hrow + 8 * cn);
            const float4 b0n = wp[(cn * 3 + 0) * 64], b1n = wp[(cn * 3 + 1) * 64], b2n = wp[(cn * 3 + 2) * 64];
            ar = MFMA_32x32x2(a.x, b0.x, ar); au = MFMA_32x32x2(a.x, b1.x, au); anh = MFMA_32x32x2(a.x, b2.x, anh);
            ar = MFMA_32x32x2(a.y, b0.y, ar); au = MFMA_32x32x2(a.y, b1.y, au); anh = MFMA_32x32x2(a.y, b2.y, anh);
            ar = MFMA_32x32x2(a.z, b0.z, ar); au = MFMA_32x32x2(a.z, b1.z, au); anh = MFMA_32x32x2(a.z, b2.z, anh);
            ar = MFMA_32x32x2(a.w, b0.w, ar); au = MFMA_32x32x2(a.w, b1.w, au); anh = MFMA_32x32x2(a.w, b2.w, anh);
            a = an; b0 = b0n; b1 = b1n; b2 = b2n;
        }
        float* hnext = &hs[cur ^ 1][lrow * LDH + col0 + li];
        float* yt = y_base ? y_base + (int64_t)t * S.y_t : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float rr = fast_sigmoid(ar[r]);
            const float uu = fast_sigmoid(au[r]);
            const float nn = fast_tanh(ani[r] + rr * anh[r]);
            const float hv = nn + uu * (hprev[r] - nn);
            ar[r] = rr; au[r] = uu; ani[r] = nn;
            hprev[r] = hv;
            hnext[CR(r) * LDH] = hv;
            if (yt && (full || CR(r) + lrow < nvalid)) (yt + (int64_t)CR(r) * S.y_row)[lo_y] = hv;
        }
        if (stash) {
            float4* sp = stash + ((((int64_t)tile * T + t) * NW + w) * 16) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sp[(0 * 4 + q) * 64] = make_float4(ar[4 * q], ar[4 * q + 1], ar[4 * q + 2], ar[4 * q + 3]);
                sp[(1 * 4 + q) * 64] = make_float4(au[4 * q], au[4 * q + 1], au[4 * q + 2], au[4 * q + 3]);
                sp[(2 * 4 + q) * 64] = make_float4(ani[4 * q], ani[4 * q + 1], ani[4 * q + 2], ani[4 * q + 3]);
                sp[(3 * 4 + q) * 64] = make_float4(anh[4 * q], anh[4 * q + 1], anh[4 * q + 2], anh[4 * q + 3]);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (S.hn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int grow = row0 + CR(r) + lrow;
            if (grow < B) S.hn[(int64_t)grow * S.hn_row + col0 + li] = hprev[r];
        }
    }
}

// ------------------------------------------------------------------------------------------- backward
template <int H>
__global__ __launch_bounds__(H / 32 * 64) void gru_seq_bwd_kernel(GruBwdParams P) {
    constexpr int NW = H / 32, K3 = 3 * H, LDG = K3 + 4, KC = K3 / 8;
    __shared__ float gs[32 * LDG];
    int sidx, tile;
    if (!map_block(P.nstreams, P.ntiles, sidx, tile)) return;
    const GruBwdStream& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    const int row0 = tile * 32, col0 = 32 * w, lrow = 4 * hh;
    const int nvalid = B - row0;
    const bool full = nvalid >= 32;
    const int lo_y = lrow * (int)S.y_row + li, lo_dy = lrow * (int)S.dy_row + li, lo_dg = lrow * T * 4 * H + li;
    const float* y_base = S.y + (int64_t)row0 * S.y_row + col0;
    const float* dy_base = S.dy ? S.dy + (int64_t)row0 * S.dy_row + col0 : nullptr;
    float* dg_base = S.dg + (int64_t)row0 * T * 4 * H + col0;

    f32x16 dh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int grow = row0 + CR(r) + lrow;
        dh[r] = (S.dhn && grow < B) ? S.dhn[(int64_t)grow * S.dhn_row + col0 + li] : 0.0f;
    }
    float dbs0 = 0.f, dbs1 = 0.f, dbs2 = 0.f, dbs3 = 0.f;
    const float4* __restrict__ wpt = reinterpret_cast<const float4*>(S.wpt) + (int64_t)w * KC * 64 + lane;
    const float4* stash = reinterpret_cast<const float4*>(S.stash);
    const float* grow_a = &gs[li * LDG + 4 * hh];
    float* gs_w = &gs[lrow * LDG + col0 + li];

    // per-step operands (forward stash r,u,n,gh_n; h_{t-1}; dy_t), loaded one step ahead: the loads for
    // step s+1 are issued right before the MFMA loop of step s into registers that are dead during it
    float4 sr[4], su[4], sn[4], sg[4];
    f32x16 hpv, dyv;
    auto load_step = [&](int step) {
        const int fstep = T - 1 - step;
        const int t = S.reverse ? T - 1 - fstep : fstep;
        const int tprev = S.reverse ? t + 1 : t - 1;
        const float4* sp = stash + ((((int64_t)tile * T + t) * NW + w) * 16) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sr[q] = sp[(0 * 4 + q) * 64]; su[q] = sp[(1 * 4 + q) * 64]; sn[q] = sp[(2 * 4 + q) * 64]; sg[q] = sp[(3 * 4 + q) * 64];
        }
        const bool have_prev = fstep > 0 || S.pad;
        const float* yp = y_base + (int64_t)tprev * S.y_t;
        const float* dyt = dy_base ? dy_base + (int64_t)t * S.dy_t : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float hp = 0.f, dv = 0.f;
            if (full || CR(r) + lrow < nvalid) {
                if (have_prev) hp = (yp + (int64_t)CR(r) * S.y_row)[lo_y];
                else if (S.h0) hp = S.h0[(int64_t)(row0 + CR(r) + lrow) * S.h0_row + col0 + li];
                if (dyt) dv = (dyt + (int64_t)CR(r) * S.dy_row)[lo_dy];
            }
            hpv[r] = hp; dyv[r] = dv;
        }
    };
    load_step(0);
    for (int step = 0; step < T; ++step) {
        const int fstep = T - 1 - step;
        const int t = S.reverse ? T - 1 - fstep : fstep;
        float* dgt = dg_base + (int64_t)t * 4 * H;
        f32x16 dhp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float rv[4] = {sr[q].x, sr[q].y, sr[q].z, sr[q].w}, uv[4] = {su[q].x, su[q].y, su[q].z, su[q].w},
                        nv[4] = {sn[q].x, sn[q].y, sn[q].z, sn[q].w}, gv[4] = {sg[q].x, sg[q].y, sg[q].z, sg[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * q + j;
                const float hp = hpv[r], d = dh[r] + dyv[r];
                const float rr = rv[j], uu = uv[j], nn = nv[j], gh = gv[j];
                const float dn = d * (1.0f - uu);
                const float du = d * (hp - nn);
                dhp[r] = d * uu;
                const float dan = dn * (1.0f - nn * nn);
                const float dau = du * uu * (1.0f - uu);
                const float dar = dan * gh * rr * (1.0f - rr);
                const float dgh = dan * rr;
                float* gw = gs_w + CR(r) * LDG;
                gw[0] = dar; gw[H] = dau; gw[2 * H] = dgh;
                if (full || CR(r) + lrow < nvalid) {
                    float* o = dgt + (int64_t)CR(r) * T * 4 * H;
                    o[lo_dg] = dar; o[lo_dg + H] = dau; o[lo_dg + 2 * H] = dan; o[lo_dg + 3 * H] = dgh;
                }
                dbs0 += dar; dbs1 += dau; dbs2 += dan; dbs3 += dgh;
            }
        }
        __syncthreads();
        if (step + 1 < T) load_step(step + 1);
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = dhp[r]; acc1[r] = 0.f; }
        float4 b0 = wpt[0], b1 = wpt[64];
#pragma unroll 2
        for (int c = 0; c < KC; c += 2) {
            const int cn = c + 2 < KC ? c + 2 : c;
            const float4 a0 = *reinterpret_cast<const float4*>(grow_a + 8 * c);
            const float4 a1 = *reinterpret_cast<const float4*>(grow_a + 8 * c + 8);
            const float4 b0n = wpt[cn * 64], b1n = wpt[(cn + 1) * 64];      // weights of the next chunk pair (L2) in flight
            acc0 = MFMA_32x32x2(a0.x, b0.x, acc0); acc1 = MFMA_32x32x2(a1.x, b1.x, acc1);
            acc0 = MFMA_32x32x2(a0.y, b0.y, acc0); acc1 = MFMA_32x32x2(a1.y, b1.y, acc1);
            acc0 = MFMA_32x32x2(a0.z, b0.z, acc0); acc1 = MFMA_32x32x2(a1.z, b1.z, acc1);
            acc0 = MFMA_32x32x2(a0.w, b0.w, acc0); acc1 = MFMA_32x32x2(a1.w, b1.w, acc1);
            b0 = b0n; b1 = b1n;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = acc0[r] + acc1[r];
        __syncthreads();
    }
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
}

// ------------------------------------------------------------------------------------------- host
template <int H>
static void launch_fwd(const GruFwdParams& P, hipStream_t st) {
    hipLaunchKernelGGL(gru_seq_fwd_kernel<H>, dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
}
template <int H>
static void launch_bwd(const GruBwdParams& P, hipStream_t st) {
    hipLaunchKernelGGL(gru_seq_bwd_kernel<H>, dim3(grid_blocks(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
}

extern "C" int vame_gru_seq_fwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream) {
    VAME_CHECK_ARG(desc && nstreams >= 1 && nstreams <= 8, VAME_E_BADARG, "gru_seq_fwd: nstreams=%d not in 1..8", nstreams);
    VAME_CHECK_ARG(B >= 1, VAME_E_SHAPE, "gru_seq_fwd: empty batch");
    GruFwdParams P;
    P.nstreams = nstreams; P.B = B; P.ntiles = (int)cdiv64(B, 32);
    for (int i = 0; i < nstreams; ++i) {
        const int64_t* d = desc + (int64_t)i * VAME_GRU_FWD_FIELDS;
        GruFwdStream& s = P.s[i];
        s.gi = (const float*)d[GF_GI]; s.gi_row = d[GF_GI_ROW]; s.gi_t = d[GF_GI_T];
        s.wp = (const float*)d[GF_WP]; s.bhn = (const float*)d[GF_BHN];
        s.h0 = (const float*)d[GF_H0]; s.h0_row = d[GF_H0_ROW];
        s.y = (float*)d[GF_Y]; s.y_row = d[GF_Y_ROW]; s.y_t = d[GF_Y_T];
        s.hn = (float*)d[GF_HN]; s.hn_row = d[GF_HN_ROW];
        s.stash = (float*)d[GF_STASH];
        s.T = d[GF_T]; s.reverse = d[GF_REVERSE]; s.pad = d[GF_PAD];
        VAME_CHECK_ARG(s.gi && s.wp && s.bhn, VAME_E_BADARG, "gru_seq_fwd: stream %d: gi/wp/bhn null", i);
        VAME_CHECK_ARG(s.T >= 1, VAME_E_SHAPE, "gru_seq_fwd: stream %d: T=%lld", i, (long long)s.T);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 32: launch_fwd<32>(P, st); break;
        case 64: launch_fwd<64>(P, st); break;
        case 128: launch_fwd<128>(P, st); break;
        case 256: launch_fwd<256>(P, st); break;
        default: VAME_CHECK_ARG(false, VAME_E_UNSUPPORTED, "gru_seq_fwd: H=%d unsupported (32,64,128,256)", H);
    }
    VAME_LAUNCH_CHECK("gru_seq_fwd");
    return VAME_OK;
}

extern "C" int vame_gru_seq_bwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream) {
    VAME_CHECK_ARG(desc && nstreams >= 1 && nstreams <= 8, VAME_E_BADARG, "gru_seq_bwd: nstreams=%d not in 1..8", nstreams);
    VAME_CHECK_ARG(B >= 1, VAME_E_SHAPE, "gru_seq_bwd: empty batch");
    GruBwdParams P;
    P.nstreams = nstreams; P.B = B; P.ntiles = (int)cdiv64(B, 32);
    for (int i = 0; i < nstreams; ++i) {
        const int64_t* d = desc + (int64_t)i * VAME_GRU_BWD_FIELDS;
        GruBwdStream& s = P.s[i];
        s.stash = (const float*)d[GB_STASH]; s.y = (const float*)d[GB_Y]; s.y_row = d[GB_Y_ROW]; s.y_t = d[GB_Y_T];
        s.h0 = (const float*)d[GB_H0]; s.h0_row = d[GB_H0_ROW];
        s.wpt = (const float*)d[GB_WPT];
        s.dy = (const float*)d[GB_DY]; s.dy_row = d[GB_DY_ROW]; s.dy_t = d[GB_DY_T];
        s.dhn = (const float*)d[GB_DHN]; s.dhn_row = d[GB_DHN_ROW];
        s.dg = (float*)d[GB_DG];
        s.dh0 = (float*)d[GB_DH0]; s.dh0_row = d[GB_DH0_ROW];
        s.dbias = (float*)d[GB_DBIAS];
        s.T = d[GB_T]; s.reverse = d[GB_REVERSE]; s.pad = d[GB_PAD];
        VAME_CHECK_ARG(s.stash && s.y && s.wpt && s.dg, VAME_E_BADARG, "gru_seq_bwd: stream %d: stash/y/wpt/dg null", i);
        VAME_CHECK_ARG(s.T >= 1, VAME_E_SHAPE, "gru_seq_bwd: stream %d: T=%lld", i, (long long)s.T);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 32: launch_bwd<32>(P, st); break;
        case 64: launch_bwd<64>(P, st); break;
        case 128: launch_bwd<128>(P, st); break;
        case 256: launch_bwd<256>(P, st); break;
        default: VAME_CHECK_ARG(false, VAME_E_UNSUPPORTED, "gru_seq_bwd: H=%d unsupported (32,64,128,256)", H);
    }
    VAME_LAUNCH_CHECK("gru_seq_bwd");
    return VAME_OK;
}
