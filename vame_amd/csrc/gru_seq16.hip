// 16-row-tile GRU sequence kernels (v_mfma_f32_16x16x4_f32).
//
// Same algorithm and stream descriptors as gru_seq.hip, but a workgroup owns 16 batch rows instead of 32 and needs
// half the registers, so TWO workgroups fit on a CU: while one is in its gate-math / barrier / store phase the other
// keeps the matrix pipe busy, and small batches spread over twice as many CUs (per-step latency halves).  The price is
// that W_hh is streamed from L2 once per 16 rows instead of once per 32.
//
// Fragment conventions (gfx950, 16x16x4 f32): lane l -> lj = l & 15, kq = l >> 4.
//   A[i = lj][k = kq], B[k = kq][j = lj], C/D: column lj, rows 4*kq + r (r = 0..3).
// A wave owns hidden columns [32w, 32w+32) as two INTERLEAVED 16-column tiles: tile ct holds columns 32w + 2*lj + ct,
// so a lane's two tiles are adjacent floats (8-byte loads / stores of gi, h, y).
#include "vame_common.h"

struct GruFwdStream16 {            // identical field order to GruFwdStream (gru_seq.hip); re-declared to keep the TU standalone
    const float* gi; int64_t gi_row, gi_t;
    const float* wp; const float* bhn;
    const float* h0; int64_t h0_row;
    float* y; int64_t y_row, y_t;
    float* hn; int64_t hn_row;
    float* stash;
    int64_t T, reverse, pad;
    const float* wpx; const float* bgi; int64_t xf;
};
struct GruFwdParams16 { GruFwdStream16 s[8]; int nstreams; int B; int ntiles; };

__device__ __forceinline__ bool map_block16(int nstreams, int ntiles, int& s, int& tile) {
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
static int grid_blocks16(int nstreams, int ntiles) {
    if (nstreams == 1) return (int)cdiv64(ntiles, 8) * 8;
    if (nstreams == 2) return (int)cdiv64(ntiles, 4) * 8;
    if (nstreams == 4) return 2 * (int)cdiv64(ntiles, 4) * 8;
    return nstreams * ntiles;
}

// wp16[((((w*(H/16) + c)*6 + (g*2 + ct))*64 + l)*4 + e] = W_hh[(g*H + 32w + 2*(l&15) + ct)*H + 16c + 4*(l>>4) + e]
__global__ __launch_bounds__(256) void gru_pack16_kernel(const float* __restrict__ W, int H, float* __restrict__ wp) {
    const int64_t n = (int64_t)3 * H * H;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = i & 3, l = (i >> 2) & 63;
        int64_t r = i >> 8;
        const int gc = r % 6; r /= 6;
        const int c = r % (H / 16), w = r / (H / 16);
        const int g = gc >> 1, ct = gc & 1;
        wp[i] = W[(int64_t)(g * H + 32 * w + 2 * (l & 15) + ct) * H + 16 * c + 4 * (l >> 4) + e];
    }
}

extern "C" int vame_gru_pack16_f32(const float* W_hh, int H, float* wp16, void* stream) {
    VAME_CHECK_ARG(H >= 32 && H % 32 == 0 && W_hh && wp16, VAME_E_BADARG, "gru_pack16: bad argument");
    const int64_t n = (int64_t)3 * H * H;
    hipLaunchKernelGGL(gru_pack16_kernel, dim3((unsigned)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024)), dim3(256), 0,
                       (hipStream_t)stream, W_hh, H, wp16);
    VAME_LAUNCH_CHECK("gru_pack16");
    return VAME_OK;
}

extern "C" int64_t vame_gru_stash16_floats(int B, int T, int H) { return cdiv64(B, 16) * 16 * (int64_t)T * 5 * H; }

// stash (fragment order): float4 index ((((tile*T + t)*NW + w)*5 + k)*2 + ct)*64 + lane, k = cA, cB, u, r, gh_n
template <int H>
__global__ __launch_bounds__(H / 32 * 64, 4) void gru_seq_fwd16_kernel(GruFwdParams16 P) {
    constexpr int NW = H / 32, LDH = H + 8, KC = H / 16, NT = NW * 64;
    __shared__ __attribute__((aligned(16))) float hs[2][16 * LDH];
    int sidx, tile;
    if (!map_block16(P.nstreams, P.ntiles, sidx, tile)) return;
    const GruFwdStream16& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 15, kq = lane >> 4;
#ifdef VAME_EMU
    const int w = tid >> 6;
#else
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int row0 = tile * 16, col0 = 32 * w + 2 * lj;          // this lane's two columns: col0, col0 + 1
    const int nvalid = B - row0;
    const bool full = nvalid >= 16;
    const int lrow = 4 * kq;                                       // lane rows: lrow + r
    float* y_tile = S.y ? S.y + (int64_t)row0 * S.y_row : nullptr;
    const int crow = tid / (H / 4), cc4 = tid % (H / 4);           // copy pass: (NT / (H/4)) = 8 rows per pass, 2 passes
    auto store_h = [&](const float* hbuf, int t) {
        float* yt = y_tile + (int64_t)t * S.y_t + (int64_t)crow * S.y_row + 4 * cc4;
        const float* src = hbuf + crow * LDH + 4 * cc4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (full || crow + 8 * i < nvalid)
                *reinterpret_cast<float4*>(yt + (int64_t)(8 * i) * S.y_row) = *reinterpret_cast<const float4*>(src + 8 * i * LDH);
    };
    float2 hprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lrow + r, grow = row0 + row;
        float2 v = make_float2(0.f, 0.f);
        if (S.h0 && grow < B) { const float* p = S.h0 + (int64_t)grow * S.h0_row + col0; v.x = p[0]; v.y = p[1]; }
        hprev[r] = v;
        *reinterpret_cast<float2*>(&hs[0][row * LDH + col0]) = v;
    }
    __syncthreads();
    if (y_tile && S.pad) store_h(hs[0], S.reverse ? T : -1);
    const float2 bhn = *reinterpret_cast<const float2*>(S.bhn + col0);
    const float4* __restrict__ wp = reinterpret_cast<const float4*>(S.wp) + (int64_t)w * KC * 6 * 64 + lane;
    float4* stash = S.stash ? reinterpret_cast<float4*>(S.stash) : nullptr;
    const float* gi_lane = S.gi + (int64_t)(row0 + lrow) * S.gi_row + col0;
    // gi of one step: rows lrow..lrow+3, gates r,u,n, the lane's column pair
    float2 g_r[4], g_u[4], g_n[4];
    auto load_gi = [&](int t) {
        const float* gt = gi_lane + (int64_t)t * S.gi_t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (full || lrow + r < nvalid) {
                const float* g = gt + (int64_t)r * S.gi_row;
                g_r[r] = *reinterpret_cast<const float2*>(g);
                g_u[r] = *reinterpret_cast<const float2*>(g + H);
                g_n[r] = *reinterpret_cast<const float2*>(g + 2 * H);
            } else { g_r[r] = g_u[r] = g_n[r] = make_float2(0.f, 0.f); }
        }
    };
    load_gi(S.reverse ? T - 1 : 0);
    constexpr int PD = 2;                       // ring depth in 16-k chunks (24 MFMAs each)
    float4 wq[PD][6];
#pragma unroll
    for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int q = 0; q < 6; ++q) wq[c][q] = wp[(c * 6 + q) * 64];
    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = S.reverse ? T - 1 - step : step;
        f32x4 ar[2], au[2], ani[2], anh[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ar[0][r] = g_r[r].x; ar[1][r] = g_r[r].y; au[0][r] = g_u[r].x; au[1][r] = g_u[r].y;
            ani[0][r] = g_n[r].x; ani[1][r] = g_n[r].y; anh[0][r] = bhn.x; anh[1][r] = bhn.y;
        }
        const float* hrow = &hs[cur][lj * LDH + 4 * kq];
#pragma unroll 1
        for (int c0 = 0; c0 < KC; c0 += PD) {
            if (c0 == KC / 2 / PD * PD && step + 1 < T && S.gi_t != 0) load_gi(S.reverse ? t - 1 : t + 1);
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const int c = c0 + j;
                const float4 a = *reinterpret_cast<const float4*>(hrow + 16 * c);
                float4 b[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) b[q] = wq[j][q];
                {
                    const int cn = (c0 + PD == KC) ? j : c + PD;            // wraps into the next step's first chunks
#pragma unroll
                    for (int q = 0; q < 6; ++q) wq[j][q] = wp[(cn * 6 + q) * 64];
                }
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bv0 = e == 0 ? b[0].x : e == 1 ? b[0].y : e == 2 ? b[0].z : b[0].w;
                    const float bv1 = e == 0 ? b[1].x : e == 1 ? b[1].y : e == 2 ? b[1].z : b[1].w;
                    const float bv2 = e == 0 ? b[2].x : e == 1 ? b[2].y : e == 2 ? b[2].z : b[2].w;
                    const float bv3 = e == 0 ? b[3].x : e == 1 ? b[3].y : e == 2 ? b[3].z : b[3].w;
                    const float bv4 = e == 0 ? b[4].x : e == 1 ? b[4].y : e == 2 ? b[4].z : b[4].w;
                    const float bv5 = e == 0 ? b[5].x : e == 1 ? b[5].y : e == 2 ? b[5].z : b[5].w;
                    ar[0] = MFMA_16x16x4(av[e], bv0, ar[0]); ar[1] = MFMA_16x16x4(av[e], bv1, ar[1]);
                    au[0] = MFMA_16x16x4(av[e], bv2, au[0]); au[1] = MFMA_16x16x4(av[e], bv3, au[1]);
                    anh[0] = MFMA_16x16x4(av[e], bv4, anh[0]); anh[1] = MFMA_16x16x4(av[e], bv5, anh[1]);
                }
            }
        }
        float* hnext = &hs[cur ^ 1][lrow * LDH + col0];
        float4 st[5][2];          // cA, cB, u, r, gh_n for the lane's 4 rows, per column tile
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rr = fast_sigmoid(ar[ct][r]);
                const float uu = fast_sigmoid(au[ct][r]);
                const float ghn = anh[ct][r];
                const float nn = fast_tanh(ani[ct][r] + rr * ghn);
                const float hp = ct ? hprev[r].y : hprev[r].x;
                const float hv = nn + uu * (hp - nn);
                const float omu = 1.0f - uu;
                ar[ct][r] = omu * (1.0f - nn * nn);       // cA
                au[ct][r] = (hp - nn) * uu * omu;         // cB
                ani[ct][r] = uu;
                anh[ct][r] = rr;
                (&st[4][ct].x)[r] = ghn;
                if (ct) hprev[r].y = hv; else hprev[r].x = hv;
            }
            st[0][ct] = make_float4(ar[ct][0], ar[ct][1], ar[ct][2], ar[ct][3]);
            st[1][ct] = make_float4(au[ct][0], au[ct][1], au[ct][2], au[ct][3]);
            st[2][ct] = make_float4(ani[ct][0], ani[ct][1], ani[ct][2], ani[ct][3]);
            st[3][ct] = make_float4(anh[ct][0], anh[ct][1], anh[ct][2], anh[ct][3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float2*>(hnext + r * LDH) = hprev[r];
        if (stash) {
            float4* sp = stash + ((((int64_t)tile * T + t) * NW + w) * 10) * 64 + lane;
#pragma unroll
            for (int k = 0; k < 5; ++k) { sp[(k * 2 + 0) * 64] = st[k][0]; sp[(k * 2 + 1) * 64] = st[k][1]; }
        }
        __syncthreads();
        cur ^= 1;
        if (y_tile) store_h(hs[cur], t);
    }
    if (S.hn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int grow = row0 + lrow + r;
            if (grow < B) *reinterpret_cast<float2*>(S.hn + (int64_t)grow * S.hn_row + col0) = hprev[r];
        }
    }
}

template <int H>
static void launch_fwd16(const GruFwdParams16& P, hipStream_t st) {
    hipLaunchKernelGGL(gru_seq_fwd16_kernel<H>, dim3(grid_blocks16(P.nstreams, P.ntiles)), dim3(H / 32 * 64), 0, st, P);
}

extern "C" int vame_gru_seq_fwd16_f32(const int64_t* desc, int nstreams, int B, int H, void* stream) {
    VAME_CHECK_ARG(desc && nstreams >= 1 && nstreams <= 8, VAME_E_BADARG, "gru_seq_fwd16: nstreams=%d not in 1..8", nstreams);
    VAME_CHECK_ARG(B >= 1, VAME_E_SHAPE, "gru_seq_fwd16: empty batch");
    GruFwdParams16 P;
    P.nstreams = nstreams; P.B = B; P.ntiles = (int)cdiv64(B, 16);
    for (int i = 0; i < nstreams; ++i) {
        const int64_t* d = desc + (int64_t)i * VAME_GRU_FWD_FIELDS;
        GruFwdStream16& s = P.s[i];
        s.gi = (const float*)d[GF_GI]; s.gi_row = d[GF_GI_ROW]; s.gi_t = d[GF_GI_T];
        s.wp = (const float*)d[GF_WP]; s.bhn = (const float*)d[GF_BHN];
        s.h0 = (const float*)d[GF_H0]; s.h0_row = d[GF_H0_ROW];
        s.y = (float*)d[GF_Y]; s.y_row = d[GF_Y_ROW]; s.y_t = d[GF_Y_T];
        s.hn = (float*)d[GF_HN]; s.hn_row = d[GF_HN_ROW];
        s.stash = (float*)d[GF_STASH];
        s.T = d[GF_T]; s.reverse = d[GF_REVERSE]; s.pad = d[GF_PAD];
        s.wpx = nullptr; s.bgi = nullptr; s.xf = 0;
        VAME_CHECK_ARG(s.gi && s.wp && s.bhn && d[GF_XF] == 0, VAME_E_BADARG, "gru_seq_fwd16: stream %d: gi/wp/bhn null or fused input", i);
        VAME_CHECK_ARG(s.T >= 1 && s.gi_row % 2 == 0 && s.gi_t % 2 == 0 && s.hn_row % 2 == 0 && s.h0_row % 2 == 0, VAME_E_SHAPE,
                       "gru_seq_fwd16: stream %d: strides must be even", i);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 32: launch_fwd16<32>(P, st); break;
        case 64: launch_fwd16<64>(P, st); break;
        case 128: launch_fwd16<128>(P, st); break;
        case 256: launch_fwd16<256>(P, st); break;
        default: VAME_CHECK_ARG(false, VAME_E_UNSUPPORTED, "gru_seq_fwd16: H=%d unsupported (32,64,128,256)", H);
    }
    VAME_LAUNCH_CHECK("gru_seq_fwd16");
    return VAME_OK;
}
