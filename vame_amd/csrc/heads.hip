// Output head of a decoder in the TRAINING step as ONE pass over the decoder's states: hidden_to_output Linear -> MSE(sum) -> its gradient ->
// back through the Linear to the state sequence, AND the Linear's weight gradient.
//
//   reference: prediction = hidden_to_output(decoder_states)                      vame/model/rnn_model.py:107-108, 139-140
//              rec_loss = mse_loss(x_tilde, x, reduction)  (and the future one)   vame/model/rnn_vae.py:35-43, 124-125
//              loss.backward(): d loss / d prediction, d prediction / d states, d loss / d hidden_to_output.weight   rnn_vae.py:141-143
//
// As separate launches this is a (B*T x F x K) GEMM, the MSE kernel, a (B*T x K x F) GEMM and a split-K (F x K x B*T) GEMM + its reduction
// (F = 24, K = 512, B*T = 122,880 at the headline shape): the 252 MB of states are read twice and the 252 MB of state gradients written once,
// each launch at 2.4-3.5 TB/s because a tile of any of them has almost no arithmetic to hide its memory time behind (profiles/r05_narrow_gemms.txt).
// Here a workgroup streams 16-row tiles of the states through LDS ONCE:
//   P1  pred partials = tile x W^T        16x16x4 f32 MFMAs, K split over the four waves (W^T fragments re-read from L2 per tile)
//   --  pred = bias + partials (fixed order), error, loss, dpred -> global + a zero-padded LDS tile
//   P3  dW += dpred^T x tile              M = F (two 16-row MFMA tiles), N = the wave's K/4 state columns, K = the 16 rows; accumulators live
//                                         in registers across all tiles of the workgroup
//   P2  dY tile = dpred x W               written over the state tile in LDS, then copied out as full 128-byte row segments (16 B per lane)
// while the next tile's global loads (issued behind P1) are in flight; two workgroups per CU overlap each other's phases.  The per-workgroup dW
// sums go to a workspace and a second launch adds them in a fixed order (deterministic; no float atomics except the scalar loss sum, as in mse_kernel).
// Both f32 MFMA shapes, plain loads and stores only: the host emulator runs this file as is.
#include "vame_common.h"
#include "gru_desc.h"

#if defined(VAME_PROBE) && !defined(VAME_EMU)   // tuning build (make probe): per-wave phase cycle sums, tools/head_probe.py
__device__ long long* g_head_probe;
extern "C" int vame_probe_set_head(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_head_probe), &p, sizeof(p)); }
#define HS_PHASE_DECL() long long pp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pa_ = (long long)__builtin_amdgcn_s_memtime()
#define HS_PHASE(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); pp_[i] += t_ - pa_; pa_ = t_; } while (0)
#define HS_PHASE_END()                                                                          \
    if ((threadIdx.x & 63) == 0 && g_head_probe) {                                               \
        long long* o_ = g_head_probe + ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;     \
        for (int i_ = 0; i_ < 8; ++i_) o_[i_] = pp_[i_];                                         \
    }
#else
#define HS_PHASE_DECL()
#define HS_PHASE(i)
#define HS_PHASE_END()
#endif

namespace {
constexpr int HS_ROWS = 16;
constexpr int HS_DLD = 48;                 // dpred tile row stride: rows 4s + q land 16 q banks apart (conflict-free A^T reads in P3)

struct HeadParams {
    const float* Y; int64_t y_ld, y_seg, y_seg_stride;      // row m = (b,t), b = m / y_seg: Y + b*y_seg_stride + t*y_ld, K floats
    const float* W; const float* bias;                      // (F, K) row-major, (F)
    const float* tgt; int64_t tgt_row, tgt_off;             // target of row (b,t), feature f: tgt[b*tgt_row + tgt_off + t*F + f]
    float* pred; float* dpred;                              // (M, F) each; pred may be null
    float* dY; int64_t dy_ld;                               // (M, dy_ld), columns [0, K) written
    float* loss;                                            // loss[0] += sum of squared errors
    float* ws;                                              // (gridDim.x, 2, NW, NT, 64) float4: per-workgroup dW sums in accumulator order
    int M, F, K, ntiles;
    float gscale;
};

// NW waves per workgroup (4: K <= 512, two workgroups per CU; 8: K <= 1024 -- hidden sizes up to 512 --, one workgroup per CU); NT = K / (16 NW):
// 16-column MFMA tiles per wave (a wave owns K / NW state columns in P2 / P3 and K / NW of the contraction in P1); NFS = ceil(F / 4) k-steps of P2
// (6 covers F <= 24, 8 covers F <= 32)
template <int NT, int NFS, int NW>
__global__ __launch_bounds__(64 * NW, 2) void head_stream_kernel(HeadParams P) {
    VAME_DYN_SMEM(smem_raw);
    constexpr int K = 16 * NT * NW, LD = K + 16, LPR = 4 * NW; // LD = 16 (mod 32): rows 4s + q of a column sit 16 q banks apart; LPR loader lanes per row
    float* yt = reinterpret_cast<float*>(smem_raw);            // [16][LD]   state tile, later the dY tile
    float* part = yt + HS_ROWS * LD;                           // [NW][16][32] P1 partial sums per wave
    float* db = part + NW * HS_ROWS * 32;                      // [16][HS_DLD] dpred tile (0 for f >= F and rows >= M)
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, q = lane >> 4;
    const int w = UNIFORM(tid >> 6);
    const int F = P.F, seg = (int)P.y_seg;
    const int kw = w * (K / NW);                               // P1: this wave's k range; P2 / P3: this wave's state columns
    // ---- W fragments.  P1 (B operand of tile x W^T: column n = 16 nt + l16, k = kw + 16 c + 4 q + e) is 8 NT registers per lane: re-read from
    // L2 for every tile (requested at the top of the iteration, consumed by P1) -- all of {P1 fragments, P2 fragments, dW accumulators, next tile
    // in flight} resident at once spills, and a spilled register costs a scratch round trip in the middle of a phase.  Rows n >= F of the padded
    // 32-column prediction tile read row F - 1; their partial sums are zeroed by the mask.
    int w1i[2];                                                // (32-bit element indices into W: one scalar base + a register per row)
    float w1m[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = 16 * nt + l16;
        w1i[nt] = (n < F ? n : F - 1) * K + kw + 4 * q;
        w1m[nt] = n < F ? 1.f : 0.f;
    }
    // P2 (B operand of dpred x W): k = f = 4 s + q, column kw + 16 nt + l16; resident (rows f >= F: 0)
    float w2[NT][NFS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < NFS; ++s) {
            const int f = 4 * s + q;
            w2[nt][s] = P.W[(f < F ? f : F - 1) * K + kw + 16 * nt + l16] * (f < F ? 1.f : 0.f);
        }
    f32x4 acc3[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc3[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // loader / copy-out / dpred role of a thread: row tid / LPR, 16-byte column units c0 + LPR j; features c0 (and c0 + 16 when a row has 16 lanes)
    const int lrow = tid / LPR, c0 = tid % LPR;
    constexpr int NFT = 32 / LPR;                              // prediction features per thread (2 or 1)
    float bF[NFT];
#pragma unroll
    for (int h = 0; h < NFT; ++h) bF[h] = c0 + LPR * h < F ? P.bias[c0 + LPR * h] : 0.f;
    float lsum = 0.f;
    f32x4 nxt[NT];
    // (b, t) of this thread's row in the tile being fetched, advanced by one grid stride per tile without divisions: every VALU instruction
    // here waits behind the f32 MFMAs of the CU's other workgroup
    const int step_rows = (int)gridDim.x * HS_ROWS, step_b = step_rows / seg, step_t = step_rows - step_b * seg;
    int fb = (blockIdx.x * HS_ROWS + lrow) / seg, ft = (blockIdx.x * HS_ROWS + lrow) - fb * seg, cb = 0, ct = 0;
    auto fetch = [&](int tile) {
        const int m = tile * HS_ROWS + lrow;
        const bool ok = tile < P.ntiles && m < P.M;
        const float* yrow = P.Y + (int64_t)(ok ? fb : 0) * P.y_seg_stride + (int64_t)(ok ? ft : 0) * P.y_ld + 4 * c0;
#pragma unroll
        for (int j = 0; j < NT; ++j) nxt[j] = ok ? *reinterpret_cast<const f32x4*>(yrow + 4 * LPR * j) : f32x4{0.f, 0.f, 0.f, 0.f};
        cb = fb; ct = ft;                                             // what the tile just requested will be when it is the current one
        fb += step_b; ft += step_t;
        if (ft >= seg) { ft -= seg; ++fb; }
    };
    int tile = blockIdx.x;
    fetch(tile);
    int tb = cb, tt = ct;                                              // (b, t) of the CURRENT tile's row
    HS_PHASE_DECL();
    for (; tile < P.ntiles; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(&yt[lrow * LD + 4 * c0 + 4 * LPR * j]) = nxt[j];
#ifndef VAME_EMU
        asm volatile("" ::: "memory");      // (the tile's registers are free before P1's fragments are requested: both live at once would spill)
#endif
        f32x4 w1[NT][2];
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) w1[c][nt] = *reinterpret_cast<const f32x4*>(P.W + (w1i[nt] + 16 * c));
        // this tile's targets: requested before P1, used behind it (thread = (row lrow, features c0 and c0 + 16))
        const int m_t = tile * HS_ROWS + lrow;
        const bool ok_t = m_t < P.M;
        float tgv[NFT];
        {
            const float* tg = P.tgt + (int64_t)(ok_t ? tb : 0) * P.tgt_row + P.tgt_off + (int64_t)(ok_t ? tt : 0) * F;
#pragma unroll
            for (int h = 0; h < NFT; ++h) tgv[h] = (ok_t && c0 + LPR * h < F) ? tg[c0 + LPR * h] : 0.f;
        }
        __syncthreads();
        HS_PHASE(0);                                                   // tile -> LDS (waits for the prefetch), requests, barrier
        // ---- P1: partial prediction of the wave's quarter of K.  A: row l16, k = kw + 16 c + 4 q + e
        {
            f32x4 a1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&yt[l16 * LD + kw + 16 * c + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a1[0] = MFMA_16x16x4(a[e], w1[c][0][e], a1[0]);
                    a1[1] = MFMA_16x16x4(a[e], w1[c][1][e], a1[1]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(w * HS_ROWS + 4 * q + r) * 32 + 16 * nt + l16] = a1[nt][r] * w1m[nt];      // (columns >= F: 0)
        }
        __syncthreads();
        HS_PHASE(1);                                                   // P1 + barrier
        // the next tile's states: the YOUNGEST loads of the iteration (vmcnt retires in order: the wait for the targets leaves these in flight),
        // covered by the dpred phase, P3, P2 and the copy-out; not in front of P1: its fragments + these would spill
        fetch(tile + gridDim.x);
        const int nb_ = cb, nt_ = ct;                                  // (the next iteration's current row position)
        // ---- prediction, error, loss, dpred
        {
#pragma unroll
            for (int h = 0; h < NFT; ++h) {
                const int f = c0 + LPR * h;
                const float* pp = part + lrow * 32 + f;
                float p = (pp[0] + pp[HS_ROWS * 32]) + (pp[2 * HS_ROWS * 32] + pp[3 * HS_ROWS * 32]);
                if (NW == 8) p += (pp[4 * HS_ROWS * 32] + pp[5 * HS_ROWS * 32]) + (pp[6 * HS_ROWS * 32] + pp[7 * HS_ROWS * 32]);
                p += bF[h];
                float g = 0.f;
                if (ok_t && f < F) {
                    const float e = p - tgv[h];
                    lsum += e * e;
                    g = P.gscale * e;
                    if (P.pred) P.pred[(int64_t)m_t * F + f] = p;
                    P.dpred[(int64_t)m_t * F + f] = g;
                }
                db[lrow * HS_DLD + f] = g;
            }
        }
        __syncthreads();
        HS_PHASE(2);                                                   // dpred phase + barrier
        // ---- P3: dW (F x the wave's K/4 columns) += dpred^T (A: row f = 16 mt + l16, k = tile row 4 s + q) x tile (B: column kw + 16 nt + l16)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a0 = db[(4 * s + q) * HS_DLD + l16], a1v = db[(4 * s + q) * HS_DLD + 16 + l16];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float y = yt[(4 * s + q) * LD + kw + 16 * nt + l16];
                acc3[0][nt] = MFMA_16x16x4(a0, y, acc3[0][nt]);
                acc3[1][nt] = MFMA_16x16x4(a1v, y, acc3[1][nt]);
            }
        }
        __syncthreads();                                               // every wave is done with the state tile
        HS_PHASE(3);                                                   // P3 + barrier
        // ---- P2: gradient tile = dpred (A: row l16, k = f = 4 s + q) x W, written over the state tile
        {
            float a2[NFS];
#pragma unroll
            for (int s = 0; s < NFS; ++s) a2[s] = db[l16 * HS_DLD + 4 * s + q];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NFS; ++s) o = MFMA_16x16x4(a2[s], w2[nt][s], o);
#pragma unroll
                for (int r = 0; r < 4; ++r) yt[(4 * q + r) * LD + kw + 16 * nt + l16] = o[r];
            }
        }
        __syncthreads();
        HS_PHASE(4);                                                   // P2 + barrier
        // ---- copy-out: full row segments, 16 bytes per lane (the same thread refills these words with the next tile: no barrier in between)
        {
            const int m = tile * HS_ROWS + lrow;
            if (m < P.M) {
                float* drow = P.dY + (int64_t)m * P.dy_ld + 4 * c0;
#pragma unroll
                for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(drow + 4 * LPR * j) = *reinterpret_cast<const f32x4*>(&yt[lrow * LD + 4 * c0 + 4 * LPR * j]);
            }
        }
        HS_PHASE(5);                                                   // copy-out
        tb = nb_; tt = nt_;
    }
    HS_PHASE_END();
    // ---- this workgroup's dW sums -> workspace in ACCUMULATOR order: [mt][wave][nt][lane] float4 units (one coalesced 1 KB store per
    // accumulator tile); head_dw_reduce_kernel maps unit (mt, wave, nt, lane = 16 q + l16), element r to dW[16 mt + 4 q + r][wave K/4 + 16 nt + l16]
    f32x4* wsb = reinterpret_cast<f32x4*>(P.ws) + (int64_t)blockIdx.x * (2 * NW * NT * 64);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wsb[((mt * NW + w) * NT + nt) * 64 + lane] = acc3[mt][nt];
    // one float atomic per workgroup (atomics on one address serialise)
    __shared__ float red[8];
    lsum = wave_sum(lsum);
    if (lane == 0) red[w] = lsum;
    __syncthreads();
    if (tid == 0) atomicAdd(P.loss, NW == 8 ? ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])) : (red[0] + red[1]) + (red[2] + red[3]));
}

// dW = sum over the workgroups' partial sums, read in the accumulator order they were stored in.  A block = 64 consecutive float4 units x 16
// segments of the workgroup range (1024 threads; a unit's 16 segment sums meet in LDS and are added as a fixed tree), so 0.4 M loads are in
// flight and the order of the additions depends on nothing but nwg.
__global__ __launch_bounds__(1024) void head_dw_reduce_kernel(const f32x4* ws, int nwg, int NT, int NW, int F, int K, float* dW) {
    __shared__ f32x4 seg[16][64];
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6, units = 2 * NW * NT * 64;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    // unit -> (mt, wave, nt, lane); rows 16 mt + 4 q + r >= F are padding: never read
    const int lane = u & 63, nt = (u >> 6) % NT, wv = (u >> 6) / NT % NW, mt = (u >> 6) / NT / NW, q = lane >> 4, l16 = lane & 15;
    const int f0 = 16 * mt + 4 * q;
    const bool live = u < units && f0 < F;
    if (live) {
        const int per = (nwg + 15) / 16, g0 = sg * per, g1 = g0 + per < nwg ? g0 + per : nwg;
        int g = g0;
        for (; g + 2 <= g1; g += 2) { a += ws[(int64_t)g * units + u]; b += ws[(int64_t)(g + 1) * units + u]; }
        if (g < g1) a += ws[(int64_t)g * units + u];
    }
    seg[sg][threadIdx.x & 63] = a + b;
    __syncthreads();
    if (sg == 0 && live) {
        f32x4 t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = seg[i][threadIdx.x & 63];
#pragma unroll
        for (int st = 1; st < 16; st *= 2)
#pragma unroll
            for (int i = 0; i < 16; i += 2 * st) t[i] += t[i + st];
        const int col = wv * (K / NW) + 16 * nt + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (f0 + r < F) dW[(int64_t)(f0 + r) * K + col] = t[0][r];
    }
}

// 4 waves per workgroup for K a multiple of 64 up to 512, 8 waves for multiples of 128 up to 1024
int head_nw(int K) { return (K <= 512 && K % 64 == 0) ? 4 : 8; }
bool head_covers(int M, int F, int K) { return M >= 1 && F >= 1 && F <= 32 && K >= 64 && ((K <= 512 && K % 64 == 0) || (K <= 1024 && K % 128 == 0)); }
int head_nwg(int M, int K) {
    const int tiles = (int)cdiv64(M, HS_ROWS), cap = head_nw(K) == 4 ? 512 : 256;     // two (one) workgroups per CU
    return tiles < cap ? tiles : cap;
}

template <int NT, int NFS, int NW>
int head_launch(const HeadParams& P, int nwg, hipStream_t st) {
    const size_t lds = ((size_t)HS_ROWS * (16 * NT * NW + 16) + NW * HS_ROWS * 32 + HS_ROWS * HS_DLD) * sizeof(float);
#ifndef VAME_EMU
    VAME_CHECK_ARG(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_stream_kernel<NT, NFS, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                   VAME_E_HIP, "head_stream: cannot reserve %d bytes of LDS", (int)lds);
#endif
    hipLaunchKernelGGL((head_stream_kernel<NT, NFS, NW>), dim3((unsigned)nwg), dim3(64 * NW), lds, st, P);
    return VAME_OK;
}

template <int NFS>
int head_dispatch(const HeadParams& P, int nwg, hipStream_t st) {
    if (head_nw(P.K) == 4) {
        switch (P.K / 64) {
            case 1: return head_launch<1, NFS, 4>(P, nwg, st);
            case 2: return head_launch<2, NFS, 4>(P, nwg, st);
            case 3: return head_launch<3, NFS, 4>(P, nwg, st);
            case 4: return head_launch<4, NFS, 4>(P, nwg, st);
            case 5: return head_launch<5, NFS, 4>(P, nwg, st);
            case 6: return head_launch<6, NFS, 4>(P, nwg, st);
            case 7: return head_launch<7, NFS, 4>(P, nwg, st);
            default: return head_launch<8, NFS, 4>(P, nwg, st);
        }
    }
    switch (P.K / 128) {
        case 5: return head_launch<5, NFS, 8>(P, nwg, st);
        case 6: return head_launch<6, NFS, 8>(P, nwg, st);
        case 7: return head_launch<7, NFS, 8>(P, nwg, st);
        default: return head_launch<8, NFS, 8>(P, nwg, st);
    }
}
}  // namespace

extern "C" int64_t vame_head_stream_ws_floats(int M, int F, int K) {
    if (!head_covers(M, F, K)) return -1;                                           // -1: shape not covered (callers keep the separate launches)
    return (int64_t)head_nwg(M, K) * (2 * (K / 16) * 64 * 4);                     // accumulator-order units (2 x NW x NT x 64 float4) incl. the padding rows f >= F
}

extern "C" int vame_head_stream_f32(const float* Y, int64_t y_ld, int64_t y_seg, int64_t y_seg_stride, int M, int F, int K, const float* W,
                                    const float* bias, const float* tgt, int64_t tgt_row, int64_t tgt_off, float gscale, float* pred,
                                    float* dpred, float* dY, int64_t dy_ld, float* loss, float* dW, float* ws, void* stream) {
    VAME_CHECK_ARG(Y && W && bias && tgt && dpred && dY && loss && dW && ws, VAME_E_BADARG, "head_stream: null pointer");
    VAME_CHECK_ARG(vame_head_stream_ws_floats(M, F, K) > 0 && y_seg >= 1, VAME_E_SHAPE,
                   "head_stream: M=%d F=%d (1..32) K=%d (a multiple of 64 up to 512 or of 128 up to 1024)", M, F, K);
    VAME_CHECK_ARG(y_ld % 4 == 0 && y_seg_stride % 4 == 0 && (uintptr_t)Y % 16 == 0 && (uintptr_t)W % 16 == 0 && dy_ld >= K && dy_ld % 4 == 0 &&
                       (uintptr_t)dY % 16 == 0, VAME_E_SHAPE,
                   "head_stream: state rows, W and dY rows must be 16-byte aligned (y_ld=%lld, y_seg_stride=%lld, dy_ld=%lld)", (long long)y_ld,
                   (long long)y_seg_stride, (long long)dy_ld);
    HeadParams P;
    P.Y = Y; P.y_ld = y_ld; P.y_seg = y_seg; P.y_seg_stride = y_seg_stride; P.W = W; P.bias = bias; P.tgt = tgt; P.tgt_row = tgt_row;
    P.tgt_off = tgt_off; P.pred = pred; P.dpred = dpred; P.dY = dY; P.dy_ld = dy_ld; P.loss = loss; P.ws = ws; P.M = M; P.F = F; P.K = K;
    P.ntiles = (int)cdiv64(M, HS_ROWS); P.gscale = gscale;
    const int nwg = head_nwg(M, K), NWv = head_nw(K);
    const int rc = F <= 24 ? head_dispatch<6>(P, nwg, (hipStream_t)stream) : head_dispatch<8>(P, nwg, (hipStream_t)stream);
    if (rc != VAME_OK) return rc;
    VAME_LAUNCH_CHECK("head_stream");
    hipLaunchKernelGGL(head_dw_reduce_kernel, dim3((unsigned)(2 * (K / 16))), dim3(1024), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(ws), nwg, K / 16 / NWv, NWv, F, K, dW);
    VAME_LAUNCH_CHECK("head_dw_reduce");
    return VAME_OK;
}
