// Batch-tile-persistent GRU forward for WIDE hidden sizes, 256 < H <= 512 (BASELINE config 4: H = 512, T = 60, batch 8192).
//
// Same mapping as gru_seq.hip -- a workgroup owns 32 batch rows of one (layer, direction) stream for all T steps, h_t lives in LDS
// (double-buffered MFMA A operand), W_hh is streamed from L2 every step in the packed B-fragment order of vame_gru_pack_f32 -- but
// a wave owns TWO 32-column blocks of the hidden state (64 columns x 3 gates = 8 accumulator tiles of 32x32), so that H = 512 fits 8
// waves = 2 per SIMD with a 256-register budget, and every LDS A-fragment read feeds 24 MFMAs.  What changes against the H <= 256
// kernel to make room: the next step's gi tile is loaded DIRECTLY into the gate accumulators at the end of the epilogue (no 48-register
// prefetch copy), h_{t-1} of the wave's own columns is re-read from LDS for the blend instead of living in registers, and the
// weight-fragment ring is two chunks deep.  I/O contract = vame_gru_seq_fwd_f32 (descriptor table, padded sequence layout, BPTT stash in
// accumulator-fragment order with NB = H/32 column blocks) without the fused layer-0 input projection.
//
// Reference semantics: torch.nn.GRU as instantiated at vame/model/rnn_model.py:34-35,91-92,125-126.
#include "vame_common.h"
#include "gru_desc.h"

#ifdef VAME_EMU
#define WIDE_WAIT6(n, a, b, c, d, e, f)
#else
#define WIDE_WAIT6(n, a, b, c, d, e, f) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(n))
#endif

// blockIdx -> (stream, tile): streams dealt to XCD parity classes like gru_seq.hip's map_block (one XCD's L2 keeps at most two streams' W_hh)
__device__ __forceinline__ bool wide_map_block(int nstreams, int ntiles, int& s, int& tile) {
    const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
    if (nstreams == 1) { s = 0; tile = q * 8 + xcd; }
    else if (nstreams == 2) { s = xcd & 1; tile = q * 4 + (xcd >> 1); }
    else { s = bid % nstreams; tile = bid / nstreams; }
    return tile < ntiles;
}
static int wide_grid_blocks(int nstreams, int ntiles) {
    if (nstreams == 1) return (int)cdiv64(ntiles, 8) * 8;
    if (nstreams == 2) return (int)cdiv64(ntiles, 4) * 8;
    return nstreams * ntiles;
}

template <int H>
__global__ __launch_bounds__(H / 64 * 64) void gru_wide_fwd_kernel(GruFwdParams P) {
    constexpr int NW = H / 64, NB = H / 32, NT = NW * 64, LDH = H + 4, KC = H / 8, PD = 2;
    static_assert(H % 64 == 0 && H > 256 && H <= 512 && KC % PD == 0, "wide kernel: H in {320, 384, 448, 512}");
    __shared__ float hs[2][32 * LDH];
    int sidx, tile;
    if (!wide_map_block(P.nstreams, P.ntiles, sidx, tile)) return;
    const GruFwdStream& S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    const int row0 = tile * 32, col0 = 64 * w, lrow = 4 * hh;
    const int nvalid = B - row0;
    const bool full = nvalid >= 32;
    const int lo_gi = lrow * (int)S.gi_row + li;
    const float* gi_base = S.gi + (int64_t)row0 * S.gi_row + col0;
    float* y_tile = S.y ? S.y + (int64_t)row0 * S.y_row : nullptr;
    // h_t leaves through LDS as 16-byte row stores: NT threads cover (NT / (H/4)) rows x (H/4) float4 per pass
    constexpr int RPP = NT / (H / 4), NPASS = 32 / RPP;
    static_assert(NT % (H / 4) == 0 && 32 % RPP == 0, "copy pass geometry");
    const int crow = tid / (H / 4), cc4 = tid % (H / 4);
    auto store_h = [&](const float* hbuf, int t) {
        float* yt = y_tile + (int64_t)t * S.y_t + (int64_t)crow * S.y_row + 4 * cc4;
        const float* src = hbuf + crow * LDH + 4 * cc4;
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            if (full || crow + RPP * i < nvalid)
                *reinterpret_cast<float4*>(yt + (int64_t)(RPP * i) * S.y_row) = *reinterpret_cast<const float4*>(src + RPP * i * LDH);
    };
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = CR(r) + lrow, grow = row0 + row;
            float v = 0.0f;
            if (S.h0 && grow < B) v = S.h0[(int64_t)grow * S.h0_row + col0 + 32 * cb + li];
            hs[0][row * LDH + col0 + 32 * cb + li] = v;
        }
    __syncthreads();
    if (y_tile && S.pad) store_h(hs[0], S.reverse ? T : -1);
    float bhn[2];
    bhn[0] = S.bhn[col0 + li]; bhn[1] = S.bhn[col0 + 32 + li];
    const float4* __restrict__ wp0 = reinterpret_cast<const float4*>(S.wp) + (int64_t)(2 * w) * KC * 3 * 64 + lane;
    const float4* __restrict__ wp1 = wp0 + (int64_t)KC * 3 * 64;
    float4* stash = S.stash ? reinterpret_cast<float4*>(S.stash) : nullptr;
    f32x16 ar[2], au[2], ani[2], anh[2];
    // gi of step t straight into the gate accumulators (the MFMAs of the step accumulate on top)
    auto load_gi = [&](int t, int cb) {
        const float* gt = gi_base + (int64_t)t * S.gi_t + 32 * cb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* g = gt + (int64_t)CR(r) * S.gi_row;
            if (full || CR(r) + lrow < nvalid) { ar[cb][r] = g[lo_gi]; au[cb][r] = g[lo_gi + H]; ani[cb][r] = g[lo_gi + 2 * H]; }
            else { ar[cb][r] = 0.f; au[cb][r] = 0.f; ani[cb][r] = 0.f; }
        }
    };
    load_gi(S.reverse ? T - 1 : 0, 0);
    load_gi(S.reverse ? T - 1 : 0, 1);
    f32x4 wq[PD][2][3];
#pragma unroll
    for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int g = 0; g < 3; ++g) { RING_LOAD(wq[c][0][g], wp0, (c * 3 + g) * 64); RING_LOAD(wq[c][1][g], wp1, (c * 3 + g) * 64); }
    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = S.reverse ? T - 1 - step : step;
#pragma unroll
        for (int r = 0; r < 16; ++r) { anh[0][r] = bhn[0]; anh[1][r] = bhn[1]; }
        const float* hrow = &hs[cur][li * LDH + 4 * hh];
        if (!(step == 0 && S.h0 == nullptr)) {          // zero initial state: h W_hh^T contributes nothing to the first step
#pragma unroll 1
            for (int c0 = 0; c0 < KC; c0 += PD)
#pragma unroll
                for (int j = 0; j < PD; ++j) {
                    const int c = c0 + j;
                    const float4 a = *reinterpret_cast<const float4*>(hrow + 8 * c);
                    WIDE_WAIT6(6 * (PD - 1), wq[j][0][0], wq[j][0][1], wq[j][0][2], wq[j][1][0], wq[j][1][1], wq[j][1][2]);
                    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ar[0] = MFMA_32x32x2(av[e], wq[j][0][0][e], ar[0]); ar[1] = MFMA_32x32x2(av[e], wq[j][1][0][e], ar[1]);
                        au[0] = MFMA_32x32x2(av[e], wq[j][0][1][e], au[0]); au[1] = MFMA_32x32x2(av[e], wq[j][1][1][e], au[1]);
                        anh[0] = MFMA_32x32x2(av[e], wq[j][0][2][e], anh[0]); anh[1] = MFMA_32x32x2(av[e], wq[j][1][2][e], anh[1]);
                    }
                    RING_FENCE();
                    // refill after the slot's last use (lands in the same registers); the last group wraps into the next step's chunks
                    const int cn = (c0 + PD == KC) ? j : c + PD;
#pragma unroll
                    for (int g = 0; g < 3; ++g) { RING_LOAD(wq[j][0][g], wp0, (cn * 3 + g) * 64); RING_LOAD(wq[j][1][g], wp1, (cn * 3 + g) * 64); }
                }
        }
        // epilogue, one column block at a time: gates, h_t -> LDS, BPTT coefficients -> stash, then the block's accumulators are re-loaded
        // with the next step's gi
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float* hold = &hs[cur][lrow * LDH + col0 + 32 * cb + li];
            float* hnext = &hs[cur ^ 1][lrow * LDH + col0 + 32 * cb + li];
            float4* sp = stash ? stash + ((((int64_t)tile * T + t) * NB + 2 * w + cb) * 20) * 64 + lane : nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float ust[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 4 * q + jj;
                    const float rr = fast_sigmoid(ar[cb][r]);
                    const float uu = fast_sigmoid(au[cb][r]);
                    const float nn = fast_tanh(ani[cb][r] + rr * anh[cb][r]);
                    const float hp = hold[CR(r) * LDH];
                    const float hv = nn + uu * (hp - nn);
                    const float omu = 1.0f - uu;
                    ani[cb][r] = omu * (1.0f - nn * nn);          // cA
                    au[cb][r] = (hp - nn) * uu * omu;             // cB
                    ar[cb][r] = rr;
                    ust[jj] = uu;
                    hnext[CR(r) * LDH] = hv;
                }
                if (stash) {
                    sp[(0 * 4 + q) * 64] = make_float4(ani[cb][4 * q], ani[cb][4 * q + 1], ani[cb][4 * q + 2], ani[cb][4 * q + 3]);
                    sp[(1 * 4 + q) * 64] = make_float4(au[cb][4 * q], au[cb][4 * q + 1], au[cb][4 * q + 2], au[cb][4 * q + 3]);
                    sp[(2 * 4 + q) * 64] = make_float4(ust[0], ust[1], ust[2], ust[3]);
                    sp[(3 * 4 + q) * 64] = make_float4(ar[cb][4 * q], ar[cb][4 * q + 1], ar[cb][4 * q + 2], ar[cb][4 * q + 3]);
                    sp[(4 * 4 + q) * 64] = make_float4(anh[cb][4 * q], anh[cb][4 * q + 1], anh[cb][4 * q + 2], anh[cb][4 * q + 3]);
                }
                SCHED_FENCE();
            }
            if (step + 1 < T) load_gi(S.reverse ? t - 1 : t + 1, cb);
        }
        __syncthreads();
        cur ^= 1;
        if (y_tile) store_h(hs[cur], t);
    }
    if (S.hn) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = CR(r) + lrow, grow = row0 + row;
                if (grow < B) S.hn[(int64_t)grow * S.hn_row + col0 + 32 * cb + li] = hs[cur][row * LDH + col0 + 32 * cb + li];
            }
    }
}

extern "C" int vame_gru_wide_supported(int H) { return H > 256 && H <= 512 && H % 64 == 0; }

extern "C" int vame_gru_wide_fwd_f32(const int64_t* desc, int nstreams, int B, int H, void* stream) {
    VAME_CHECK_ARG(desc && nstreams >= 1 && nstreams <= 8, VAME_E_BADARG, "gru_wide_fwd: nstreams=%d not in 1..8", nstreams);
    VAME_CHECK_ARG(B >= 1, VAME_E_SHAPE, "gru_wide_fwd: empty batch");
    GruFwdParams P;
    if (int rc = gru_parse_fwd(desc, nstreams, B, P)) return rc;
    for (int i = 0; i < nstreams; ++i)
        VAME_CHECK_ARG(P.s[i].xf == 0, VAME_E_UNSUPPORTED, "gru_wide_fwd: stream %d: the fused input projection is an H <= 256 feature", i);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(wide_grid_blocks(nstreams, P.ntiles));
    switch (H) {
        case 320: hipLaunchKernelGGL((gru_wide_fwd_kernel<320>), grid, dim3(320), 0, st, P); break;
        case 384: hipLaunchKernelGGL((gru_wide_fwd_kernel<384>), grid, dim3(384), 0, st, P); break;
        case 448: hipLaunchKernelGGL((gru_wide_fwd_kernel<448>), grid, dim3(448), 0, st, P); break;
        case 512: hipLaunchKernelGGL((gru_wide_fwd_kernel<512>), grid, dim3(512), 0, st, P); break;
        default: VAME_CHECK_ARG(false, VAME_E_UNSUPPORTED, "gru_wide_fwd: H=%d unsupported (320, 384, 448, 512)", H);
    }
    VAME_LAUNCH_CHECK("gru_wide_fwd");
    return VAME_OK;
}
