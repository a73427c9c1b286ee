// Output head of a decoder in the TRAINING step, one kernel: hidden_to_output Linear -> MSE(sum) -> its gradient -> back through the
// Linear to the decoder's output sequence.
//
//   reference: prediction = hidden_to_output(decoder_states)                      vame/model/rnn_model.py:107-108, 139-140
//              rec_loss = mse_loss(x_tilde, x, reduction)  (and the future one)   vame/model/rnn_vae.py:35-43, 124-125
//              their backward: d loss / d prediction, d prediction / d states
//
// As separate launches this was a (B*T x 24 x 512) GEMM, the MSE kernel and a (B*T x 512 x 24) GEMM: the first reads the 252 MB of
// decoder states, the last writes 252 MB of state gradients, each at 2.6-3.3 TB/s inside the step because a tile of either has almost
// no arithmetic to hide its memory time behind.  Here a wave owns 32 rows (b,t): it contracts their states with W (F x K, resident in
// LDS) into a 32 x 32 accumulator tile = prediction, forms error / loss / dpred in registers, turns the tile from the MFMA C layout into
// the A layout through a 4.6 KB LDS scratch and contracts it with W again into the 32 x K gradient rows, which leave through the same
// scratch as full 128-byte row segments (16 bytes per lane).  Reads and writes of the two big streams overlap in one kernel; pred and dpred (12 MB each) are still written
// (dpred feeds the weight / bias gradients of the Linear, pred is part of the engine's observable state).
#include "vame_common.h"
#include "gru_desc.h"

struct HeadParams {
    const float* Y; int64_t y_ld, y_seg, y_seg_stride;      // row m = (b,t), b = m / y_seg: Y + b*y_seg_stride + t*y_ld, K floats
    const float* W; const float* bias;                      // (F, K) row-major, (F)
    const float* tgt; int64_t tgt_row, tgt_off;             // target of row (b,t), feature f: tgt[b*tgt_row + tgt_off + t*F + f]
    float* pred; float* dpred;                              // (M, F) each; pred may be null
    float* dY; int64_t dy_ld;                               // (M, dy_ld), columns [0, K) written
    float* loss;                                            // loss[0] += sum of squared errors
    int M, F, K, Fp;                                        // Fp = F rounded up to 8 (rows of W kept in LDS)
    float gscale;
};

// LDS: W as Fp rows of (K + 4) floats (the +4 makes the 16-byte row-strided reads of phase 1 conflict-free), then one 32 x 36 scratch
// per wave for the C -> A layout turn.
__global__ __launch_bounds__(256) void head_fused_kernel(HeadParams P) {
    VAME_DYN_SMEM(smem_raw);
    float* wl = reinterpret_cast<float*>(smem_raw);
    const int K = P.K, LDW = K + 4, F = P.F;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = UNIFORM(tid >> 6);
    float* sc = wl + (size_t)P.Fp * LDW + w * (32 * 36);
    for (int i = tid; i < P.Fp * (K / 4); i += 256) {
        const int f = i / (K / 4), k4 = i % (K / 4);
        const float4 v = f < F ? reinterpret_cast<const float4*>(P.W + (int64_t)f * K)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&wl[f * LDW + 4 * k4]) = v;
    }
    __syncthreads();
    const int m0 = (blockIdx.x * 4 + w) * 32;                    // (M < 2^31: 32-bit row arithmetic, no 64-bit divisions)
    const int seg = (int)P.y_seg;
    float part = 0.f;
    if (m0 < P.M) {
        // ---- phase 1: prediction tile = states (A: row li, k = 8c + 4hh + e) x W^T (B: column n = li, same k)
        const int ma = m0 + li;
        const bool arow = ma < P.M;
        const int ba = arow ? ma / seg : 0;
        const float* yrow = P.Y + (int64_t)ba * P.y_seg_stride + (int64_t)(arow ? ma - ba * seg : 0) * P.y_ld + 4 * hh;
        const float* wrow = wl + (li < P.Fp ? li : 0) * LDW + 4 * hh;
        const bool bcol = li < P.Fp;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // passes of four chunks; the loads of two passes ahead are in flight while a pass feeds the MFMAs (a wave has nobody to hide
        // its HBM latency behind but itself and one neighbour on the SIMD)
        constexpr int PC = 4, PD = 3;
        const int npass = K / (8 * PC);                                          // K % 32 == 0
        float4 a[PD][PC];
        auto fetch = [&](int p, float4 (&dst)[PC]) {
#pragma unroll
            for (int j = 0; j < PC; ++j)
                dst[j] = (arow && p < npass) ? *reinterpret_cast<const float4*>(yrow + 8 * (p * PC + j)) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto feed = [&](int p, const float4 (&src)[PC]) {
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                const float4 b = bcol ? *reinterpret_cast<const float4*>(wrow + 8 * (p * PC + j)) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = MFMA_32x32x2(src[j].x, b.x, acc); acc = MFMA_32x32x2(src[j].y, b.y, acc);
                acc = MFMA_32x32x2(src[j].z, b.z, acc); acc = MFMA_32x32x2(src[j].w, b.w, acc);
            }
        };
        fetch(0, a[0]); fetch(1, a[1]);
        for (int p0 = 0; p0 < npass; p0 += PD) {                                 // unrolled by PD so that the buffer index is static
            fetch(p0 + 2, a[2]); feed(p0, a[0]);
            if (p0 + 1 < npass) { fetch(p0 + 3, a[0]); feed(p0 + 1, a[1]); }
            if (p0 + 2 < npass) { fetch(p0 + 4, a[1]); feed(p0 + 2, a[2]); }
        }
        // ---- error, loss, dpred: the lane holds column li of rows CR(r) + 4hh
        const float bl = li < F ? P.bias[li] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = CR(r) + 4 * hh;
            const int m = m0 + row;
            float g = 0.f;
            if (m < P.M && li < F) {
                const int b = m / seg, t = m - b * seg;
                const float p = acc[r] + bl;
                const float e = p - P.tgt[(int64_t)b * P.tgt_row + P.tgt_off + t * F + li];
                part += e * e;
                g = P.gscale * e;
                if (P.pred) P.pred[(int64_t)m * F + li] = p;
                P.dpred[(int64_t)m * F + li] = g;
            }
            sc[row * 36 + li] = g;
        }
        WAVE_SYNC();
        // ---- phase 2: gradient rows = dpred (A from the scratch: row li, k = f) x W (B: column n, row f of the LDS copy)
        float4 ga[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            ga[c] = 8 * c < P.Fp ? *reinterpret_cast<const float4*>(&sc[li * 36 + 8 * c + 4 * hh]) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int nb = 0; nb < K / 32; ++nb) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const float* wc = wl + nb * 32 + li;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (8 * c >= P.Fp) break;
                const float* wf = wc + (8 * c + 4 * hh) * LDW;
                const float av[4] = {ga[c].x, ga[c].y, ga[c].z, ga[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) o = MFMA_32x32x2(av[e], wf[e * LDW], o);
            }
            // the tile leaves as full 128-byte rows: C layout -> scratch -> 16 bytes per lane (row 8i + lane/8, columns 4 (lane % 8) ...)
            WAVE_SYNC();
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[(CR(r) + 4 * hh) * 36 + li] = o[r];
            WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + (lane >> 3), m = m0 + row;
                const float4 v = *reinterpret_cast<const float4*>(&sc[row * 36 + 4 * (lane & 7)]);
                if (m < P.M) *reinterpret_cast<float4*>(P.dY + (int64_t)m * P.dy_ld + nb * 32 + 4 * (lane & 7)) = v;
            }
        }
    }
    // one float atomic per workgroup (atomics on one address serialise)
    __shared__ float red[4];
    part = wave_sum(part);
    if (lane == 0) red[w] = part;
    __syncthreads();
    if (tid == 0) atomicAdd(P.loss, (red[0] + red[1]) + (red[2] + red[3]));
}

extern "C" int64_t vame_head_fused_lds_bytes(int F, int K) {
    const int Fp = (F + 7) / 8 * 8;
    return ((int64_t)Fp * (K + 4) + 4 * 32 * 36) * 4;
}

extern "C" int vame_head_fused_f32(const float* Y, int64_t y_ld, int64_t y_seg, int64_t y_seg_stride, int M, int F, int K, const float* W,
                                   const float* bias, const float* tgt, int64_t tgt_row, int64_t tgt_off, float gscale, float* pred,
                                   float* dpred, float* dY, int64_t dy_ld, float* loss, void* stream) {
    VAME_CHECK_ARG(Y && W && bias && tgt && dpred && dY && loss, VAME_E_BADARG, "head_fused: null pointer");
    VAME_CHECK_ARG(M >= 1 && F >= 1 && F <= 32 && K >= 32 && K % 32 == 0 && y_seg >= 1, VAME_E_SHAPE,
                   "head_fused: M=%d F=%d (1..32) K=%d (multiple of 32)", M, F, K);
    VAME_CHECK_ARG(y_ld % 4 == 0 && y_seg_stride % 4 == 0 && (uintptr_t)Y % 16 == 0 && (uintptr_t)W % 16 == 0 && dy_ld >= K && dy_ld % 4 == 0 &&
                       (uintptr_t)dY % 16 == 0, VAME_E_SHAPE,
                   "head_fused: state rows and W must be 16-byte aligned (y_ld=%lld, y_seg_stride=%lld)", (long long)y_ld, (long long)y_seg_stride);
    const int64_t lds = vame_head_fused_lds_bytes(F, K);
    VAME_CHECK_ARG(lds <= 160 * 1024, VAME_E_UNSUPPORTED, "head_fused: F=%d K=%d needs %lld bytes of LDS", F, K, (long long)lds);
#ifndef VAME_EMU
    VAME_CHECK_ARG(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                   VAME_E_HIP, "head_fused: cannot reserve %d bytes of LDS", (int)lds);
#endif
    HeadParams P;
    P.Y = Y; P.y_ld = y_ld; P.y_seg = y_seg; P.y_seg_stride = y_seg_stride; P.W = W; P.bias = bias; P.tgt = tgt; P.tgt_row = tgt_row;
    P.tgt_off = tgt_off; P.pred = pred; P.dpred = dpred; P.dY = dY; P.dy_ld = dy_ld; P.loss = loss; P.M = M; P.F = F; P.K = K;
    P.Fp = (F + 7) / 8 * 8; P.gscale = gscale;
    hipLaunchKernelGGL(head_fused_kernel, dim3((unsigned)cdiv64(M, 128)), dim3(256), (size_t)lds, (hipStream_t)stream, P);
    VAME_LAUNCH_CHECK("head_fused");
    return VAME_OK;
}
