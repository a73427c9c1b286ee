// Generic fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// k-ordered fmaf chain, 157 TF peak).  Used for every non-recurrent contraction of the path:
// GRU input projections, Linear layers, data gradients and the large-K weight gradients
// (split-K over batch x time).  Operand tiles are staged k-major in LDS so that the MFMA A/B
// fragments (lane = row/col, lane>>5 = k) are conflict-free ds_read_b32's; global loads are
// 16-byte and register-prefetched one k-tile ahead.
#include "vame_common.h"

#ifndef VAME_GEMM_EPI_DEFAULT
#define VAME_GEMM_EPI_DEFAULT 2      /* 128x128 tiles: LDS-transposed full-line stores; other tiles fall back to form 1 */
#endif
struct GemmOperand {
    const float* p;
    int64_t ld, seg, seg_stride;   // row i -> (seg ? (i/seg)*seg_stride + (i%seg)*ld : i*ld)
    int vec;                       // 16-byte loads legal
    int gap_at, gap;               // k-major operands: column x >= gap_at is read at x + gap (skips a block of a wider row)
};
struct GemmParams {
    GemmOperand A, B;
    const float* bias;
    float* C; int64_t ldc;
    float* ws;
    int M, N, K, kper, splitk, accumulate;
    int tiles_m, tiles_n, by_z;      // XCD-aware block->tile mapping (see map_tile)
    int cvec, wsvec;                 // 16-byte stores to C / to the split-K workspace are legal
    int group;                       // > 1: that many problems of identical shape / layout in one launch (split-K >= 8 only):
    const float* gA[8];              //      per-problem operand bases; partial sums of problem g go to ws + g * splitk * M * N
    const float* gB[8];
};

__device__ __forceinline__ int64_t op_row(const GemmOperand& o, int64_t i) {
    return o.seg ? (i / o.seg) * o.seg_stride + (i % o.seg) * o.ld : i * o.ld;
}

// 1-D grid -> (m-tile, n-tile, k-split).  Workgroup b runs on XCD b%8 (observed; speed only) and each XCD
// has a private 4 MiB L2, so tiles that stream the same operand panel are kept on ONE XCD: with
// split-K >= 8 a whole k-slab (all its m,n tiles) is a unit, otherwise an (k-split, m-panel) row of n-tiles.
// Units are dealt round-robin to XCDs; inside an XCD consecutive workgroups walk one unit's tiles.
__device__ __forceinline__ bool map_tile(const GemmParams& p, int& tm, int& tn, int& z, int& g) {
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int per_unit = p.by_z ? p.tiles_m * p.tiles_n : p.tiles_n;
    const int nunits = p.by_z ? p.splitk * p.group : p.splitk * p.tiles_m;
    const int u = (slot / per_unit) * 8 + xcd, w = slot % per_unit;
    g = 0;
    if (u >= nunits) return false;
    if (p.by_z) { z = u / p.group; g = u - z * p.group; tm = w / p.tiles_n; tn = w % p.tiles_n; }   // consecutive units cycle over the problems
    else { z = u / p.tiles_m; tm = u % p.tiles_m; tn = w; }
    return true;
}

// One thread's share of a (BK x BX) operand tile: fetched into registers (global, 16 B), stored k-major to LDS.
//   !KM (stored X x K, K contiguous): item = (x, k-quad); the thread's rows never change, so their
//        (two-level) row offsets are resolved once; 8 lanes cover one 128-byte line of a row.
//    KM (stored K x X, X contiguous): item = (k, x-quad); rows advance by BK per k-tile, the (segment,
//        index) pair of the two-level addressing is advanced incrementally (no division in the loop).
template <int BK, int BX, int NT, bool KM, bool XM>
struct TileIO {
    static constexpr int ITEMS = BK * BX / 4;
    static constexpr int PER = (ITEMS + NT - 1) / NT;
    static constexpr int KQ = BK / 4, XQ = BX / 4;
    float4 v[PER];
    int rowo[KM ? 1 : PER];               // !KM: row offsets in elements (-1 = out of range)
    static constexpr int KSTEP = KM ? NT / XQ : 1;     // KM: rows between consecutive items of a thread

    // Fast-path state (interior tiles: every load legal and 16 bytes wide): a workgroup-uniform base pointer (scalar registers) + ONE
    // 32-bit byte offset per item, advanced by constants from one k-tile to the next.  The f32-input MFMA runs on the SIMD's vector ALU
    // (MI355X_MICROARCH.md: at the f32 vector rate), so every VALU instruction of a GEMM wave is matrix time lost; the first form of this
    // path recomputed (segment, index) -> offset per item and k-tile with 64-bit multiplications and a `while` per item: 164 VALU
    // instructions per k-tile, 56 of them quarter-rate multiplies (~25 % of the 64 MFMAs' time, profiles/r04_gemm_valu.txt).  Now: 1 VALU
    // per item for plain rows, 6 for two-level rows, and no more registers than before (the split-K weight gradients share their CUs with
    // the narrow contractions of the side stream: a fourth wave per SIMD must still fit beside three of these).
    const char* fbase;                     // uniform: operand base + the offset of this workgroup's first row (rebased: offsets stay < 2^31)
    unsigned fo[KM ? 1 : PER];             // byte offset of the next load: per item (!KM: arbitrary rows), or of item 0 alone (KM: the thread's
                                           // items are rows KSTEP apart -- item i sits at fo[0] + i * fitem (+ fwrap if a segment border lies between))
    int ft[1];                             // KM, two-level rows: item 0's row index inside its segment
    unsigned fitem;                        // KM: bytes between two items of a thread inside one segment
    unsigned fstep, fwrap;                 // KM: bytes from one k-tile to the next (incl. the segment borders BK rows always cross); one more border
    int frem, fq, fseg;                    // KM: BK = fq * seg + frem; fseg = seg, or INT_MAX for plain rows
    bool f32ok;                            // the workgroup's rows span < 2^31 bytes (else: the general path)
    __device__ __forceinline__ void init(const GemmOperand& o, int x0, int X, int kb, int tid, int kper) {
        if (!KM) {
            const int64_t row_first = op_row(o, x0 < X ? x0 : 0);
            int64_t lo = 0, hi = 0;
#pragma unroll
            for (int it = 0; it < PER; ++it) {
                const int idx = tid + it * NT, x = idx / KQ, gx = x0 + x;
                rowo[it] = ((ITEMS % NT == 0 || idx < ITEMS) && gx < X) ? (int)op_row(o, gx) : -1;
                const int64_t rel = (rowo[it] >= 0 ? (int64_t)rowo[it] - row_first : 0) + 4 * (tid % KQ);
                fo[it] = (unsigned)(rel * 4);
                lo = rel < lo ? rel : lo; hi = rel > hi ? rel : hi;
            }
            fbase = reinterpret_cast<const char*>(o.p + row_first + kb);
            // rows of one tile: BX x the row pitch; k advances inside a row by at most kper floats
            const int64_t pitch = o.seg ? (o.seg_stride > o.ld * o.seg ? o.seg_stride : o.ld * o.seg) : o.ld;
            f32ok = ((int64_t)BX * pitch + kper + BK) * 4 < ((int64_t)1 << 31) && (!o.seg || o.seg_stride >= 0) && o.ld >= 0;
            fstep = BK * 4; fwrap = 0; frem = 0; fq = 0; ft[0] = 0; fitem = 0; fseg = 0x7fffffff;
        } else {
            const unsigned g = (unsigned)(kb + tid / XQ);
            rowo[0] = 0;
            const int gx = x0 + 4 * (tid % XQ);
            const int gxs = gx + (gx >= o.gap_at ? o.gap : 0);
            fq = o.seg ? BK / (int)o.seg : 0;
            frem = o.seg ? BK % (int)o.seg : 0;
            fseg = o.seg ? (int)o.seg : 0x7fffffff;
            const int64_t wrap = o.seg ? o.seg_stride - o.seg * o.ld : 0;
            const int64_t row_first = o.seg ? (int64_t)(kb / (int)o.seg) * o.seg_stride + (int64_t)(kb % (int)o.seg) * o.ld : (int64_t)kb * o.ld;
            fbase = reinterpret_cast<const char*>(o.p + row_first);
            fwrap = (unsigned)(wrap * 4);
            fstep = (unsigned)(((int64_t)BK * o.ld + (int64_t)fq * wrap) * 4);
            // the workgroup's k-slab: kper (+ BK of run-ahead) rows of pitch ld, plus one border per segment; at most one border between a
            // thread's first and last item (else: the general path)
            const int64_t span = ((int64_t)(kper + 2 * BK) * o.ld + (o.seg ? ((kper + 2 * BK) / o.seg + 2) * (wrap > 0 ? wrap : 0) : 0) + o.gap + BX) * 4;
            f32ok = span < ((int64_t)1 << 31) && wrap >= 0 && o.ld >= 0 && (!o.seg || KSTEP * (PER - 1) < o.seg);
            fitem = (unsigned)((int64_t)KSTEP * o.ld * 4);
            {
                const unsigned b = o.seg ? g / (unsigned)o.seg : 0u, t = o.seg ? g % (unsigned)o.seg : 0u;
                ft[0] = (int)t;
                fo[0] = (unsigned)((((o.seg ? (int64_t)b * o.seg_stride + (int64_t)t * o.ld : (int64_t)g * o.ld) - row_first) + gxs) * 4);
            }
        }
    }
    // KM fast path: byte offset of item `it` (see fo) and the step to the next k-tile.  Branch-free (the pipelined loop places these
    // between MFMAs): plain rows have fseg = INT_MAX, frem = 0, fwrap = 0 and fall through the same selects.
    __device__ __forceinline__ unsigned item_off(int it) const {
        unsigned off = fo[0] + (unsigned)it * fitem;
        if (it > 0) off += (ft[0] + KSTEP * it >= fseg) ? fwrap : 0u;
        return off;
    }
    __device__ __forceinline__ void next_tile() {
        const int tn = ft[0] + frem;
        const bool w = tn >= fseg;
        ft[0] = w ? tn - fseg : tn;
        fo[0] += fstep + (w ? fwrap : 0u);
    }
    // `fast` (wave-uniform): the tile is interior in X and in k, 16-byte loads are legal -> no predication
    template <bool FAST>
    __device__ __forceinline__ void fetch(const GemmOperand& o, int x0, int X, int k0, int ke, int tid) {
        const bool fast = FAST && o.vec && f32ok && (ITEMS % NT == 0) && x0 + BX <= X && k0 + BK <= ke;
        if (!KM) {
            const int gk = k0 + 4 * (tid % KQ);
            if (fast) {
#pragma unroll
                for (int it = 0; it < PER; ++it) v[it] = *reinterpret_cast<const float4*>(fbase + fo[it]);
#pragma unroll
                for (int it = 0; it < PER; ++it) fo[it] += BK * 4;
                return;
            }
#pragma unroll
            for (int it = 0; it < PER; ++it) {
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowo[it] >= 0 && gk < ke) {
                    const float* src = o.p + rowo[it] + gk;
                    if (o.vec && gk + 3 < ke) r = *reinterpret_cast<const float4*>(src);
                    else { r.x = src[0]; if (gk + 1 < ke) r.y = src[1]; if (gk + 2 < ke) r.z = src[2]; if (gk + 3 < ke) r.w = src[3]; }
                }
                v[it] = r;
            }
        } else {
            const int gx = x0 + 4 * (tid % XQ);
            const int gxs = gx + (gx >= o.gap_at ? o.gap : 0);       // source column (gap_at, gap multiples of 4)
            if (fast) {
#pragma unroll
                for (int it = 0; it < PER; ++it) v[it] = *reinterpret_cast<const float4*>(fbase + item_off(it));
                next_tile();
                return;
            }
            // (general path: edge tiles and the last, partial k-tile of a slab -- rows resolved from scratch, rare)
#pragma unroll
            for (int it = 0; it < PER; ++it) {
                const int gk = k0 + tid / XQ + it * KSTEP;
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((ITEMS % NT == 0 || tid + it * NT < ITEMS) && gk < ke && gx < X) {
                    const float* src = o.p + op_row(o, gk) + gxs;
                    if (o.vec && gx + 3 < X) r = *reinterpret_cast<const float4*>(src);
                    else { r.x = src[0]; if (gx + 1 < X) r.y = src[1]; if (gx + 2 < X) r.z = src[2]; if (gx + 3 < X) r.w = src[3]; }
                }
                v[it] = r;
            }
        }
    }
    // LDS image.  k-major [BK][BX+4] (default; fragment = ds_read_b32 of row k; a row-major (!KM) operand is transposed by
    // four ds_write_b32 per global float4), or for !KM operands with XM: x-major [BX][BK+4] -- the global float4 is stored
    // as-is with one ds_write_b128 and a lane reads float4 [x][8c + 4*(lane>>5)] = its operand for 4 MFMA steps (row
    // stride 36 floats = 9 x 16 B: conflict-free b128).  Measured: XM wins for the A operand (NN; NT since round 4), not for B.
    static constexpr bool XMAJ = !KM && XM;
    static constexpr int LD = XMAJ ? BK + 4 : BX + 4;
    static constexpr int LDS_FLOATS = XMAJ ? BX * (BK + 4) : BK * (BX + 4);
    // ---- item-wise, branch-free forms for the software-pipelined loop (interior tiles only: 16-byte loads legal, nothing
    // predicated; k-major operands: seg == 0 or seg >= BK, so advancing a row index by BK crosses at most one segment border)
    // (the fast-path state fp / ft / fstep / fwrap of init() serves the pipelined loop too: one pointer per item, constants per k-tile;
    // eligibility there guarantees seg == 0 or seg >= BK, i.e. at most the one conditional border)
    __device__ __forceinline__ void pipe_init(const GemmOperand&, int, int, int) {}
    __device__ __forceinline__ void pipe_fetch_item(const GemmOperand& o, int it) {
        if (KM) {
            v[it] = *reinterpret_cast<const float4*>(fbase + item_off(it));
            if (it == PER - 1) next_tile();           // (the pipelined loop fetches a tile's items in order 0 .. PER-1)
        } else {
            v[it] = *reinterpret_cast<const float4*>(fbase + fo[it]);
            fo[it] += BK * 4;
        }
    }
    __device__ __forceinline__ void store_item(float* lds, int tid, int it) const {
        const int idx = tid + it * NT;
        if (KM) {
            const int k = idx / XQ, xq = idx % XQ;
            *reinterpret_cast<float4*>(&lds[k * LD + 4 * xq]) = v[it];
        } else if (XMAJ) {
            const int x = idx / KQ, kq = idx % KQ;
            *reinterpret_cast<float4*>(&lds[x * LD + 4 * kq]) = v[it];
        } else {
            const int x = idx / KQ, kq = idx % KQ;
            lds[(4 * kq + 0) * LD + x] = v[it].x;
            lds[(4 * kq + 1) * LD + x] = v[it].y;
            lds[(4 * kq + 2) * LD + x] = v[it].z;
            lds[(4 * kq + 3) * LD + x] = v[it].w;
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + it * NT;
            if (ITEMS % NT == 0 || idx < ITEMS) {
                if (KM) {
                    const int k = idx / XQ, xq = idx % XQ;
                    *reinterpret_cast<float4*>(&lds[k * LD + 4 * xq]) = v[it];
                } else if (XMAJ) {
                    const int x = idx / KQ, kq = idx % KQ;
                    *reinterpret_cast<float4*>(&lds[x * LD + 4 * kq]) = v[it];
                } else {
                    const int x = idx / KQ, kq = idx % KQ;
                    lds[(4 * kq + 0) * LD + x] = v[it].x;
                    lds[(4 * kq + 1) * LD + x] = v[it].y;
                    lds[(4 * kq + 2) * LD + x] = v[it].z;
                    lds[(4 * kq + 3) * LD + x] = v[it].w;
                }
            }
        }
    }
    // fragment values of the 4 MFMA steps e = 0..3 of chunk c (k = 8c + 4*hh + e) for operand column block at xw
    __device__ __forceinline__ static void frag(const float* lds, int xw, int li, int hh, int c, float (&f)[4]) {
        if (!XMAJ) {
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = lds[(8 * c + 4 * hh + e) * LD + xw + li];
        } else {
            const float4 q = *reinterpret_cast<const float4*>(&lds[(xw + li) * LD + 8 * c + 4 * hh]);
            f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
        }
    }
};

#ifdef VAME_PROBE   // tools/probe_gemm.hip only: per-wave phase timers (s_memtime) written to a global buffer
__device__ long long* g_gemm_probe;
extern "C" int vame_probe_set_gemm(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_probe), &p, sizeof(p)); }
#define PROBE_T() ((long long)__builtin_amdgcn_s_memtime())
#define PROBE_SLOTS 12
#define PROBE_ENTRY const long long pr_entry = (long long)__builtin_amdgcn_s_memrealtime()
#define PROBE_DECL long long pt0 = PROBE_T(), pr0 = (long long)__builtin_amdgcn_s_memrealtime(), pa = pt0, ps = 0, pb1 = 0, pm = 0, pb2 = 0
#define PROBE_ADD(acc) do { const long long t_ = PROBE_T(); acc += t_ - pa; pa = t_; } while (0)
#else
#define PROBE_DECL
#define PROBE_ADD(acc)
#define PROBE_ENTRY
#endif

struct GemmTrue { static constexpr bool value = true; };
struct GemmFalse { static constexpr bool value = false; };

// One output value: split-K partial -> workspace slab z, else C (+ bias, + previous C when accumulating)
__device__ __forceinline__ void emit4(const GemmParams& p, float* ws, int z, int row, int col, float4 v) {
    if (p.splitk > 1) {
        float* d = ws + ((int64_t)z * p.M + row) * p.N + col;
        if (p.wsvec && col + 3 < p.N) { *reinterpret_cast<float4*>(d) = v; return; }
        d[0] = v.x; if (col + 1 < p.N) d[1] = v.y; if (col + 2 < p.N) d[2] = v.z; if (col + 3 < p.N) d[3] = v.w;
        return;
    }
    float* c = p.C + (int64_t)row * p.ldc + col;
    if (p.cvec && col + 3 < p.N) {
        if (p.bias) { const float* b = p.bias + col; v.x += b[0]; v.y += b[1]; v.z += b[2]; v.w += b[3]; }
        if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(c); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *reinterpret_cast<float4*>(c) = v;
        return;
    }
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (col + k < p.N) {
            float x = e[k] + (p.bias ? p.bias[col + k] : 0.f);
            if (p.accumulate) x += c[k];
            c[k] = x;
        }
}

// VAR (tuning variants, tools/microbench.py A/B): 1 unpredicated interior fetch, 4 s_setprio around the MFMAs,
//   8 software pipeline: two LDS images of the operand tiles and ONE barrier per k-tile -- while the MFMAs of tile t run from image
//   t&1, the same wave writes tile t+1 (fetched during tile t-1) into the other image and then re-issues the global loads for
//   tile t+2, all between MFMAs (an f32 MFMA holds the matrix pipe for 64 cycles: ~16 issue slots per MFMA are free).  The plain
//   form serialises [stage -> barrier -> MFMAs -> barrier] per workgroup and relies on three co-resident workgroups to fill each
//   other's gaps, which they do only partly (they fall into step: 83 % MFMA-pipe use inside the loop, tools/probe_gemm).
// EPI (epilogue): 0 = accumulator layout as computed (lane = output column): 16 dword stores per 32x32 tile, two full 128-byte
//   lines each.  1 = the MFMA operands are SWAPPED (D^T = B^T A^T), which puts an output ROW in each lane and four consecutive
//   columns in registers 4q..4q+3: 4 dwordx4 stores per tile.  2 = layout 0 transposed through a per-wave LDS scratch (the operand
//   buffers, free after the k loop): 4 dwordx4 stores per tile, 8 full lines each.
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int VAR, int EPI>
__global__ __launch_bounds__(WM * WN * 64, ((BM / WM) * (BN / WN) > 64 * 64 || (VAR & 8)) ? 2 : 3) void gemm_kernel(GemmParams p) {
    constexpr int BK = 32, NT = WM * WN * 64, TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr bool PIPE = (VAR & 8) != 0;
    // x-major image (global float4 stored as-is, b128 fragments) for a row-major A operand -- NN and, since round 4, NT: with the fetch's
    // address arithmetic gone the transposing stores' own (53 v_add_u32 + 16 ds_write2_b32 per k-tile) showed: NT 109.3 -> 113.2 TF;
    // the B operand keeps the transposing store (x-major B: 112.2, both: 111.4; tools/gemm_lib_ab.py)
    typedef TileIO<BK, BM, NT, AKM, true> TA;
    typedef TileIO<BK, BN, NT, BKM, false> TB;
    __shared__ __attribute__((aligned(16))) float As[(PIPE ? 2 : 1) * TA::LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[(PIPE ? 2 : 1) * TB::LDS_FLOATS];
    PROBE_ENTRY;
    int tm_, tn_, z, g;
    if (!map_tile(p, tm_, tn_, z, g)) return;
    GemmOperand opA = p.A, opB = p.B;
    float* ws = p.ws;
    if (p.group > 1) { opA.p = p.gA[g]; opB.p = p.gB[g]; ws += (int64_t)g * p.splitk * p.M * p.N; }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, hh = lane >> 5;
    const int wm = wv / WN, wn = wv % WN;
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const int kb = z * p.kper, ke = (kb + p.kper < p.K) ? kb + p.kper : p.K;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TA ta;
    TB tb;
    ta.init(opA, m0, p.M, kb, tid, p.kper);
    tb.init(opB, n0, p.N, kb, tid, p.kper);
    PROBE_DECL;
    bool piped = false;
    constexpr bool PIPE_STATIC = PIPE && TA::ITEMS == 4 * NT && TB::ITEMS == 4 * NT && TM == 2 && TN == 2;
    if constexpr (PIPE_STATIC) {
    // eligibility of the pipelined loop (wave-uniform): interior tile, 16-byte loads, whole k-tiles and at least three of them,
    // at most one segment border per BK rows of a k-major operand; everything else takes the plain loop below
    const bool pipe_ok = opA.vec && opB.vec && ta.f32ok && tb.f32ok && m0 + BM <= p.M &&
                         n0 + BN <= p.N && (ke - kb) % BK == 0 && ke - kb >= 3 * BK && (!AKM || opA.seg == 0 || opA.seg >= BK) &&
                         (!BKM || opB.seg == 0 || opB.seg >= BK);
    if (pipe_ok) {
        ta.pipe_init(opA, m0, kb, tid);
        tb.pipe_init(opB, n0, kb, tid);
        // one k-tile: 16 steps of 4 MFMAs (step s = 4c + e contracts k = {8c + e, 8c + 4 + e}); the fragments of chunk c + 1 are read
        // during chunk c, and ONE staging operation rides in every step: steps 0..7 write the registers of tile t+1 into the other
        // LDS image, steps 8..15 re-load them with tile t+2.  sched_barrier pins that order (hipcc would batch them otherwise).
        auto tile = [&](const float* Ac, const float* Bc, float* An, float* Bn, auto wtag, auto ltag) {
            constexpr bool W = decltype(wtag)::value, L = decltype(ltag)::value;
            float a[2][2][4], b[2][2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) { TA::frag(Ac, wm * (BM / WM) + i * 32, li, hh, 0, a[0][i]); TB::frag(Bc, wn * (BN / WN) + i * 32, li, hh, 0, b[0][i]); }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int q = c & 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int st = 4 * c + e;
                    acc[0][0] = EPI == 1 ? MFMA_32x32x2(b[q][0][e], a[q][0][e], acc[0][0]) : MFMA_32x32x2(a[q][0][e], b[q][0][e], acc[0][0]);
                    if (e == 0 && c < 3) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) { TA::frag(Ac, wm * (BM / WM) + i * 32, li, hh, c + 1, a[q ^ 1][i]); TB::frag(Bc, wn * (BN / WN) + i * 32, li, hh, c + 1, b[q ^ 1][i]); }
                    }
                    SCHED_FENCE();
                    acc[0][1] = EPI == 1 ? MFMA_32x32x2(b[q][1][e], a[q][0][e], acc[0][1]) : MFMA_32x32x2(a[q][0][e], b[q][1][e], acc[0][1]);
                    if (W && st < 4) ta.store_item(An, tid, st);
                    if (W && st >= 4 && st < 8) tb.store_item(Bn, tid, st - 4);
                    if (L && st >= 8 && st < 12) ta.pipe_fetch_item(opA, st - 8);
                    if (L && st >= 12) tb.pipe_fetch_item(opB, st - 12);
                    SCHED_FENCE();
                    acc[1][0] = EPI == 1 ? MFMA_32x32x2(b[q][0][e], a[q][1][e], acc[1][0]) : MFMA_32x32x2(a[q][1][e], b[q][0][e], acc[1][0]);
                    acc[1][1] = EPI == 1 ? MFMA_32x32x2(b[q][1][e], a[q][1][e], acc[1][1]) : MFMA_32x32x2(a[q][1][e], b[q][1][e], acc[1][1]);
                    SCHED_FENCE();
                }
            }
        };
#pragma unroll
        for (int it = 0; it < 4; ++it) { ta.pipe_fetch_item(opA, it); tb.pipe_fetch_item(opB, it); }
        ta.store(As, tid);
        tb.store(Bs, tid);
#pragma unroll
        for (int it = 0; it < 4; ++it) { ta.pipe_fetch_item(opA, it); tb.pipe_fetch_item(opB, it); }
        __syncthreads();
        int cur = 0, k0 = kb;
        for (; k0 + 2 * BK < ke; k0 += BK) {
            tile(As + cur * TA::LDS_FLOATS, Bs + cur * TB::LDS_FLOATS, As + (cur ^ 1) * TA::LDS_FLOATS, Bs + (cur ^ 1) * TB::LDS_FLOATS, GemmTrue{}, GemmTrue{});
            PROBE_ADD(pm);
            __syncthreads();           // image cur fully read, image cur^1 fully written
            PROBE_ADD(pb2);
            cur ^= 1;
        }
        tile(As + cur * TA::LDS_FLOATS, Bs + cur * TB::LDS_FLOATS, As + (cur ^ 1) * TA::LDS_FLOATS, Bs + (cur ^ 1) * TB::LDS_FLOATS, GemmTrue{}, GemmFalse{});
        __syncthreads();
        cur ^= 1;
        tile(As + cur * TA::LDS_FLOATS, Bs + cur * TB::LDS_FLOATS, As, Bs, GemmFalse{}, GemmFalse{});
        __syncthreads();               // (the LDS-transposed epilogue reuses the images)
        piped = true;
    }
    }
    if (!piped && kb < ke) { ta.template fetch<(VAR & 1) != 0>(opA, m0, p.M, kb, ke, tid); tb.template fetch<(VAR & 1) != 0>(opB, n0, p.N, kb, ke, tid); }
    if (!piped)
    for (int k0 = kb; k0 < ke; k0 += BK) {
        ta.store(As, tid);
        tb.store(Bs, tid);
        PROBE_ADD(ps);
        __syncthreads();
        PROBE_ADD(pb1);
        if (k0 + BK < ke) { ta.template fetch<(VAR & 1) != 0>(opA, m0, p.M, k0 + BK, ke, tid); tb.template fetch<(VAR & 1) != 0>(opB, n0, p.N, k0 + BK, ke, tid); }   // in flight during the MFMAs
        if (VAR & 4) SETPRIO(1);
        // MFMA step (c, e) contracts k = {8c + e, 8c + 4 + e}: any k order is valid as long as A and B agree
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) TA::frag(As, wm * (BM / WM) + i * 32, li, hh, c, a[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) TB::frag(Bs, wn * (BN / WN) + j * 32, li, hh, c, b[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = EPI == 1 ? MFMA_32x32x2(b[j][e], a[i][e], acc[i][j]) : MFMA_32x32x2(a[i][e], b[j][e], acc[i][j]);
        }
        if (VAR & 4) SETPRIO(0);
        PROBE_ADD(pm);
        __syncthreads();
        PROBE_ADD(pb2);
    }
#ifdef VAME_PROBE
    if (lane == 0 && g_gemm_probe) {
        long long* o = g_gemm_probe + ((long long)blockIdx.x * (NT / 64) + wv) * PROBE_SLOTS;
        o[0] = pt0; o[1] = PROBE_T(); o[2] = pr0; o[3] = (long long)__builtin_amdgcn_s_memrealtime();
        o[4] = ps; o[5] = pb1; o[6] = pm; o[7] = pb2;
    }
#endif
    if (EPI == 1) {
        // lane = output row, registers 4q..4q+3 = columns 8q + 4*hh + (0..3) of the 32x32 tile
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * (BM / WM) + i * 32 + li;
            if (row >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int colb = n0 + wn * (BN / WN) + j * 32 + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = colb + 8 * q;
                    if (col < p.N) emit4(p, ws, z, row, col, make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
                }
            }
        }
    } else if (EPI == 2) {
        constexpr int LDT = 36;
        static_assert(EPI != 2 || (TA::LDS_FLOATS >= (WM * WN / 2) * 32 * LDT && TB::LDS_FLOATS >= (WM * WN / 2) * 32 * LDT), "scratch does not fit");
        float* scr = (wv < WM * WN / 2 ? As + wv * 32 * LDT : Bs + (wv - WM * WN / 2) * 32 * LDT);      // (the loop's last barrier freed As / Bs)
        const int rr = lane >> 3, c4 = 4 * (lane & 7);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[frag_row(r, lane) * LDT + li] = acc[i][j][r];
                WAVE_SYNC();
                const int col = n0 + wn * (BN / WN) + j * 32 + c4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = m0 + wm * (BM / WM) + i * 32 + rr + 8 * q;
                    const float4 v = *reinterpret_cast<const float4*>(&scr[(rr + 8 * q) * LDT + c4]);
                    if (row < p.M && col < p.N) emit4(p, ws, z, row, col, v);
                }
                WAVE_SYNC();
            }
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + li;
            if (col >= p.N) continue;
            const float bv = (p.bias && p.splitk == 1) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 32 + frag_row(r, lane);
                if (row >= p.M) continue;
                float v = acc[i][j][r];
                if (p.splitk > 1) {
                    ws[((int64_t)z * p.M + row) * p.N + col] = v;
                } else {
                    float* c = p.C + (int64_t)row * p.ldc + col;
                    v += bv;
                    if (p.accumulate) v += *c;
                    *c = v;
                }
            }
        }
    }
#ifdef VAME_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wave's slot is held until its stores are acknowledged
    if (lane == 0 && g_gemm_probe) {
        long long* o = g_gemm_probe + ((long long)blockIdx.x * (NT / 64) + wv) * PROBE_SLOTS;
        o[8] = pr_entry; o[9] = (long long)__builtin_amdgcn_s_memrealtime();
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[10] = (long long)hwid | ((long long)xcc << 32);
    }
#endif
}

// Split-K reduction.  One summation order everywhere, so that a problem gives the same bits whether it ran alone or in a group and
// whichever thread layout reduced it: four interleaved partial sums p_k = sum over slabs z = k (mod 4) in increasing z, then
// (p0 + p1) + (p2 + p3).  SPREAD: few outputs and many slabs (the 24-row heads: 12288 outputs x 192 slabs) -- a block takes 64
// outputs and gives each partial sum its own 64 threads, 4x the loads in flight; otherwise one thread per output.
template <bool SPREAD>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ ws, int splitk, int M, int N, const float* __restrict__ bias,
                                                   float* __restrict__ C, int64_t ldc, int accumulate) {
    const int64_t n = (int64_t)M * N;
    auto finish = [&](int64_t i, float v) {
        const int row = (int)(i / N), col = (int)(i % N);
        if (bias) v += bias[col];
        float* c = C + (int64_t)row * ldc + col;
        if (accumulate) v += *c;
        *c = v;
    };
    if (SPREAD) {
        __shared__ float part[4][64];
        const int o = threadIdx.x & 63, k = threadIdx.x >> 6;
        for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < n; i0 += (int64_t)gridDim.x * 64) {
            const int64_t i = i0 + o;
            float p = 0.f;
            if (i < n)
                for (int z = k; z < splitk; z += 4) p += ws[(int64_t)z * n + i];
            part[k][o] = p;
            __syncthreads();
            if (k == 0 && i < n) finish(i, (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]));
            __syncthreads();
        }
    } else {
        // 16-byte form (same chains, same order per output: the same bits): four outputs per thread, a quarter of the load instructions -- the scalar
        // form below moves the 38 MB of the six-problem reduction at 1.5 TB/s
        if ((N & 3) == 0 && (ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(C)) & 15) == 0 &&
            (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) {
            const int64_t n4 = n / 4;
            const float4* w4 = reinterpret_cast<const float4*>(ws);
            auto add4 = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
            for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
                float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;
                int z = 0;
                for (; z + 4 <= splitk; z += 4) {
                    add4(p0, w4[(int64_t)z * n4 + i]); add4(p1, w4[(int64_t)(z + 1) * n4 + i]); add4(p2, w4[(int64_t)(z + 2) * n4 + i]); add4(p3, w4[(int64_t)(z + 3) * n4 + i]);
                }
                if (z < splitk) add4(p0, w4[(int64_t)z * n4 + i]);
                if (z + 1 < splitk) add4(p1, w4[(int64_t)(z + 1) * n4 + i]);
                if (z + 2 < splitk) add4(p2, w4[(int64_t)(z + 2) * n4 + i]);
                float4 v = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
                const int64_t e = 4 * i;
                const int row = (int)(e / N), col = (int)(e % N);
                if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                float4* c = reinterpret_cast<float4*>(C + (int64_t)row * ldc + col);
                if (accumulate) { const float4 o = *c; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *c = v;
            }
            return;
        }
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            int z = 0;
            for (; z + 4 <= splitk; z += 4) {
                p0 += ws[(int64_t)z * n + i]; p1 += ws[(int64_t)(z + 1) * n + i]; p2 += ws[(int64_t)(z + 2) * n + i]; p3 += ws[(int64_t)(z + 3) * n + i];
            }
            if (z < splitk) p0 += ws[(int64_t)z * n + i];
            if (z + 1 < splitk) p1 += ws[(int64_t)(z + 1) * n + i];
            if (z + 2 < splitk) p2 += ws[(int64_t)(z + 2) * n + i];
            finish(i, (p0 + p1) + (p2 + p3));
        }
    }
}
template <bool SPREAD>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splitk, int M, int N,
                                                            const float* __restrict__ bias, float* __restrict__ C,
                                                            int64_t ldc, int accumulate) {
    splitk_reduce_body<SPREAD>(ws, splitk, M, N, bias, C, ldc, accumulate);
}

struct GemmGroupOut { float* C[8]; };
template <bool SPREAD>
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const float* __restrict__ ws, int splitk, int M, int N, GemmGroupOut out,
                                                                  int64_t ldc, int accumulate) {
    splitk_reduce_body<SPREAD>(ws + (int64_t)blockIdx.y * splitk * (int64_t)M * N, splitk, M, N, nullptr, out.C[blockIdx.y], ldc, accumulate);
}
// launch geometry of a reduction over n outputs: {spread, blocks}
static inline bool reduce_spread(int64_t n, int splitk) { return splitk >= 16 && n < 256 * 256; }
static inline int reduce_blocks(int64_t n, bool spread) {
    const int64_t b = cdiv64(n, spread ? 64 : 256);
    return (int)(b < 1024 ? b : 1024);
}

#include <stdlib.h>
template <int BM, int BN, int WM, int WN, int VAR, int EPI>
static int launch_gemm_epi(const GemmParams& p, int akm, int bkm, dim3 grid, dim3 block, hipStream_t st) {
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, false, VAR, EPI>), grid, block, 0, st, p);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, true, VAR, EPI>), grid, block, 0, st, p);
    else if (akm && bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, true, VAR, EPI>), grid, block, 0, st, p);
    else return VAME_E_UNSUPPORTED;
    return VAME_OK;
}
// epilogue form per tile shape (see gemm_kernel): the LDS-transposed one needs the 128x128 tile's operand buffers as scratch
template <int BM, int BN, int WM, int WN, int VAR>
static int launch_gemm_var(const GemmParams& p, int akm, int bkm, dim3 grid, dim3 block, hipStream_t st) {
    int epi = VAME_GEMM_EPI_DEFAULT;
#if defined(VAME_EMU) || defined(VAME_GEMM_AB)      // host-emulator tests and the A/B tuning build exercise every form
    if (const char* e = getenv("VAME_GEMM_EPI")) epi = atoi(e);
#endif
    if (BM == 128 && BN == 128) {
        if (epi == 2) return launch_gemm_epi<BM, BN, WM, WN, VAR, (BM == 128 && BN == 128) ? 2 : 1>(p, akm, bkm, grid, block, st);
        if (epi == 0) return launch_gemm_epi<BM, BN, WM, WN, VAR, 0>(p, akm, bkm, grid, block, st);
        return launch_gemm_epi<BM, BN, WM, WN, VAR, 1>(p, akm, bkm, grid, block, st);
    }
    if (epi == 0) return launch_gemm_epi<BM, BN, WM, WN, VAR, 0>(p, akm, bkm, grid, block, st);
    return launch_gemm_epi<BM, BN, WM, WN, VAR, 1>(p, akm, bkm, grid, block, st);
}
template <int BM, int BN, int WM, int WN>
static int launch_gemm(GemmParams p, int akm, int bkm, hipStream_t st) {
    p.tiles_m = (int)cdiv64(p.M, BM); p.tiles_n = (int)cdiv64(p.N, BN);
    p.by_z = p.splitk >= 8;
    p.cvec = ((uintptr_t)p.C % 16 == 0) && (p.ldc % 4 == 0);
    p.wsvec = p.ws && ((uintptr_t)p.ws % 16 == 0) && (p.N % 4 == 0);
    const int64_t per_unit = p.by_z ? (int64_t)p.tiles_m * p.tiles_n : p.tiles_n;
    const int64_t nunits = p.by_z ? (int64_t)p.splitk * p.group : (int64_t)p.splitk * p.tiles_m;
    dim3 grid((unsigned)(cdiv64(nunits, 8) * per_unit * 8)), block(WM * WN * 64);
#if defined(VAME_EMU) || defined(VAME_GEMM_AB)
    if (!(BM == 128 && BN == 128)) {         // other tile shapes: the plain-loop variants only
        const char* e = getenv("VAME_GEMM_VAR_SKINNY");
        switch (e ? atoi(e) : -1) {
            case 0: return launch_gemm_var<BM, BN, WM, WN, 0>(p, akm, bkm, grid, block, st);
            case 1: return launch_gemm_var<BM, BN, WM, WN, 1>(p, akm, bkm, grid, block, st);
            case 5: return launch_gemm_var<BM, BN, WM, WN, 5>(p, akm, bkm, grid, block, st);
            default: break;
        }
    }
    if (BM == 128 && BN == 128) {            // A/B tuning build (make ab) and host-emulator tests: variant chosen per call from the environment
        const char* e = getenv("VAME_GEMM_VAR");
        switch (e ? atoi(e) : -1) {
            case 0: return launch_gemm_var<BM, BN, WM, WN, 0>(p, akm, bkm, grid, block, st);
            case 1: return launch_gemm_var<BM, BN, WM, WN, 1>(p, akm, bkm, grid, block, st);
            case 4: return launch_gemm_var<BM, BN, WM, WN, 4>(p, akm, bkm, grid, block, st);
            case 5: return launch_gemm_var<BM, BN, WM, WN, 5>(p, akm, bkm, grid, block, st);
            case 8: return launch_gemm_var<BM, BN, WM, WN, 8>(p, akm, bkm, grid, block, st);
            case 9: return launch_gemm_var<BM, BN, WM, WN, 9>(p, akm, bkm, grid, block, st);
            case 13: return launch_gemm_var<BM, BN, WM, WN, 13>(p, akm, bkm, grid, block, st);
            default: break;
        }
    }
#endif
    // measured on MI355X (interleaved A/B, tools/microbench.py gemm_ab / gemm_sk, profiles/r02_gemm_variants.txt): the
    // software-pipelined loop (VAR 13) wins wherever a tile has few k-tiles per output tile -- +14 % on the NT form (gi
    // projections, 104 -> 119 TF), +9 % on the NN form (dx, 115 -> 125 TF), +11 % on an un-split TN square -- and ties the plain loop
    // with 3 workgroups per CU on the split-K weight gradients (112-122 TF both: those run at a 1.99 GHz shader clock, i.e. at
    // 86-92 % of what the clock allows), which keep the plain loop + s_setprio (VAR 5).  Other tile shapes: plain loop.
    if (BM == 128 && BN == 128) {
        if (akm && bkm && p.splitk >= 8) return launch_gemm_var<BM, BN, WM, WN, 5>(p, akm, bkm, grid, block, st);
        // a single k-tile (the K = 24 / 30 forms: dY of the output heads, projections of z) has nothing to pipeline and is bound by its
        // epilogue: the plain loop's 3 workgroups per CU stream it 25-35 % faster than the pipelined form's 2 (40 vs 30 TF at 122880 x 512 x 24)
        if (p.kper <= 32) return launch_gemm_var<BM, BN, WM, WN, 0>(p, akm, bkm, grid, block, st);
        return launch_gemm_var<BM, BN, WM, WN, 13>(p, akm, bkm, grid, block, st);
    }
    if (!akm && bkm) return launch_gemm_var<BM, BN, WM, WN, 1>(p, akm, bkm, grid, block, st);
    return launch_gemm_var<BM, BN, WM, WN, 0>(p, akm, bkm, grid, block, st);
}

static int operand_vec(const float* p, int64_t ld, int64_t seg, int64_t seg_stride) {
    return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0) && (seg == 0 || seg_stride % 4 == 0);
}

extern "C" int vame_gemm_f32(int M, int N, int K, const float* A, int64_t lda, int a_kmajor, int64_t a_seg,
                             int64_t a_seg_stride, const float* B, int64_t ldb, int b_kmajor, int64_t b_seg,
                             int64_t b_seg_stride, const float* bias, float* C, int64_t ldc, int accumulate, int splitk,
                             float* ws, int a_gap_at, int a_gap, void* stream) {
    VAME_CHECK_ARG(M >= 1 && N >= 1 && K >= 1, VAME_E_SHAPE, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
    VAME_CHECK_ARG(A && B && C, VAME_E_BADARG, "gemm: null operand");
    VAME_CHECK_ARG(!(a_kmajor && !b_kmajor), VAME_E_UNSUPPORTED, "gemm: A k-major with B n-major is not provided");
    VAME_CHECK_ARG(splitk >= 1 && (splitk == 1 || ws), VAME_E_BADARG, "gemm: splitk=%d needs a workspace", splitk);
    GemmParams p;
    VAME_CHECK_ARG(a_gap == 0 || (a_kmajor && a_gap_at % 4 == 0 && a_gap % 4 == 0), VAME_E_BADARG,
                   "gemm: a column gap needs a k-major A and multiples of 4");
    p.A = {A, lda, a_seg, a_seg_stride, operand_vec(A, lda, a_seg, a_seg_stride), a_gap ? a_gap_at : 0x7fffffff, a_gap};
    p.B = {B, ldb, b_seg, b_seg_stride, operand_vec(B, ldb, b_seg, b_seg_stride), 0x7fffffff, 0};
    p.bias = bias; p.C = C; p.ldc = ldc; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.accumulate = accumulate; p.group = 1;
    int kper = (int)(cdiv64(cdiv64(K, splitk), 32) * 32);
    p.kper = kper;
    p.splitk = (int)cdiv64(K, kper);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (M <= 32 && N > 64) rc = launch_gemm<32, 128, 1, 4>(p, a_kmajor, b_kmajor, st);      // skinny outputs (dW of the 24/30-row heads)
#if !defined(VAME_EMU) && defined(VAME_GEMM_AB)
    else if (N > 64 && getenv("VAME_GEMM_TILE") && atoi(getenv("VAME_GEMM_TILE")) == 1 && M >= 256)
        rc = launch_gemm<256, 128, 2, 2>(p, a_kmajor, b_kmajor, st);                          // 128x64 per wave (A/B tuning build only)
    else if (N > 64 && getenv("VAME_GEMM_TILE") && atoi(getenv("VAME_GEMM_TILE")) == 2 && M >= 256)
        rc = launch_gemm<256, 128, 4, 2>(p, a_kmajor, b_kmajor, st);                          // 8 waves, 64x64 per wave
    else if (N <= 32 && getenv("VAME_GEMM_TILE") && atoi(getenv("VAME_GEMM_TILE")) == 3)
        rc = launch_gemm<64, 32, 2, 1>(p, a_kmajor, b_kmajor, st);                            // skinny N, more workgroups in flight (A/B)
    else if (N <= 32 && getenv("VAME_GEMM_TILE") && atoi(getenv("VAME_GEMM_TILE")) == 4)
        rc = launch_gemm<256, 32, 8, 1>(p, a_kmajor, b_kmajor, st);
#endif
    else if (N > 64) rc = launch_gemm<128, 128, 2, 2>(p, a_kmajor, b_kmajor, st);
    else if (N > 32) rc = launch_gemm<128, 64, 4, 1>(p, a_kmajor, b_kmajor, st);
    else rc = launch_gemm<128, 32, 4, 1>(p, a_kmajor, b_kmajor, st);
    VAME_CHECK_ARG(rc == VAME_OK, rc, "gemm: unsupported layout");
    VAME_LAUNCH_CHECK("gemm");
    if (p.splitk > 1) {
        const int64_t n = (int64_t)M * N;
        if (reduce_spread(n, p.splitk))
            hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(reduce_blocks(n, true)), dim3(256), 0, st, (const float*)ws, p.splitk, M, N,
                               bias, C, ldc, accumulate);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(reduce_blocks(n, false)), dim3(256), 0, st, (const float*)ws, p.splitk, M, N,
                               bias, C, ldc, accumulate);
        VAME_LAUNCH_CHECK("gemm splitk reduce");
    }
    return VAME_OK;
}

// Several weight-gradient style problems of one shape and layout in ONE launch: C_g = A_g^T B_g (or any supported layout) with
// split-K >= 8.  The k-slabs of all problems are dealt to the XCDs together, so the launch has count x as many workgroups as a
// single problem: fewer partial sums per problem for the same occupancy, no kernel boundary (and no idle tail) between the
// problems, one reduction launch.  Only the operand base pointers differ between the problems.  Problems that all name the SAME C
// are summed into it: C (+)= sum_g A_g B_g (the input gradient of a vector that feeds several layers).
extern "C" int vame_gemm_group_f32(int count, int M, int N, int K, const float* const* A, int64_t lda, int a_kmajor, int64_t a_seg,
                                   int64_t a_seg_stride, const float* const* B, int64_t ldb, int b_kmajor, int64_t b_seg,
                                   int64_t b_seg_stride, float* const* C, int64_t ldc, int accumulate, int splitk, float* ws,
                                   int a_gap_at, int a_gap, void* stream) {
    VAME_CHECK_ARG(count >= 1 && count <= 8 && A && B && C && ws, VAME_E_BADARG, "gemm_group: count=%d (1..8) / null table", count);
    VAME_CHECK_ARG(M >= 1 && N >= 1 && K >= 1, VAME_E_SHAPE, "gemm_group: M=%d N=%d K=%d", M, N, K);
    VAME_CHECK_ARG(!(a_kmajor && !b_kmajor), VAME_E_UNSUPPORTED, "gemm_group: A k-major with B n-major is not provided");
    VAME_CHECK_ARG(a_gap == 0 || (a_kmajor && a_gap_at % 4 == 0 && a_gap % 4 == 0), VAME_E_BADARG,
                   "gemm_group: a column gap needs a k-major A and multiples of 4");
    GemmParams p;
    int vecA = 1, vecB = 1;
    for (int g = 0; g < count; ++g) {
        VAME_CHECK_ARG(A[g] && B[g] && C[g], VAME_E_BADARG, "gemm_group: problem %d has a null operand", g);
        p.gA[g] = A[g]; p.gB[g] = B[g];
        vecA &= operand_vec(A[g], lda, a_seg, a_seg_stride);
        vecB &= operand_vec(B[g], ldb, b_seg, b_seg_stride);
    }
    for (int g = count; g < 8; ++g) { p.gA[g] = A[0]; p.gB[g] = B[0]; }
    p.A = {A[0], lda, a_seg, a_seg_stride, vecA, a_gap ? a_gap_at : 0x7fffffff, a_gap};
    p.B = {B[0], ldb, b_seg, b_seg_stride, vecB, 0x7fffffff, 0};
    p.bias = nullptr; p.C = C[0]; p.ldc = ldc; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.accumulate = accumulate; p.group = count;
    p.kper = (int)(cdiv64(cdiv64(K, splitk), 32) * 32);
    p.splitk = (int)cdiv64(K, p.kper);
    VAME_CHECK_ARG(p.splitk >= 8, VAME_E_SHAPE, "gemm_group: split-K %d < 8 (grouping is for large-K contractions)", p.splitk);
    hipStream_t st = (hipStream_t)stream;
    // same tile choice as vame_gemm_f32 for the output width (narrow outputs: the dW_ih of GRUs fed by the latent vector -- tiny
    // problems that are grouped for their launch count, not for occupancy)
    int rc;
#if !defined(VAME_EMU) && defined(VAME_GEMM_AB)      // tuning build: 256 x 128 tiles for the grouped form (tools/gemm_group_ab.py; measured 15-30 % slower)
    if (N > 64 && M >= 256 && getenv("VAME_GEMM_TILE") && atoi(getenv("VAME_GEMM_TILE")) == 1) rc = launch_gemm<256, 128, 2, 2>(p, a_kmajor, b_kmajor, st);
    else if (N > 64 && M >= 256 && getenv("VAME_GEMM_TILE") && atoi(getenv("VAME_GEMM_TILE")) == 2) rc = launch_gemm<256, 128, 4, 2>(p, a_kmajor, b_kmajor, st);
    else
#endif
    rc = N > 64 ? launch_gemm<128, 128, 2, 2>(p, a_kmajor, b_kmajor, st)
       : N > 32 ? launch_gemm<128, 64, 4, 1>(p, a_kmajor, b_kmajor, st) : launch_gemm<128, 32, 4, 1>(p, a_kmajor, b_kmajor, st);
    VAME_CHECK_ARG(rc == VAME_OK, rc, "gemm_group: unsupported layout");
    VAME_LAUNCH_CHECK("gemm_group");
    GemmGroupOut out;
    bool one_c = count > 1;
    for (int g = 0; g < 8; ++g) { out.C[g] = C[g < count ? g : 0]; one_c = one_c && out.C[g] == C[0]; }
    const int64_t n = (int64_t)M * N;
    if (one_c) {        // every problem names the same C: C = sum_g A_g B_g, the problems' partial sums are one stack of count x splitk slabs
        const int slabs = count * p.splitk;
        if (reduce_spread(n, slabs))
            hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(reduce_blocks(n, true)), dim3(256), 0, st, (const float*)ws, slabs, M, N,
                               (const float*)nullptr, C[0], ldc, accumulate);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(reduce_blocks(n, false)), dim3(256), 0, st, (const float*)ws, slabs, M, N,
                               (const float*)nullptr, C[0], ldc, accumulate);
    } else if (reduce_spread(n, p.splitk)) {
        hipLaunchKernelGGL(splitk_reduce_group_kernel<true>, dim3(reduce_blocks(n, true), count), dim3(256), 0, st, (const float*)ws, p.splitk, M, N,
                           out, ldc, accumulate);
    } else {
        hipLaunchKernelGGL(splitk_reduce_group_kernel<false>, dim3(reduce_blocks(n, false), count), dim3(256), 0, st, (const float*)ws, p.splitk, M, N,
                           out, ldc, accumulate);
    }
    VAME_LAUNCH_CHECK("gemm_group reduce");
    return VAME_OK;
}

// =====================================================================================================================================
// Error-compensated SPLIT-bf16 form of the grouped k-major x k-major contraction (weight gradients: C_g = A_g^T B_g over K = batch x time).
// Opt-in (vame_gemm_group_bf16x6_f32); the default path above stays on the f32-input matrix cores.
//
// Why: v_mfma_f32_32x32x2_f32 issues at the vector-ALU rate (157 TF) and occupies the SIMD's VALU lanes, v_mfma_f32_32x32x16_bf16 runs
// 16x faster on its own pipe.  Every fp32 operand value is split EXACTLY into three bf16 planes by truncation, x = x1 + x2 + x3 (8 + 8 + 8
// significand bits, fp32's exponent range, so no scaling: x1 = the upper 16 bits of x, x2 = the upper 16 bits of the exact remainder
// x - x1, x3 = what is left), and the contraction is evaluated as the six plane products (1,1) (1,2) (2,1) (2,2) (1,3) (3,1) with fp32
// accumulation: the dropped products (2,3) (3,2) (3,3) are below 2^-21 of |x||y| and a bf16 x bf16 product is exact in fp32.  With the
// leading product on its own accumulator (NACC = 2) the low-order terms (<= 2^-7 of it) add no rounding of their own: measured error 0.4x
// the f32-input kernel's (one rounding per K = 16 block instead of sixteen); with one accumulator for all six products (NACC = 1) it is
// the f32-input kernel's level.  6/16 of the f32-input MFMA's pipe time per flop.
//
// Mapping (the fastest of five that were built and measured, see below): 128 x 128 output tile, BK = 32, 256-thread workgroups of FOUR
// SYMMETRIC waves, two (NACC = 2) or three (NACC = 1) workgroups per CU.  Per k-tile every wave
//   1. splits its share of the tile -- operand wave >> 1, rows [16 (wave & 1), + 16), a lane owns two adjacent columns: 16
//      buffer_load_dwordx2 of wave-uniform rows (the two-level (batch, time) row offsets are computed once per k-tile on 16 lanes and handed
//      out with v_readlane as the loads' scalar offsets), ~9 VALU per four values (v_and / v_pk_add_f32 / v_perm_b32), 12 ds_write_b128;
//   2. issues the loads of the next k-tile (in flight during the MFMAs), LDS-only barrier;
//   3. contracts its 64 x 64 quarter: 24 ds_read_b128 + 48 MFMAs; LDS-only barrier.
// The workgroups of a CU fill each other's split / barrier phases with MFMAs (the scheme of gemm_kernel's plain loop).
// LDS image (48 KB): [plane][operand][k-octet][column] in 16-byte units (8 bf16 of one column = one MFMA operand of a lane); columns are
// stored even | odd, the odd half rotated by 8 slots, so the 16-byte stores and the fragment reads are both conflict-free.  The k order
// inside a k-tile is a permutation (an octet = eight consecutive rows), identical for A and B.
//
// What bounds it (tools/split_abl.py on the dominant launch, 6 x 768 x 256 x 122,880, profiles/r05_split_abl.txt; tools/
// mfma_valu_overlap_probe.hip, profiles/r05_mfma_valu_overlap_probe.txt): the MFMAs alone take 0.75-0.8 ms, but (a) with 128 x 128 tiles the
// operand panels pass through the L1s 9.06 GB per launch (A twice, B six times) and that stream alone takes 0.85 ms whatever the load width
// or the number of waves in flight (~10.6 TB/s L2 -> L1); (b) the split is 576 VALU wave-instructions per CU and k-tile; one wave issues
// them at 7-8 cycles each and at 14 beside another wave's MFMAs (v_pk_add_f32 15.5), so they need every wave of the CU; (c) the planes are
// 48 KB of ds_write_b128 per k-tile at ~73 B/clk; (d) with MFMAs, L2 traffic and VALU all active the shader clock falls to 1.9-2.1 GHz
// (f32-input kernel: 2.2).  Four other mappings measured 12-35 % slower and live in tools/gemm_split_variants.inc (tuning build only):
// wave-specialised 4 MFMA + 4 split waves with three LDS images (2.0 ms: one split wave per SIMD is VALU-issue-bound), 256 x 128 tiles with
// eight waves and two images (1.85 ms: 6.8 GB through the L1s, but the phases of a wave's in-order stream do not overlap), 4 MFMA + 8 split
// waves (1.89 ms: the MFMA stream alone runs at 0.77 ms = 34 cycles per MFMA with one fragment read between the MFMAs, the producers need 1.06), and the
// 256 x 128 mapping as one hand-ordered instruction stream per interval (2.09 ms).
// =====================================================================================================================================
namespace split6 {
constexpr int BK = 32, TILE = 128;
constexpr int REGION = 2048;               // one (plane, operand, k-octet): 128 columns x 8 bf16
constexpr int PLANE = 8 * REGION;          // [operand A | B][octet 0..3]
constexpr int BUF = 3 * PLANE;             // three planes of one k-tile
constexpr int NBUF = 3;
constexpr unsigned HI16 = 0xffff0000u;

__device__ __forceinline__ int xslot(int x) { return ((x & 1) << 10) + (((x >> 1) ^ ((x & 1) << 3)) << 4); }
__device__ __forceinline__ unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float ufloat(unsigned u) { return __builtin_bit_cast(float, u); }
// the bf16 (upper) halves of two fp32 values in one dword: lo in bits 0..15
__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
#ifdef VAME_EMU
    return (fbits(lo) >> 16) | (fbits(hi) & HI16);
#else
    return __builtin_amdgcn_perm(fbits(hi), fbits(lo), 0x07060302u);
#endif
}
__device__ __forceinline__ int uniform(int v) {
#ifdef VAME_EMU
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// A producer wave's position in a k-major operand: the first of its 16 rows of the k-tile it loads next.  Wave-uniform (scalar
// registers).  Two-level rows: row g = (b, t) = (g / seg, g % seg) sits at b * seg_stride + t * ld.  Offsets are bytes relative to the
// slab's first row (the buffer range's base), < 2^30 (checked on the host).  The cursor never moves past the slab's last row: tiles
// behind the end of the slab re-read its last rows (they are prefetched unconditionally so that every interval issues the same loads)
// and rows >= ke of the last tile are zeroed by the caller.
struct RowCursor {
    unsigned off0, ld4, wrap4;
    int g0, t0, seg, ke;
    __device__ __forceinline__ void init(const GemmOperand& o, int kb, int g, int ke_) {
        ke = ke_;
        ld4 = (unsigned)(o.ld * 4);
        g0 = g < ke ? g : ke - 1;
        if (o.seg) {
            seg = (int)o.seg; wrap4 = (unsigned)((o.seg_stride - o.seg * o.ld) * 4);
            t0 = g0 % seg;
        } else { seg = 0x7fffffff; wrap4 = 0; t0 = 0; }
        off0 = (unsigned)((op_row(o, g0) - op_row(o, kb)) * 4);
    }
    // byte offsets of rows g0 + min(j, rows left) for j = lane & 15 (one row per lane, computed on the vector ALU once per k-tile)
    __device__ __forceinline__ unsigned lane_off(int lane) const {
        const int j = lane & 15, left = ke - 1 - g0, jc = j < left ? j : left, tt = t0 + jc;
        const unsigned nb = seg >= 16 ? (tt >= seg ? 1u : 0u) : (unsigned)(tt / seg);
        return off0 + (unsigned)jc * ld4 + nb * wrap4;
    }
    __device__ __forceinline__ void advance() {       // to the same rows of the next k-tile
        if (g0 + BK < ke) {
            g0 += BK; t0 += BK; off0 += BK * ld4;
            while (t0 >= seg) { t0 -= seg; off0 += wrap4; }
        }
    }
};
__device__ __forceinline__ unsigned lane_bcast(unsigned v, int j) {
#ifdef VAME_EMU
    return __shfl(v, j);
#else
    return (unsigned)__builtin_amdgcn_readlane((int)v, j);
#endif
}

struct TileRegs { f32x2 r[16]; };          // a producer lane's share of one k-tile: 16 rows x 2 adjacent columns

// rows gt .. gt+15 of the operand (gt = the true first row: rows >= ke read as zero), then the cursor moves on to the next k-tile.
// voff: the lane's column offset in bytes, poisoned (out of the buffer range: reads 0) for columns outside the matrix.
__device__ __forceinline__ void prod_load(TileRegs& R, RowCursor& c, BufRange rng, unsigned voff, int lane, int gt) {
    const unsigned lo = c.lane_off(lane);
#pragma unroll
    for (int j = 0; j < 16; ++j) R.r[j] = buf_load_f32x2(rng, voff, lane_bcast(lo, j));
    c.advance();
    if (gt + 16 > c.ke) {                   // the slab's last, partial k-tile (wave-uniform, once per workgroup at most)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (gt + j >= c.ke) R.r[j] = f32x2{0.f, 0.f};
    }
}

__device__ __forceinline__ f32x2 hi16_of(f32x2 v) { return f32x2{ufloat(fbits(v[0]) & HI16), ufloat(fbits(v[1]) & HI16)}; }

// split the 32 values into their three bf16 planes and store them as MFMA operands: dst = this wave's first octet region of the image
__device__ __forceinline__ void prod_split(const TileRegs& R, char* dst, int slot_e, int slot_o, bool nostore = false) {
#pragma unroll
    for (int oc = 0; oc < 2; ++oc) {
        u32x4 p1[2], p2[2], p3[2];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const f32x2 a = R.r[8 * oc + 2 * d], b = R.r[8 * oc + 2 * d + 1];      // rows 2d, 2d+1 of the octet; [0], [1] = the lane's two columns
            const f32x2 ra = a - hi16_of(a), rb = b - hi16_of(b);                  // exact
            const f32x2 sa = ra - hi16_of(ra), sb = rb - hi16_of(rb);              // exact, <= 8 significant bits
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                p1[e][d] = pack_hi16(a[e], b[e]);
                p2[e][d] = pack_hi16(ra[e], rb[e]);
                p3[e][d] = pack_hi16(sa[e], sb[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            char* q = dst + (e ? slot_o : slot_e) + oc * REGION;
#if defined(VAME_TUNING_BUILD) && !defined(VAME_EMU)
            if (nostore) { asm volatile("" ::"v"(p1[e]), "v"(p2[e]), "v"(p3[e])); continue; }
#endif
            *reinterpret_cast<u32x4*>(q) = p1[e];
            *reinterpret_cast<u32x4*>(q + PLANE) = p2[e];
            *reinterpret_cast<u32x4*>(q + 2 * PLANE) = p3[e];
        }
    }
}

// ---- producer of a ROW-major operand (image column = operand row, K contiguous: activations x weights^T and activations x weights).
// A lane owns four (row, k-octet) units of the k-tile: row = 16 u + (lane >> 2) of its wave's 64, octet = lane & 3 -- four lanes read one
// 128-byte line of a row -- each unit 2 x 16 B = the eight consecutive k of one MFMA operand.  Rows never change along K: their (two-level)
// offsets are resolved once, relative to the tile's first row (the buffer range's base; rows outside the matrix are poisoned), and the
// k-tile is the loads' scalar offset.  In the image the four octets of one row would sit 2 KB apart = on the same banks, so units of this
// producer are stored at column slot ^ (octet << 1): the 16 lanes of a store pass (4 rows x 4 octets) then cover all 64 banks once, and a
// fragment read (one octet, 32 columns) sees a fixed permutation of its conflict-free slots.
struct RowProducer {
    unsigned voff[4], soff;
    int slot0;                      // image offset of unit 0; unit u sits at slot0 ^ ((u & 1) << 7 | (u >> 1) << 8) (16 rows on = bits 3, 4 of the column's slot index)
    __device__ __forceinline__ void init(const GemmOperand& o, int x0, int X, int op, int h, int lane) {
        const int xr = lane >> 2, oct = lane & 3;
        const int64_t first = op_row(o, x0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = 64 * h + 16 * u + xr, gx = x0 + x;
            voff[u] = gx < X ? (unsigned)((op_row(o, gx) - first) * 4 + oct * 32) : 0xC0000000u;
        }
        slot0 = (op * 4 + oct) * REGION + (xslot(64 * h + xr) ^ (oct << 5));
        soff = 0;
    }
    __device__ __forceinline__ void load(TileRegs& R, BufRange rng) {          // unit u -> R.r[4u .. 4u+3] = (k0,k1) (k2,k3) (k4,k5) (k6,k7)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 q0 = buf_load_f32x4(rng, voff[u], soff), q1 = buf_load_f32x4(rng, voff[u] + 16u, soff);
            R.r[4 * u] = f32x2{q0.x, q0.y}; R.r[4 * u + 1] = f32x2{q0.z, q0.w};
            R.r[4 * u + 2] = f32x2{q1.x, q1.y}; R.r[4 * u + 3] = f32x2{q1.z, q1.w};
        }
        soff += BK * 4;
    }
    __device__ __forceinline__ void split(const TileRegs& R, char* img) const {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 p1, p2, p3;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f32x2 a = R.r[4 * u + d];
                const f32x2 ra = a - hi16_of(a);                               // exact
                const f32x2 sa = ra - hi16_of(ra);                             // exact, <= 8 significant bits
                p1[d] = pack_hi16(a[0], a[1]);
                p2[d] = pack_hi16(ra[0], ra[1]);
                p3[d] = pack_hi16(sa[0], sa[1]);
            }
            char* q = img + (slot0 ^ (((u & 1) << 7) | ((u >> 1) << 8)));
            *reinterpret_cast<u32x4*>(q) = p1;
            *reinterpret_cast<u32x4*>(q + PLANE) = p2;
            *reinterpret_cast<u32x4*>(q + 2 * PLANE) = p3;
        }
    }
};

struct Frags { u32x4 a[3][2], b[3][2]; };   // [plane][MFMA tile] operands of one K = 16 step

__device__ __forceinline__ void cons_read(Frags& f, const char* img, int step, const int (&oa)[2], const int (&ob)[2]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.a[pl][i] = *reinterpret_cast<const u32x4*>(img + pl * PLANE + 2 * step * REGION + oa[i]);
            f.b[pl][i] = *reinterpret_cast<const u32x4*>(img + pl * PLANE + 2 * step * REGION + ob[i]);
        }
}

// the six plane products of one K = 16 step on the wave's 2 x 2 tiles; consecutive MFMAs go to different accumulators
template <int NACC>
__device__ __forceinline__ void cons_mfma(const Frags& f, f32x16 (&acc)[NACC][2][2]) {
    constexpr int LO = NACC - 1;
    constexpr int PA[6] = {0, 0, 2, 1, 0, 1}, PB[6] = {0, 2, 0, 1, 1, 0};      // (1,1) | (1,3) (3,1) (2,2) (1,2) (2,1)
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16& c = acc[q == 0 ? 0 : LO][i][j];
                c = MFMA_BF16_32x32x16(f.a[PA[q]][i], f.b[PB[q]][j], c);
            }
}
}  // namespace split6

#ifdef VAME_TUNING_BUILD      // timing-only ablations (tools/split_abl.py; results are garbage): opt bits 8.. = 1 no fragment reads / MFMAs, 2 fragment
#define SPLIT_ABL_MASK 0x3f00      // reads but no MFMAs, 4 no split / LDS stores, 8 no global loads, 16 split but no LDS stores, 32 the loads as dwordx4
#define SPLIT_ABL(opt) (((opt) & SPLIT_ABL_MASK) >> 8)
#define SPLIT_VARIANT_MASK 0x9c   // bits 2, 3, 7 (+ 4): the other mappings (tools/gemm_split_variants.inc)
#else
#define SPLIT_ABL_MASK 0
#define SPLIT_ABL(opt) 0
#define SPLIT_VARIANT_MASK 0
#endif
// opt: see vame_gemm_group_bf16x6_f32.  OCC = workgroups per CU the register budget is set for (3 with one accumulator set, 2 with two).
template <int NACC, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_split_kernel(GemmParams p, int opt) {
    using namespace split6;
    __shared__ __attribute__((aligned(16))) char smem[BUF];
    int tm_, tn_, z, g;
    if (!map_tile(p, tm_, tn_, z, g)) return;
    GemmOperand opA = p.A, opB = p.B;
    float* ws = p.ws;
    if (p.group > 1) { opA.p = p.gA[g]; opB.p = p.gB[g]; ws += (int64_t)g * p.splitk * p.M * p.N; }
    const int tid = threadIdx.x, lane = tid & 63, wv = uniform(tid >> 6);
    const int m0 = tm_ * TILE, n0 = tn_ * TILE;
    const int kb = z * p.kper, ke = (kb + p.kper < p.K) ? kb + p.kper : p.K;
    const int nt = (ke - kb + BK - 1) / BK;
    // producer role
    const int op = wv >> 1, h = wv & 1;
    const GemmOperand& o = op ? opB : opA;
    const int x0 = op ? n0 : m0, X = op ? p.N : p.M;
    const int x = x0 + 2 * lane;
    const unsigned voff = x < X ? (unsigned)(x + (x >= o.gap_at ? o.gap : 0)) * 4u : 0xC0000000u;
    const BufRange rng = buf_range(o.p + op_row(o, kb), (uint64_t)(op_row(o, ke - 1) - op_row(o, kb) + o.ld) * 4);
    RowCursor c;
    c.init(o, kb, kb + 16 * h, ke);
    int gt = kb + 16 * h;
    char* dst = smem + (op * 4 + 2 * h) * REGION;
    const int se = lane * 16, so = 1024 + ((lane ^ 8) * 16);
    // consumer role
    const int li = lane & 31, hh = lane >> 5, wm = wv >> 1, wn = wv & 1;
    int oa[2], ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        oa[i] = hh * REGION + xslot(wm * 64 + i * 32 + li);
        ob[i] = (4 + hh) * REGION + xslot(wn * 64 + i * 32 + li);
    }
    f32x16 acc[NACC][2][2];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
    TileRegs R;
    const int abl = SPLIT_ABL(opt);
    auto LD = [&]() {
#if defined(VAME_TUNING_BUILD) && !defined(VAME_EMU)
        if (abl & 32) {      // load-rate experiment: the same bytes as 8 x dwordx4 (garbage layout)
            const unsigned lo = c.lane_off(lane), v4 = (unsigned)((x0 + 4 * (lane & 31)) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 q = buf_load_f32x4(rng, v4, lane_bcast(lo, 2 * j) + (lane >> 5) * c.ld4);
                R.r[2 * j] = f32x2{q.x, q.y}; R.r[2 * j + 1] = f32x2{q.z, q.w};
            }
            c.advance();
            gt += BK;
            return;
        }
#endif
        if (!(abl & 8)) prod_load(R, c, rng, voff, lane, gt);
        gt += BK;
    };
    LD();
    for (int it = 0; it < nt; ++it) {
        if (!(abl & 4)) prod_split(R, dst, se, so, (abl & 16) != 0);
        LD();                                           // tile it+1 (behind the slab's end: its last rows again, never used)
        LDS_BARRIER();
        if (!(abl & 1)) {
            Frags f;
            cons_read(f, smem, 0, oa, ob);
            if (!(abl & 2)) cons_mfma<NACC>(f, acc);
#if defined(VAME_TUNING_BUILD) && !defined(VAME_EMU)
            else asm volatile("" ::"v"(f.a[0][0]), "v"(f.b[2][1]), "v"(f.a[2][1]), "v"(f.b[0][0]));
#endif
            cons_read(f, smem, 1, oa, ob);
            if (!(abl & 2)) cons_mfma<NACC>(f, acc);
#if defined(VAME_TUNING_BUILD) && !defined(VAME_EMU)
            else asm volatile("" ::"v"(f.a[0][0]), "v"(f.b[2][1]), "v"(f.a[2][1]), "v"(f.b[0][0]));
#endif
        }
        LDS_BARRIER();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + frag_row(r, lane);
                if (row >= p.M) continue;
                float v = acc[0][i][j][r];
                if (NACC > 1) v += acc[NACC - 1][i][j][r];
                ws[((int64_t)z * p.M + row) * p.N + col] = v;
            }
        }
}

// The same contraction for a ROW-major A (activations: M = batch x time rows with two-level addressing, K contiguous) and a row-major
// (BKM = false: C = A B^T, the input projections) or k-major (BKM = true: C = A B, the data gradients) B = a weight matrix.  K is short
// here (the layer width): no split-K, the workgroup writes C itself (+ bias, + C when accumulating).  Waves 0, 1 split A's 128 rows, waves
// 2, 3 B's (RowProducer, or the k-major producer above for BKM); the consumer side is gemm_split_kernel's.
template <bool BKM, int NACC, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_split_rows_kernel(GemmParams p) {
    using namespace split6;
    __shared__ __attribute__((aligned(16))) char smem[BUF];
    int tm_, tn_, z, g;
    if (!map_tile(p, tm_, tn_, z, g)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = uniform(tid >> 6);
    const int m0 = tm_ * TILE, n0 = tn_ * TILE;
    const int nt = p.K / BK;
    const int op = wv >> 1, h = wv & 1;
    const bool kmaj = BKM && op;                                     // this wave splits the k-major B
    const GemmOperand& o = op ? p.B : p.A;
    const int x0 = op ? n0 : m0, X = op ? p.N : p.M;
    RowProducer rp;
    RowCursor c;
    BufRange rng;
    unsigned voff = 0;
    int gt = 16 * h;
    if (kmaj) {
        const int x = x0 + 2 * lane;
        voff = x < X ? (unsigned)x * 4u : 0xC0000000u;
        rng = buf_range(o.p, (uint64_t)p.K * o.ld * 4);
        c.init(o, 0, 16 * h, p.K);
    } else {
        const int xl = x0 + TILE - 1 < X ? x0 + TILE - 1 : X - 1;
        rng = buf_range(o.p + op_row(o, x0), (uint64_t)(op_row(o, xl) - op_row(o, x0) + p.K) * 4);
        rp.init(o, x0, X, op, h, lane);
    }
    const int li = lane & 31, hh = lane >> 5, wm = wv >> 1, wn = wv & 1;
    int oa[2][2], ob[2][2];                                          // [K = 16 step][MFMA tile]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rot = (2 * s + hh) << 5;
            oa[s][i] = hh * REGION + (xslot(wm * 64 + i * 32 + li) ^ rot);
            ob[s][i] = (4 + hh) * REGION + (xslot(wn * 64 + i * 32 + li) ^ (BKM ? 0 : rot));
        }
    f32x16 acc[NACC][2][2];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
    TileRegs R;
    auto LD = [&]() {
        if (kmaj) { prod_load(R, c, rng, voff, lane, gt); gt += BK; }
        else rp.load(R, rng);
    };
    LD();
    for (int it = 0; it < nt; ++it) {
        if (kmaj) prod_split(R, smem + (4 + 2 * h) * REGION, lane * 16, 1024 + ((lane ^ 8) * 16));    // (recomputed per k-tile: three registers fewer)
        else rp.split(R, smem);
        if (it + 1 < nt) LD();
        LDS_BARRIER();
        Frags f;
        cons_read(f, smem, 0, oa[0], ob[0]);
        cons_mfma<NACC>(f, acc);
        cons_read(f, smem, 1, oa[1], ob[1]);
        cons_mfma<NACC>(f, acc);
        LDS_BARRIER();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + frag_row(r, lane);
                if (row >= p.M) continue;
                float v = acc[0][i][j][r];
                if (NACC > 1) v += acc[NACC - 1][i][j][r];
                float* cp = p.C + (int64_t)row * p.ldc + col;
                v += bv;
                if (p.accumulate) v += *cp;
                *cp = v;
            }
        }
}

#ifdef VAME_TUNING_BUILD
#include "../../tools/gemm_split_variants.inc"
#endif

// vame_gemm_f32's contract for a row-major A (a_kmajor = 0) and a plain weight matrix B, no split-K, evaluated by gemm_split_rows_kernel: the
// input projections (b_kmajor = 0) and data gradients (b_kmajor = 1) whose K is a layer width.  opt as in vame_gemm_group_bf16x6_f32 (bits 0-1).
extern "C" int vame_gemm_bf16x6_f32(int M, int N, int K, const float* A, int64_t lda, int64_t a_seg, int64_t a_seg_stride, const float* B, int64_t ldb,
                                    int b_kmajor, const float* bias, float* C, int64_t ldc, int accumulate, int opt, void* stream) {
    VAME_CHECK_ARG(A && B && C, VAME_E_BADARG, "gemm_bf16x6: null operand");
    VAME_CHECK_ARG(M >= 1 && N >= 2 && K >= 32 && K % 32 == 0, VAME_E_SHAPE, "gemm_bf16x6: M=%d N=%d K=%d (K a multiple of 32)", M, N, K);
    VAME_CHECK_ARG(lda % 4 == 0 && a_seg_stride % 4 == 0 && (uintptr_t)A % 16 == 0 && a_seg >= 0 && a_seg < (1ll << 31) && lda >= K,
                   VAME_E_BADARG, "gemm_bf16x6: A is read with 16-byte loads (pitch, segment stride multiples of 4 floats, 16-byte aligned)");
    if (b_kmajor)
        VAME_CHECK_ARG(ldb % 2 == 0 && N % 2 == 0 && (uintptr_t)B % 8 == 0 && ldb >= N && (int64_t)K * ldb * 4 < (1ll << 30), VAME_E_BADARG,
                       "gemm_bf16x6: a k-major B is read with 8-byte loads (even pitch and N, 8-byte aligned, < 1 GiB)");
    else
        VAME_CHECK_ARG(ldb % 4 == 0 && (uintptr_t)B % 16 == 0 && ldb >= K && 128 * ldb * 4 < (1ll << 30), VAME_E_BADARG,
                       "gemm_bf16x6: a row-major B is read with 16-byte loads (pitch a multiple of 4 floats, 16-byte aligned)");
    const int nacc = (opt & 3) == 0 ? 2 : (opt & 3);
    VAME_CHECK_ARG((nacc == 1 || nacc == 2) && (opt & ~3) == 0, VAME_E_BADARG, "gemm_bf16x6: opt=%d", opt);
    {       // the 128 rows of a tile are addressed with 32-bit byte offsets from its first row, rows outside the matrix at 3 GiB
        const int64_t wrap = a_seg ? a_seg_stride - a_seg * lda : 0;
        VAME_CHECK_ARG(wrap >= 0 && (128 * lda + (a_seg ? 128 / a_seg + 2 : 0) * wrap + K) * 4 < (1ll << 30), VAME_E_UNSUPPORTED,
                       "gemm_bf16x6: 128 rows of A must span < 1 GiB with non-negative strides");
    }
    GemmParams p;
    p.A = {A, lda, a_seg, a_seg_stride, 1, 0x7fffffff, 0};
    p.B = {B, ldb, 0, 0, 1, 0x7fffffff, 0};
    for (int g = 0; g < 8; ++g) { p.gA[g] = A; p.gB[g] = B; }
    p.bias = bias; p.C = C; p.ldc = ldc; p.ws = nullptr;
    p.M = M; p.N = N; p.K = K; p.accumulate = accumulate; p.group = 1;
    p.kper = K; p.splitk = 1;
    p.tiles_m = (int)cdiv64(M, split6::TILE); p.tiles_n = (int)cdiv64(N, split6::TILE);
    p.by_z = 0; p.cvec = 0; p.wsvec = 0;
    dim3 grid((unsigned)(cdiv64(p.tiles_m, 8) * p.tiles_n * 8)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (b_kmajor) {
        if (nacc == 2) hipLaunchKernelGGL((gemm_split_rows_kernel<true, 2, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_split_rows_kernel<true, 1, 3>), grid, block, 0, st, p);
    } else {
        if (nacc == 2) hipLaunchKernelGGL((gemm_split_rows_kernel<false, 2, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_split_rows_kernel<false, 1, 3>), grid, block, 0, st, p);
    }
    VAME_LAUNCH_CHECK("gemm_bf16x6");
    return VAME_OK;
}

// vame_gemm_group_f32's contract for two k-major operands, evaluated by gemm_split_kernel.  opt: bits 0-1 = accumulators per output (0 = default:
// 2 -- the leading plane product on its own, two workgroups per CU; 1 = one accumulator for all six products, three workgroups per CU).
// (Tuning build only: bits 2 / 3 / 7 = the mappings of tools/gemm_split_variants.inc, bits 8-13 = timing ablations.)
extern "C" int vame_gemm_group_bf16x6_f32(int count, int M, int N, int K, const float* const* A, int64_t lda, int64_t a_seg, int64_t a_seg_stride,
                                          const float* const* B, int64_t ldb, int64_t b_seg, int64_t b_seg_stride, float* const* C, int64_t ldc,
                                          int accumulate, int splitk, float* ws, int a_gap_at, int a_gap, int opt, void* stream) {
    VAME_CHECK_ARG(count >= 1 && count <= 8 && A && B && C && ws, VAME_E_BADARG, "gemm_group_bf16x6: count=%d (1..8) / null table", count);
    VAME_CHECK_ARG(M >= 2 && N >= 2 && K >= 1 && M % 2 == 0 && N % 2 == 0, VAME_E_SHAPE, "gemm_group_bf16x6: M=%d N=%d K=%d (M, N even)", M, N, K);
    VAME_CHECK_ARG(lda % 2 == 0 && ldb % 2 == 0 && a_seg_stride % 2 == 0 && b_seg_stride % 2 == 0 && a_gap % 2 == 0 && a_gap_at % 2 == 0 &&
                   a_seg >= 0 && b_seg >= 0 && a_seg < (1ll << 31) && b_seg < (1ll << 31),
                   VAME_E_BADARG, "gemm_group_bf16x6: row pitches, segment strides and the column gap must be even (8-byte operand loads)");
    const int nacc = (opt & 3) == 0 ? 2 : (opt & 3);
    VAME_CHECK_ARG((nacc == 1 || nacc == 2) && (opt & ~(3 | SPLIT_VARIANT_MASK | SPLIT_ABL_MASK)) == 0, VAME_E_BADARG, "gemm_group_bf16x6: opt=%d", opt);
    GemmParams p;
    for (int g = 0; g < count; ++g) {
        VAME_CHECK_ARG(A[g] && B[g] && C[g], VAME_E_BADARG, "gemm_group_bf16x6: problem %d has a null operand", g);
        VAME_CHECK_ARG((uintptr_t)A[g] % 8 == 0 && (uintptr_t)B[g] % 8 == 0, VAME_E_BADARG, "gemm_group_bf16x6: operands must be 8-byte aligned");
        p.gA[g] = A[g]; p.gB[g] = B[g];
    }
    for (int g = count; g < 8; ++g) { p.gA[g] = A[0]; p.gB[g] = B[0]; }
    p.A = {A[0], lda, a_seg, a_seg_stride, 1, a_gap ? a_gap_at : 0x7fffffff, a_gap};
    p.B = {B[0], ldb, b_seg, b_seg_stride, 1, 0x7fffffff, 0};
    p.bias = nullptr; p.C = C[0]; p.ldc = ldc; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.accumulate = accumulate; p.group = count;
    p.kper = (int)(cdiv64(cdiv64(K, splitk), 32) * 32);
    p.splitk = (int)cdiv64(K, p.kper);
    VAME_CHECK_ARG(p.splitk >= 8, VAME_E_SHAPE, "gemm_group_bf16x6: split-K %d < 8 (grouping is for large-K contractions)", p.splitk);
    for (const GemmOperand* o : {&p.A, &p.B}) {      // a slab's rows are addressed with 32-bit byte offsets from its first row, lanes outside the matrix at 3 GiB
        const int64_t wrap = o->seg ? o->seg_stride - o->seg * o->ld : 0, rows = p.kper + 64;
        VAME_CHECK_ARG(o->ld > 0 && wrap >= 0 && (rows * o->ld + (o->seg ? rows / o->seg + 2 : 0) * wrap) * 4 < (1ll << 30), VAME_E_UNSUPPORTED,
                       "gemm_group_bf16x6: a k-slab must span < 1 GiB with non-negative strides (raise splitk)");
    }
    p.tiles_m = (int)cdiv64(M, split6::TILE); p.tiles_n = (int)cdiv64(N, split6::TILE);
    p.by_z = 1; p.cvec = 0; p.wsvec = 0;
    const int64_t per_unit = (int64_t)p.tiles_m * p.tiles_n, nunits = (int64_t)p.splitk * p.group;
    dim3 grid((unsigned)(cdiv64(nunits, 8) * per_unit * 8)), block(256);
    hipStream_t st = (hipStream_t)stream;
#ifdef VAME_TUNING_BUILD
    if (opt & SPLIT_VARIANT_MASK) {
        if (opt & 4) hipLaunchKernelGGL(gemm_split12_kernel, grid, dim3(768), 0, st, p, opt);
        else if (opt & 128) {
            p.tiles_m = (int)cdiv64(M, 2 * split6::TILE);
            grid = dim3((unsigned)(cdiv64(nunits, 8) * p.tiles_m * p.tiles_n * 8));
            if (opt & 16) hipLaunchKernelGGL(gemm_split_wide_il_kernel, grid, dim3(512), 0, st, p, opt);
            else if (nacc == 2) hipLaunchKernelGGL(gemm_split_wide_kernel<2>, grid, dim3(512), 0, st, p, opt);
            else hipLaunchKernelGGL(gemm_split_wide_kernel<1>, grid, dim3(512), 0, st, p, opt);
        } else if (nacc == 2) hipLaunchKernelGGL(gemm_split8_kernel<2>, grid, dim3(512), 0, st, p, opt);
        else hipLaunchKernelGGL(gemm_split8_kernel<1>, grid, dim3(512), 0, st, p, opt);
    } else
#endif
    if (nacc == 2) hipLaunchKernelGGL((gemm_split_kernel<2, 2>), grid, block, 0, st, p, opt);
    else hipLaunchKernelGGL((gemm_split_kernel<1, 3>), grid, block, 0, st, p, opt);
    VAME_LAUNCH_CHECK("gemm_group_bf16x6");
    GemmGroupOut out;
    bool one_c = count > 1;
    for (int g = 0; g < 8; ++g) { out.C[g] = C[g < count ? g : 0]; one_c = one_c && out.C[g] == C[0]; }
    const int64_t n = (int64_t)M * N;
    if (one_c) {
        const int slabs = count * p.splitk;
        if (reduce_spread(n, slabs))
            hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(reduce_blocks(n, true)), dim3(256), 0, st, (const float*)ws, slabs, M, N,
                               (const float*)nullptr, C[0], ldc, accumulate);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(reduce_blocks(n, false)), dim3(256), 0, st, (const float*)ws, slabs, M, N,
                               (const float*)nullptr, C[0], ldc, accumulate);
    } else if (reduce_spread(n, p.splitk)) {
        hipLaunchKernelGGL(splitk_reduce_group_kernel<true>, dim3(reduce_blocks(n, true), count), dim3(256), 0, st, (const float*)ws, p.splitk, M, N,
                           out, ldc, accumulate);
    } else {
        hipLaunchKernelGGL(splitk_reduce_group_kernel<false>, dim3(reduce_blocks(n, false), count), dim3(256), 0, st, (const float*)ws, p.splitk, M, N,
                           out, ldc, accumulate);
    }
    VAME_LAUNCH_CHECK("gemm_group_bf16x6 reduce");
    return VAME_OK;
}
