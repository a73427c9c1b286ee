// Column-split ("cooperative") GRU sequence kernels for SMALL batches.
//
// gru_seq.hip gives one workgroup a whole 32-row batch tile: at batch 256 that is 8 tiles x 2 directions = 16 workgroups on a
// 256-CU chip, and a time step costs what one CU needs for 32 x 3H x H MACs (~26 us at H = 256) no matter how idle the rest
// is.  Here the R rows of a tile (R = 32, or 16 where twice the workgroups still get a CU each) are shared by S = H/32 workgroups
// (a "group", all on one XCD): member s owns hidden columns [32s, 32s+32) of all three gates and keeps ITS slice of W_hh in LDS for
// the whole sequence (98-102 KB at H = 256, nothing is re-streamed).  The contraction runs on 16 x 16 x 4 MFMA tiles over eight waves
// (two per SIMD), which makes gate math, publish and the BPTT stash lane-local.  After every step the members hand each other their
// slices of h_t as self-validating (value, tag) pairs (forward: every consumer thread polls the data itself, no flag, no drain) or
// reduce-scatter their partial dh through a double-buffered scratch behind a drained flag (BPTT); loads and stores of the hand-off carry
// sc1 (L1 bypassed, L2-served inside the XCD: MI355X_MICROARCH.md, hand-off price list).  I/O contract (descriptor table, stash layout,
// padded sequence layout) is identical to vame_gru_seq_fwd_f32 / _bwd_f32; results agree to summation-order rounding (K is summed in
// quarters / by member) and are the same bits for every form of a launch (16- / 32-row groups, row-range launches).
//
// Residency: the spin wait needs every member of a group running.  The launcher refuses grids above one workgroup per CU
// (<= 256 workgroups, LDS forces 1 per CU), every poll loop is bounded and reports through `status` instead of hanging.
#include "vame_common.h"
#include "gru_desc.h"

#ifdef VAME_EMU
#include <chrono>
#include <thread>
#define COOP_STORE16(ptr, v) (*reinterpret_cast<f32x4*>(ptr) = (v))
#define COOP_STORE4(ptr, v) (*(ptr) = (v))
#define COOP_LOAD16(dst, ptr) ((dst) = *reinterpret_cast<const f32x4*>(ptr))
#define COOP_LOAD4(dst, ptr) ((dst) = *(ptr))
// (value, tag) pairs: values first, then the tags with release order; the reader takes the tags first (acquire), then the values
static inline void emu_store_ll(float* p, float a, float b, unsigned tag) {
    p[0] = a; p[2] = b;
    __atomic_store_n(reinterpret_cast<unsigned*>(p) + 1, tag, __ATOMIC_RELEASE);
    __atomic_store_n(reinterpret_cast<unsigned*>(p) + 3, tag, __ATOMIC_RELEASE);
}
static inline f32x4 emu_load_ll(const float* p) {
    const unsigned t1 = __atomic_load_n(reinterpret_cast<const unsigned*>(p) + 1, __ATOMIC_ACQUIRE);
    const unsigned t3 = __atomic_load_n(reinterpret_cast<const unsigned*>(p) + 3, __ATOMIC_ACQUIRE);
    f32x4 v; float f1, f3;
    __builtin_memcpy(&f1, &t1, 4); __builtin_memcpy(&f3, &t3, 4);
    v[0] = p[0]; v[1] = f1; v[2] = p[2]; v[3] = f3;
    return v;
}
#define COOP_STORE16_LL(ptr, a, b, tag) emu_store_ll((ptr), (a), (b), (tag))
#define COOP_LOAD16_LL(dst, ptr) ((dst) = emu_load_ll(ptr))
#define COOP_WAIT_LL8(v)
#define COOP_WAIT_LL16(v)
static inline unsigned emu_tag(float x) { unsigned u; __builtin_memcpy(&u, &x, 4); return u; }
#define COOP_TAG(x) emu_tag(x)
#define COOP_WAIT_LOADS8(a, b, c, d, e, f, g, h)
#define COOP_DRAIN()
#define COOP_WAVES_PER_SIMD
#define COOP_MFMA_SETTLE()
#define COOP_FLAG_STORE(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define COOP_FLAG_LOAD(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
/* one OS thread per workgroup, one fiber per thread: let the workgroup's other fibers run (they may be the producers), and the other workgroups */
#define COOP_BACKOFF() do { emu::yield(); if (threadIdx.x == 0) std::this_thread::sleep_for(std::chrono::microseconds(20)); } while (0)
#else
// 16-byte write-through store / L1-bypassing load (sc1); asm because HIP has no 16-byte agent-scope access
#define COOP_STORE16(ptr, v) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(ptr), "v"(v) : "memory")
#define COOP_STORE4(ptr, v) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(ptr), "v"(v) : "memory")
#define COOP_LOAD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory")
#define COOP_LOAD4(dst, ptr) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(dst) : "v"(ptr) : "memory")
#define COOP_WAIT_LOADS8(a, b, c, d, e, f, g, h) \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
// tagged hand-off: (value, tag, value, tag) in one 16-byte write-through store; the s_nop keeps a VALU write off the data registers
// until the store has read them (the recogniser does not see into the asm).
// HARDWARE ASSUMPTION (gfx950, the only target of this file -- enforced below): a naturally aligned 8-byte (value, tag) pair written by ONE
// store instruction is never observed torn by a load that covers it -- each pair lies in one 16-byte, hence one 64-byte, L2 granule, is
// written by a single write-through (sc1) request and read by single L1-bypassing (sc1) requests; the reader accepts a pair only when ITS
// OWN tag matches, so the two pairs of a 16-byte packet may arrive at different times and nothing depends on ordering between packets, on
// which XCD a producer or consumer runs (b % 8 -> XCD is a speed heuristic only: a group spread over XCDs exchanges through the shared
// memory side instead of one L2, ~1.7x slower hops, same protocol) or on release / acquire semantics (MI355X_MICROARCH.md, "handoff-1to1":
// granule = one naturally aligned 8-byte {data, tag} written by ONE sc1 store).  tools/coop_stress.py and tests/test_kernels_gpu.py::
// test_gru_coop_* compare every VALUE that travelled through the packets with the batch-tile-persistent kernels' over repeated launches.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "gru_coop.hip's tagged hand-off relies on gfx950 memory-system behaviour (see above); it has not been validated anywhere else"
#endif
#define COOP_STORE16_LL(ptr, a, b, tag)                                                                          \
    do {                                                                                                         \
        const f32x4 ll_ = {(a), __uint_as_float(tag), (b), __uint_as_float(tag)};                                \
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" : : "v"(ptr), "v"(ll_) : "memory");        \
    } while (0)
#define COOP_LOAD16_LL(dst, ptr) COOP_LOAD16(dst, ptr)
#define COOP_WAIT_LL8(v) COOP_WAIT_LOADS8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
#define COOP_WAIT_LL16(v)                                                                                        \
    do {                                                                                                         \
        COOP_WAIT_LOADS8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);                                        \
        asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])); \
    } while (0)
#define COOP_TAG(x) __float_as_uint(x)
#define COOP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// LDS allows one workgroup per CU = two waves per SIMD: the register allocator may use 256 registers per wave instead of spilling at 128
#define COOP_WAVES_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 2)))
// MFMA results read by an asm store: the hazard recognizer does not look inside inline asm, so the wait states an 8-pass MFMA needs
// before a VMEM instruction may read its destination (11) are spelled out
#define COOP_MFMA_SETTLE() asm volatile("s_nop 15\n s_nop 3" ::: "memory")
#define COOP_FLAG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define COOP_FLAG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define COOP_BACKOFF() __builtin_amdgcn_s_sleep(2)
#endif

#if defined(VAME_PROBE) && !defined(VAME_EMU)   // tuning build (make probe): per-wave phase cycle sums, tools/coop_probe.py
__device__ long long* g_coop_probe;
extern "C" int vame_probe_set_coop(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_coop_probe), &p, sizeof(p)); }
#define COOP_PHASE_DECL() long long pp_[8] = {0}, pa_ = (long long)__builtin_amdgcn_s_memtime()
#define COOP_PHASE(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); pp_[i] += t_ - pa_; pa_ = t_; } while (0)
#define COOP_PHASE_END()                                                                            \
    if ((threadIdx.x & 63) == 0 && g_coop_probe) {                                                   \
        long long* o_ = g_coop_probe + ((long long)blockIdx.x * (COOP_NT / 64) + (threadIdx.x >> 6)) * 8;         \
        for (int i_ = 0; i_ < 8; ++i_) o_[i_] = pp_[i_];                                             \
    }
#else
#define COOP_PHASE_DECL()
#define COOP_PHASE(i)
#define COOP_PHASE_END()
#endif

// The launch epoch lives on the DEVICE: epoch[0] = the tag base of this launch (every hand-off tag / flag value of the launch is base + step
// + 1 ... compared modulo 2^32), epoch[1] = a ticket.  Every workgroup reads the base when it starts; the last one to finish moves it on by
// `advance` (> the launch's steps + 1), so the next launch sharing `flags` -- also a REPLAY of a captured graph, whose arguments are
// frozen -- finds every word of this one stale.  No host counter, no argument that changes from launch to launch.
__device__ __forceinline__ void coop_advance_epoch(int* epoch, int base, int advance) {
    if (threadIdx.x == 0 && atomicAdd(epoch + 1, 1) == (int)gridDim.x - 1) {
        epoch[1] = 0;
        epoch[0] = (int)((unsigned)base + (unsigned)advance);
    }
}

// Poll budget of one hand-off wait.  A stuck group reports through *status instead of hanging the queue; once *status != 0 every
// later wait of the launch gives up at once (results undefined, the launch ends) and the optimizer kernel that follows refuses to
// apply the step (vame_adam_amsgrad_f32's abort_flag).  Device default ~0.3 s; the host emulator (one OS thread per workgroup,
// possibly oversubscribed) waits practically for ever.  vame_gru_coop_set_poll_limit overrides it (diagnostics / fail-fast tests).
#ifdef VAME_EMU
constexpr int COOP_DEFAULT_POLLS = 1 << 30;
#else
constexpr int COOP_DEFAULT_POLLS = 1 << 18;
#endif
static int g_coop_polls = COOP_DEFAULT_POLLS;
constexpr int COOP_NT = 512;             // 8 waves, two per SIMD: an MFMA stream fed from LDS fragments issues every ~34.6 cycles with two waves, 38 with one (r04_mfma_issue_probe.txt)
constexpr int COOP_LL_OFF = 4096;        // (the flag words of a launch: <= 256 workgroups)
extern "C" int vame_gru_coop_set_poll_limit(int polls) {
    const int old = g_coop_polls;
    g_coop_polls = polls != 0 ? polls : COOP_DEFAULT_POLLS;      // < 0: fault injection -- every launch reports one timeout
    return old;
}

// block -> (group, member).  Workgroup b runs on XCD b % 8 (observed; speed only): all members of a group share b % 8.
template <int NM>
__device__ __forceinline__ bool coop_map(int ngroups, int& g, int& m) {
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    m = q % NM;
    g = (q / NM) * 8 + xcd;
    return g < ngroups;
}
template <int NM>
static int coop_grid(int ngroups) { return (int)cdiv64(ngroups, 8) * 8 * NM; }
// ... with NH row halves per 32-row tile, each a group of its own on the same XCD (forward, R = 32 / NH rows per group)
template <int NM, int NH>
__device__ __forceinline__ bool coop_map(int ngroups, int& g, int& m, int& half) {
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    m = q % NM;
    half = (q / NM) % NH;
    g = (q / (NM * NH)) * 8 + xcd;
    return g < ngroups;
}

// Forward.  A group is the R rows of a batch tile (R = 32, or 16 when twice the workgroups still fit one per CU: the launches with two
// streams at batch 256) x the S members.  A member's step is R x 32 columns x 3 gates over K = H, as 16 x 16 tiles of
// v_mfma_f32_16x16x4_f32 on eight waves (two per SIMD): wave w owns the 16 x 16 tile (row half, column half) = ((w >> 1) & 1, w & 1) and
// the K half w >> 2 (as two separate quarters) at R = 32, or column half w & 1 and K quarter w >> 1 at R = 16; the quarters meet through LDS
// in the tile's first wave and are added as (q0 + q1) + (q2 + q3) in either form.  All three gates of a tile sit in the same lanes -- lane l holds column l & 15,
// rows 4 (l >> 4) .. +3 -- so the gate math, the publish of h_t and the BPTT stash (one float4 per quantity and lane, the layout of the
// batch-tile kernels) need no exchange between waves, and all four SIMDs share the contraction (192 / 96 MFMAs of 32 cycles per wave
// and step against 128 of 64 cycles on three of four SIMDs in the 32 x 32 form of rounds 2-3).  The four K quarters are summed
// separately and then added in the same order in either form, so a launch gives the same bits whichever R its row range selects.
template <int H, int R>
__device__ __forceinline__ void gru_coop_fwd_body(const GruFwdParams& P, int* __restrict__ flags, int base, int* __restrict__ status, int max_polls) {
    constexpr int NM = H / 32, NH = 32 / R, LDW = H + 4, LDH = H + 4, NCH = H / 16, QN = NCH / 4, PBF = R == 32 ? 4 * 24 * 64 : 2 * 3 * 12 * 64;
    static_assert(R == 32 || R == 16, "row tiles of 32 or 16");
    VAME_DYN_SMEM(smem_raw);
    float* wl = reinterpret_cast<float*>(smem_raw);                        // [3][32][LDW] this member's rows of W_hh, k contiguous
    float* hs = wl + 96 * LDW;                                             // [R][LDH] h_{t-1} (A operand)
    float* pb = hs + R * LDH;                                              // partial sums of the K quarters that are not the tile's first wave's: [tile][24][64] / [column half][3][12][64]
    int g, m, half;
    if (!coop_map<NM, NH>(P.nstreams * P.ntiles, g, m, half)) return;
    const bool inject = max_polls < 0;
    if (inject) max_polls = COOP_DEFAULT_POLLS;
    const int sidx = g % P.nstreams, tile = g / P.nstreams + P.tile_off;
    // a COPY of the stream's descriptor: behind a reference into the kernel arguments every asm statement with a memory clobber makes
    // hipcc re-read the fields it needs next (s_load + wait, ~1000 cycles per step in the probe's publish phase)
    const GruFwdStream S = P.s[sidx];
    const int B = P.B, T = (int)S.T;
    const int row0 = tile * 32 + half * R, col0 = 32 * m;
    const int nvalid = B - row0;
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, kg = lane >> 4;
    const int w = UNIFORM(tid >> 6);
    const int ch = w & 1, rh = R == 32 ? ((w >> 1) & 1) : 0;
    const int q0 = R == 32 ? 2 * (w >> 2) : (w >> 1);                       // first (R = 16: only) K quarter of this wave
    const int lrow = rh * 16 + 4 * kg, lcol = ch * 16 + c16;               // this lane: rows lrow .. lrow + 3, column lcol of the slice
    const bool owner = q0 == 0;                                            // the tile's first wave: finishes the sums, does the gate math, publishes
    const int gidx = g * NH + half;
    float* ll = reinterpret_cast<float*>(flags + COOP_LL_OFF);             // tagged hand-off packets, behind the flag words
    float4* stash = S.stash ? reinterpret_cast<float4*>(S.stash) : nullptr;
    const int rg = (half * R + lrow) >> 2;                                 // 4-row group inside the 32-row stash tile: CR layout (hh, q) = (rg & 1, rg >> 1)
    const int slane = (rg & 1) * 32 + lcol, sq = rg >> 1;
    if (nvalid <= 0) {                                                     // the second half of a last tile with <= 16 rows: nothing to compute, but
        if (owner && stash)                                                // BPTT multiplies the tile's stash rows past the batch by zero -> keep them finite
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int k5 = 0; k5 < 5; ++k5)
                    stash[((((int64_t)tile * T + t) * NM + m) * 20 + k5 * 4 + sq) * 64 + slane] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }

    // ---- prologue: W slice -> LDS (the 32 x 32 x 2 fragment pack un-permuted: fragment (c, gate, lane) = 4 consecutive k of one row)
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(S.wp) + (int64_t)m * (H / 8) * 3 * 64;
        for (int i = tid; i < (H / 8) * 3 * 64; i += COOP_NT) {
            const int c = i / 192, gt = (i % 192) / 64, l = i % 64;
            *reinterpret_cast<f32x4*>(&wl[(gt * 32 + (l & 31)) * LDW + 8 * c + 4 * (l >> 5)]) = src[i];
        }
    }
    for (int i = tid; i < R * (H / 4); i += COOP_NT) {
        const int r = i / (H / 4), c4 = i % (H / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (S.h0 && r < nvalid) v = *reinterpret_cast<const float4*>(S.h0 + (int64_t)(row0 + r) * S.h0_row + 4 * c4);
        *reinterpret_cast<float4*>(&hs[r * LDH + 4 * c4]) = v;
    }
    __syncthreads();
    if (S.pad && tid < R * 8 && (tid >> 3) < nvalid) {                     // padded slot of the sequence: this member's slice of h_0
        const int prow = tid >> 3, pc4 = tid & 7;
        *reinterpret_cast<float4*>(S.y + (int64_t)(row0 + prow) * S.y_row + (int64_t)(S.reverse ? T : -1) * S.y_t + col0 + 4 * pc4) =
            *reinterpret_cast<const float4*>(&hs[prow * LDH + col0 + 4 * pc4]);
    }
    float hprev[4], gcur[3][4], gnext[3][4];
    float bhn = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) hprev[i] = hs[(lrow + i) * LDH + col0 + lcol];
    const float* gi_lane = S.gi + (int64_t)(row0 + lrow) * S.gi_row + col0 + lcol;
    // rows past the batch re-read the group's first row: a load under a per-lane condition makes hipcc wait for every load in flight
    // before the other branch may write the register (probe: ~2000 cycles per step in the BPTT kernel's prefetch); h of such rows is
    // never published or stored, so what they compute from is immaterial
    int64_t gi_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gi_off[i] = (lrow + i < nvalid ? (int64_t)i : -(int64_t)lrow) * S.gi_row;
    auto load_gi = [&](int t, float (&dst)[3][4]) {
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[gt][i] = gi_lane[gi_off[i] + (int64_t)t * S.gi_t + gt * H];
    };
#pragma unroll
    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
        for (int i = 0; i < 4; ++i) { gcur[gt][i] = 0.f; gnext[gt][i] = 0.f; }
    if (owner) { load_gi(S.reverse ? T - 1 : 0, gcur); bhn = S.bhn[col0 + lcol]; }
    // K order of the contraction (any order works, A and B share it): chunk c gives lane group kg the four k of float4 slot
    // (kg & 1) * 16 + (kg >> 1) * KX + c of a row, so the two lane groups that share a ds_read_b128 bank cycle ({0-3,12-15,20-27}, ...:
    // MI355X_MICROARCH.md, LDS) sit 16 slots = one full bank row apart and every group reads 16 distinct 16-byte slots (row stride
    // H + 4 floats = one slot per row); with slot 4c + kg each group had one 2-way conflict per read
    constexpr int KX = NCH == 16 ? 32 : 8;
    static_assert(NCH == 16 || NCH == 8, "K order is written for H = 128 / 256");
    const int kslot = (kg & 1) * 16 + (kg >> 1) * KX;
    const float* arow = &hs[(rh * 16 + c16) * LDH + 4 * kslot];
    const float* brow = &wl[(ch * 16 + c16) * LDW + 4 * kslot];
    float* y_lane = S.y + (int64_t)(row0 + lrow) * S.y_row + col0 + lcol;

    COOP_PHASE_DECL();
    for (int step = 0; step < T; ++step) {
        const int t = S.reverse ? T - 1 - step : step;
        f32x4 lo[3], hi[3];                       // this wave's quarter q0 (and, at R = 32, q0 + 1)
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
#pragma unroll
            for (int i = 0; i < 4; ++i) { lo[gt][i] = owner ? (gt == 2 ? bhn : gcur[gt][i]) : 0.f; hi[gt][i] = 0.f; }
        const bool more = step + 1 < T && S.gi_t != 0;
        if (owner && more) load_gi(S.reverse ? t - 1 : t + 1, gnext);      // lands during the MFMA loop
        const bool recur = !(step == 0 && S.h0 == nullptr);                // zero initial state: no recurrent term in the first step
        if (recur) {
            // the next chunk's fragments are requested before the current chunk's MFMAs (registers double-buffered, order pinned): left to
            // itself hipcc issues every chunk's ds_reads right in front of their first use and the LDS latency is exposed once per chunk
            // (probe: 51 cycles per 32-cycle MFMA).  Element-major issue order: consecutive MFMAs go to different accumulators.
            auto ldf = [](const float* p_) { return *reinterpret_cast<const f32x4*>(p_); };
            const float* ar = arow + 4 * q0 * QN;
            const float* br = brow + 4 * q0 * QN;
            if (R == 32) {
                f32x4 a0 = ldf(ar), a1 = ldf(ar + 4 * QN), b0[3], b1[3];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) { b0[gt] = ldf(br + gt * 32 * LDW); b1[gt] = ldf(br + gt * 32 * LDW + 4 * QN); }
#pragma unroll 2
                for (int c = 0; c < QN; ++c) {
                    const int cn = c + 1 < QN ? c + 1 : c;
                    const f32x4 na0 = ldf(ar + 4 * cn), na1 = ldf(ar + 4 * (cn + QN));
                    f32x4 nb0[3], nb1[3];
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) { nb0[gt] = ldf(br + gt * 32 * LDW + 4 * cn); nb1[gt] = ldf(br + gt * 32 * LDW + 4 * (cn + QN)); }
                    SCHED_FENCE();
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int gt = 0; gt < 3; ++gt) {
                            lo[gt] = MFMA_16x16x4(a0[e], b0[gt][e], lo[gt]);
                            hi[gt] = MFMA_16x16x4(a1[e], b1[gt][e], hi[gt]);
                        }
                    SCHED_FENCE();
                    a0 = na0; a1 = na1;
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) { b0[gt] = nb0[gt]; b1[gt] = nb1[gt]; }
                }
            } else {
                f32x4 a0 = ldf(ar), b0[3];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) b0[gt] = ldf(br + gt * 32 * LDW);
#pragma unroll 2
                for (int c = 0; c < QN; ++c) {
                    const int cn = c + 1 < QN ? c + 1 : c;
                    const f32x4 na0 = ldf(ar + 4 * cn);
                    f32x4 nb0[3];
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) nb0[gt] = ldf(br + gt * 32 * LDW + 4 * cn);
                    SCHED_FENCE();
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int gt = 0; gt < 3; ++gt) lo[gt] = MFMA_16x16x4(a0[e], b0[gt][e], lo[gt]);
                    SCHED_FENCE();
                    a0 = na0;
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) b0[gt] = nb0[gt];
                }
            }
        }
        COOP_PHASE(0);
        // ---- the quarters of a tile meet in its first wave: (q0 + q1) + (q2 + q3)
        f32x4 sum[3];
        {
            if (R == 32) {
                float* pw = pb + ((w & 3) * 24) * 64 + lane;
                if (!owner) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { pw[(gt * 4 + i) * 64] = lo[gt][i]; pw[(12 + gt * 4 + i) * 64] = hi[gt][i]; }
                }
                __syncthreads();
                if (owner) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum[gt][i] = (lo[gt][i] + hi[gt][i]) + (pw[(gt * 4 + i) * 64] + pw[(12 + gt * 4 + i) * 64]);
                }
            } else {
                float* pw = pb + (ch * 3 * 12) * 64 + lane;
                if (!owner) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) pw[((q0 - 1) * 12 + gt * 4 + i) * 64] = lo[gt][i];
                }
                __syncthreads();
                if (owner) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            sum[gt][i] = (lo[gt][i] + pw[(gt * 4 + i) * 64]) + (pw[(12 + gt * 4 + i) * 64] + pw[(24 + gt * 4 + i) * 64]);
                }
            }
        }
        COOP_PHASE(1);
        float ca[4], cb[4], us[4], rs[4], an[4], hnew[4] = {0.f, 0.f, 0.f, 0.f};
        if (owner) {
#pragma clang fp contract(off)      // the two instantiations (R = 32 / 16) must round alike: no fma formed here in one and not the other
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float rr = fast_sigmoid(sum[0][i]), uu = fast_sigmoid(sum[1][i]);
                an[i] = sum[2][i];
                const float nn = fast_tanh(gcur[2][i] + rr * an[i]);
                const float hp = hprev[i];
                const float hv = nn + uu * (hp - nn);
                const float omu = 1.0f - uu;
                ca[i] = omu * (1.0f - nn * nn);
                cb[i] = (hp - nn) * uu * omu;
                us[i] = uu; rs[i] = rr;
                hprev[i] = hv;
                hnew[i] = hv;
            }
            if (more) {
#pragma unroll
                for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) gcur[gt][i] = gnext[gt][i];
            }
        }
        COOP_PHASE(2);
        const unsigned tag = (unsigned)base + (unsigned)step + 1u;
        if (owner) {
            // ---- hand-off: this lane's four values as two 16-byte stores of (value, tag) pairs -- each 8-byte pair validates itself, so the
            // consumers poll the DATA: no drain, no flag, no second round trip (the LL idiom of the collective libraries).  Rows past
            // the batch travel too (finite, unused).
            if (step + 1 < T) {
                float* xp = ll + ((int64_t)((gidx * 2 + (step & 1)) * NM + m) * R * 16 + (R == 32 ? w : ch) * 128 + lane) * 4;
                COOP_STORE16_LL(xp, hnew[0], hnew[1], tag);
                COOP_STORE16_LL(xp + 64 * 4, hnew[2], hnew[3], tag);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lrow + i < nvalid) y_lane[(int64_t)i * S.y_row + (int64_t)t * S.y_t] = hnew[i];      // the output sequence itself: plain stores
            if (stash) {
                float4* sp = stash + ((((int64_t)tile * T + t) * NM + m) * 20 + sq) * 64 + slane;
                sp[0 * 4 * 64] = make_float4(ca[0], ca[1], ca[2], ca[3]);
                sp[1 * 4 * 64] = make_float4(cb[0], cb[1], cb[2], cb[3]);
                sp[2 * 4 * 64] = make_float4(us[0], us[1], us[2], us[3]);
                sp[3 * 4 * 64] = make_float4(rs[0], rs[1], rs[2], rs[3]);
                sp[4 * 4 * 64] = make_float4(an[0], an[1], an[2], an[3]);
            }
        }
        COOP_PHASE(3);
        if (step + 1 == T) break;
        // (every wave is done reading h_{t-1}: the barrier of the quarter exchange says so)
        COOP_PHASE(4);
        // ---- every thread polls its share of the S members' packets until all tags are this step's, then rebuilds the R x H tile in LDS
        {
            constexpr int NK = R * NM * 16 / COOP_NT;                      // 16-byte packets per thread: R * 16 per member, COOP_NT threads
            static_assert(NK == 8 || NK == 4 || NK == 2, "packet poll is written for H = 128 / 256");
            f32x4 v[16];
#pragma unroll
            for (int k = NK; k < 16; ++k) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* xr = ll + (int64_t)(gidx * 2 + (step & 1)) * NM * R * 64 + (int64_t)tid * 4;
            unsigned need = (1u << NK) - 1u;
            int polls = 0;
            if (inject && step == 0 && tid == 0) atomicAdd(status, 1);      // fault injection (diagnostics): report, then wait normally
            while (true) {
#pragma unroll
                for (int k = 0; k < NK; ++k)
                    if (need >> k & 1u) COOP_LOAD16_LL(v[k], xr + (int64_t)k * (COOP_NT * 4));
                COOP_WAIT_LL8(v);
#pragma unroll
                for (int k = 0; k < NK; ++k)
                    if ((need >> k & 1u) && COOP_TAG(v[k][1]) == tag && COOP_TAG(v[k][3]) == tag) need &= ~(1u << k);
                if (need == 0u) break;
                COOP_BACKOFF();
                ++polls;
                // a launch that has already reported a timeout gives up at once (results undefined, the launch ends)
                if (polls > max_polls || (polls == 64 && !inject && COOP_FLAG_LOAD(status) != 0)) { atomicAdd(status, 1); break; }
            }
            COOP_PHASE(5);
            // packet (64-lane block bi = w + 8k of the group's packet stream, lane): member bi / (R/4), block bi % (R/4) = (tile, pair)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int bi = w + (COOP_NT / 64) * k, mm = bi / (R / 4), blk = bi % (R / 4), pj = blk & 1, tw = blk >> 1;
                const int prow_ = (R == 32 ? (tw >> 1) * 16 : 0) + 4 * kg + 2 * pj, pcol = mm * 32 + (tw & 1) * 16 + c16;
                hs[prow_ * LDH + pcol] = v[k][0];
                hs[(prow_ + 1) * LDH + pcol] = v[k][2];
            }
        }
        COOP_PHASE(6);
        __syncthreads();
        COOP_PHASE(7);
    }
    COOP_PHASE_END();
    if (S.hn && owner) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lrow + i < nvalid) S.hn[(int64_t)(row0 + lrow + i) * S.hn_row + col0 + lcol] = hprev[i];
    }
}

template <int H, int R>
__global__ __launch_bounds__(COOP_NT) COOP_WAVES_PER_SIMD void gru_coop_fwd_kernel(GruFwdParams P, int* __restrict__ flags, int* epoch, int advance,
                                                           int* __restrict__ status, int max_polls) {
    const int base = epoch[0];
    gru_coop_fwd_body<H, R>(P, flags, base, status, max_polls);
    coop_advance_epoch(epoch, base, advance);
}

#ifdef VAME_EMU
#define COOP_ALLOW_LDS(kernel, bytes)
#else      // > 64 KiB of dynamic LDS must be granted per kernel
#define COOP_ALLOW_LDS(kernel, bytes) \
    VAME_CHECK_ARG(hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) == hipSuccess, \
                   VAME_E_HIP, "gru_coop: cannot reserve %d bytes of LDS", (int)(bytes))
#endif

template <int H, int R>
static size_t coop_fwd_lds() { return (size_t)(96 * (H + 4) + R * (H + 4) + (R == 32 ? 4 * 24 * 64 : 2 * 3 * 12 * 64)) * 4; }

// one flag word per (stream, 16-row half tile, member) in the first COOP_LL_OFF words (BPTT), then the forward kernel's tagged hand-off
// packets: 2 step parities x 32 rows x H (value, tag) pairs per (stream, tile)
extern "C" int64_t vame_gru_coop_flag_ints(int nstreams, int B, int H) {
    return COOP_LL_OFF + (int64_t)nstreams * cdiv64(B, 32) * 2 * 32 * H * 2;
}

// 1 if (nstreams, B, H) can run cooperatively: every workgroup of the grid must be resident at once (one per CU)
// compute units of the current device: every workgroup of a cooperative grid needs its own (LDS allows one per CU)
static int coop_cu_count() {
#ifdef VAME_EMU
    return 256;
#else
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev] = n;
    }
    return cus[dev];
#endif
}

template <int H, int R> static size_t coop_bwd_lds();
template <int H, int R> __global__ __launch_bounds__(COOP_NT) COOP_WAVES_PER_SIMD void gru_coop_bwd_kernel(GruBwdParams, float*, int*, int*, int, int*, int);
// The runtime's own answer to "how many of these workgroups does one CU hold" (registers, LDS, waves): every cooperative kernel
// must get >= 1, and the grid is then limited to ONE workgroup per CU (their LDS footprints exclude a second one anyway).
static int coop_kernels_resident(int H) {
#ifdef VAME_EMU
    return 1;
#else
    static int ok[2] = {-1, -1};
    int& r = ok[H == 256];
    if (r < 0) {
        int nf = 0, nb = 0;
        hipError_t e1, e2;
        if (H == 256) {
            int n16 = 0;
            e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, gru_coop_fwd_kernel<256, 32>, COOP_NT, coop_fwd_lds<256, 32>());
            if (e1 == hipSuccess) e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n16, gru_coop_fwd_kernel<256, 16>, COOP_NT, coop_fwd_lds<256, 16>());
            e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gru_coop_bwd_kernel<256, 32>, COOP_NT, coop_bwd_lds<256, 32>());
            if (e2 == hipSuccess) e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n16, gru_coop_bwd_kernel<256, 16>, COOP_NT, coop_bwd_lds<256, 16>());
            nf = nf < n16 ? nf : n16;
        } else {
            int n16 = 0;
            e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, gru_coop_fwd_kernel<128, 32>, COOP_NT, coop_fwd_lds<128, 32>());
            if (e1 == hipSuccess) e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n16, gru_coop_fwd_kernel<128, 16>, COOP_NT, coop_fwd_lds<128, 16>());
            e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gru_coop_bwd_kernel<128, 32>, COOP_NT, coop_bwd_lds<128, 32>());
            if (e2 == hipSuccess) e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n16, gru_coop_bwd_kernel<128, 16>, COOP_NT, coop_bwd_lds<128, 16>());
            nf = nf < n16 ? nf : n16;
        }
        r = (e1 == hipSuccess && e2 == hipSuccess && nf >= 1 && nb >= 1) ? 1 : 0;
        (void)hipGetLastError();
    }
    return r;
#endif
}

extern "C" int vame_gru_coop_supported(int nstreams, int B, int H) {
    if (H != 128 && H != 256) return 0;
    const int64_t groups = (int64_t)nstreams * cdiv64(B, 32);
    const int64_t grid = cdiv64(groups, 8) * 8 * (H / 32);
    return grid <= 256 && grid <= coop_cu_count() && coop_kernels_resident(H);
}
// forward: split the 32-row tiles of a launch into 16-row groups?  (twice the workgroups must still get a CU each)
static bool coop_rows16(int ngroups, int H) {
    const int64_t grid = cdiv64(ngroups, 8) * 8 * (H / 32) * 2;
    return grid <= 256 && grid <= coop_cu_count();
}
// rows [row0, row0 + nrows) of the batch (row0 a multiple of 32; nrows = 0: all of it)
static int coop_row_range(int B, int row0, int nrows, int& tile_off, int& ntiles) {
    if (nrows <= 0) { row0 = 0; nrows = B; }
    if (row0 < 0 || row0 % 32 || row0 + nrows > B) return 0;
    tile_off = row0 / 32;
    ntiles = (int)cdiv64(nrows, 32);
    return 1;
}

extern "C" int vame_gru_coop_fwd_f32(const int64_t* desc, int nstreams, int B, int H, int row0, int nrows, int* flags, int64_t flag_ints, int* epoch,
                                     int* status, void* stream) {
    VAME_CHECK_ARG(desc && flags && epoch && status && nstreams >= 1 && nstreams <= 8 && B >= 1, VAME_E_BADARG, "gru_coop_fwd: bad arguments");
    int tile_off, ntiles;
    VAME_CHECK_ARG(coop_row_range(B, row0, nrows, tile_off, ntiles), VAME_E_SHAPE, "gru_coop_fwd: bad row range %d+%d of %d", row0, nrows, B);
    VAME_CHECK_ARG(vame_gru_coop_supported(nstreams, ntiles * 32, H), VAME_E_UNSUPPORTED,
                   "gru_coop_fwd: nstreams=%d rows=%d H=%d does not fit one workgroup per CU (or H not 128/256)", nstreams, ntiles * 32, H);
    VAME_CHECK_ARG(flag_ints >= vame_gru_coop_flag_ints(nstreams, ntiles * 32, H), VAME_E_SHAPE,
                   "gru_coop_fwd: flags holds %lld ints, this launch needs %lld (vame_gru_coop_flag_ints)", (long long)flag_ints,
                   (long long)vame_gru_coop_flag_ints(nstreams, ntiles * 32, H));
    GruFwdParams P;
    if (int rc = gru_parse_fwd(desc, nstreams, B, P)) return rc;
    P.ntiles = ntiles; P.tile_off = tile_off;
    int advance = 0;
    for (int i = 0; i < nstreams; ++i) advance = (int)P.s[i].T > advance ? (int)P.s[i].T : advance;
    advance += 2;
    for (int i = 0; i < nstreams; ++i) {
        VAME_CHECK_ARG(P.s[i].xf == 0 && P.s[i].y, VAME_E_UNSUPPORTED, "gru_coop_fwd: stream %d needs a precomputed gi and an output sequence", i);
        VAME_CHECK_ARG((uintptr_t)P.s[i].y % 16 == 0 && P.s[i].y_row % 4 == 0 && P.s[i].y_t % 4 == 0 &&
                       (!P.s[i].h0 || ((uintptr_t)P.s[i].h0 % 16 == 0 && P.s[i].h0_row % 4 == 0)), VAME_E_SHAPE,
                       "gru_coop_fwd: stream %d: sequence / initial-state rows must be 16-byte aligned", i);
    }
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = nstreams * P.ntiles;
#ifdef VAME_EMU
    emu::g_coop = true;
#endif
    VAME_CHECK_ARG(P.kernel == VAME_GRU_KERNEL_AUTO || P.kernel == VAME_GRU_KERNEL_LOCKSTEP, VAME_E_UNSUPPORTED,
                   "gru_coop_fwd: kernel option %d (AUTO, or LOCKSTEP = 32-row groups)", P.kernel);
    // 16-row groups when twice the workgroups still get a CU each (GF_OPT kernel = LOCKSTEP keeps 32-row groups: same bits, A/B tests)
    const bool r16 = P.kernel == VAME_GRU_KERNEL_AUTO && coop_rows16(ngroups, H);
#define COOP_FWD_LAUNCH(HH, RR)                                                                                                  \
    do {                                                                                                                         \
        const size_t lds_ = coop_fwd_lds<HH, RR>();                                                                              \
        const int grid_ = coop_grid<HH / 32>(ngroups) * (32 / RR);                                                               \
        COOP_ALLOW_LDS((gru_coop_fwd_kernel<HH, RR>), lds_);                                                                     \
        hipLaunchKernelGGL((gru_coop_fwd_kernel<HH, RR>), dim3(grid_), dim3(COOP_NT), lds_, st, P, flags, epoch, advance, status, g_coop_polls); \
    } while (0)
    if (H == 256) { if (r16) COOP_FWD_LAUNCH(256, 16); else COOP_FWD_LAUNCH(256, 32); }
    else          { if (r16) COOP_FWD_LAUNCH(128, 16); else COOP_FWD_LAUNCH(128, 32); }
#undef COOP_FWD_LAUNCH
#ifdef VAME_EMU
    emu::g_coop = false;
#endif
    VAME_LAUNCH_CHECK("gru_coop_fwd");
    return VAME_OK;
}

// ------------------------------------------------------------------------------------------------------------- backward
// BPTT with the same split: member m owns hidden columns C_m = [32m, 32m+32).  Per step it turns dh_t[:, C_m] (+ dy) into its
// slice of dG (the BPTT coefficients of its columns come from its part of the forward stash), multiplies the R x 96 tile
// [da_r | da_z | dgh_n] with its 96 x H slice of W_hh (LDS-resident, taken from the backward-packed weights) -- a partial
// dh_{t-1} over ALL H columns -- and the members reduce-scatter those partials through a double-buffered exchange buffer:
// write-through stores + flag, then every member sums the S partials of its own 32 columns in member order and adds the
// u-gated carry.  Same descriptor table / dG / dbias / dh0 contract as vame_gru_seq_bwd_f32; results equal up to the
// summation order of that K = 3H contraction (split by member here).
//
// Round 4: 16 x 16 x 4 MFMA tiles and R = 32 or 16 rows per group like the forward kernel.  A lane of the element-wise phase owns
// four consecutive rows of one column -- one float4 of each stash quantity -- which is also one accumulator of a 16 x 16 tile, so
// the partials leave the MFMA waves as float4 packets straight from the accumulators ([column tile][4-row group][column] in the
// exchange buffer), the reduce-scatter lands in the registers of the lane that needs the sum for the next step, and neither the
// partial tile nor the carry passes through LDS.  The two 16-row groups of a tile add their bias partials in a fixed order
// (the upper one hands its sums to the lower one at the end of the launch), and the 32-row form sums in the same order.
template <int H, int R>
__device__ __forceinline__ void gru_coop_bwd_body(const GruBwdParams& P, float* __restrict__ xbuf, int* __restrict__ flags, int base,
                                                  int* __restrict__ status, int max_polls) {
    constexpr int NM = H / 32, NH = 32 / R, LDK = 100, LDG = 132, NCT = H / 16, NWV = COOP_NT / 64, TPW = NCT * (R / 16) / NWV, RG = R / 4;
    static_assert((NM == 8 || NM == 4) && (R == 32 || R == 16), "written for H = 128 / 256, 32- or 16-row groups");
    VAME_DYN_SMEM(smem_raw);
    float* wl = reinterpret_cast<float*>(smem_raw);                    // [H][LDK]  W_hh[gate*H + C_m][n] as wl[n][gate*32 + j]
    float* gs = wl + H * LDK;                                           // [R][LDG]  da_r | da_z | dgh_n | dgi_n of this member's columns
    int g, m, half;
    if (!coop_map<NM, NH>(P.nstreams * P.ntiles, g, m, half)) return;
    const bool inject = max_polls < 0;
    if (inject) max_polls = COOP_DEFAULT_POLLS;
    const int sidx = g % P.nstreams, tile = g / P.nstreams + P.tile_off;
    const GruBwdStream S = P.s[sidx];              // a copy, not a reference into the kernel arguments (see the forward kernel)
    const int B = P.B, T = (int)S.T;
    const int row0 = tile * 32 + half * R, col0 = 32 * m;
    const int nvalid = B - row0;
    if (nvalid <= 0) return;                                            // the upper half of a last tile with <= 16 rows
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, kg = lane >> 4;
    const int w = UNIFORM(tid >> 6);
    // element-wise phase: lane (hh, li) of wave w < R/8 owns rows 8w + 4hh .. + 3 of the group, column li of the slice
    const bool ew = w < R / 8;
    const int li = lane & 31, hh = lane >> 5, rgl = 2 * w + hh, erow = 4 * rgl;         // 4-row group / first row inside the group
    const int sq = (half * R + erow) >> 3;                             // q of the 32-row stash tile (hh is the same)
    const int prow = tid >> 3, pc4 = tid & 7;
    const int gidx = g * NH + half;
    int* gflags = flags + (int64_t)gidx * NM;

    {   // W_hh rows {gate*H + C_m} x all columns, from the backward pack: wp_bwd[((ct*(3H/8) + c)*64 + l)*4 + e] = W[8c + 4(l>>5) + e][32ct + (l&31)]
        const f32x4* src = reinterpret_cast<const f32x4*>(S.wpt);
        for (int i = tid; i < NM * 12 * 64; i += COOP_NT) {
            const int ct = i / (12 * 64), rem = i % (12 * 64), gg = rem / 256, c64 = rem % 256, cc = c64 / 64, l = c64 % 64;
            *reinterpret_cast<f32x4*>(&wl[(32 * ct + (l & 31)) * LDK + gg * 32 + 8 * cc + 4 * (l >> 5)]) =
                src[((int64_t)ct * (3 * H / 8) + (gg * H + 32 * m) / 8) * 64 + c64];
        }
    }
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    if (ew && S.dhn) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (erow + j < nvalid) carry[j] = S.dhn[(int64_t)(row0 + erow + j) * S.dhn_row + col0 + li];
    }
    const float4* stash = reinterpret_cast<const float4*>(S.stash);
    float4 sa, sb, su, sr, sg;
    float dyv[4];
    // The next step's coefficients and dy are requested one step ahead, by EVERY wave, unconditionally: behind a per-wave or per-step
    // condition the loaded values reach the loop-carried registers through copies, and hipcc waits for the loads right where they were
    // issued (probe: 1500-3000 cycles per step).  Waves without an element-wise share re-read wave 0/1's packets, a missing dy reads the
    // stash instead, rows past the batch re-read the group's first row -- all masked where the values are used.
    const int wl_ = ew ? w : (w & (R / 8 - 1));
    const int sq_l = (half * R + 8 * wl_ + 4 * hh) >> 3, erow_l = 8 * wl_ + 4 * hh;
    const float* dy_lane = S.dy ? S.dy + (int64_t)(row0 + erow_l) * S.dy_row + col0 + li : S.stash;
    int64_t dy_off[4];
    float dy_mask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dy_off[j] = S.dy ? (erow_l + j < nvalid ? (int64_t)j : -(int64_t)erow_l) * S.dy_row : 0;
        dy_mask[j] = (S.dy && ew && erow + j < nvalid) ? 1.0f : 0.0f;
    }
    const int64_t dy_t = S.dy ? S.dy_t : 0;
    auto load_step = [&](int step) {
        const int fstep = T - 1 - step;
        const int t = S.reverse ? T - 1 - fstep : fstep;
        const float4* sp = stash + ((((int64_t)tile * T + t) * NM + m) * 20 + sq_l) * 64 + lane;
        sa = sp[0 * 4 * 64]; sb = sp[1 * 4 * 64]; su = sp[2 * 4 * 64]; sr = sp[3 * 4 * 64]; sg = sp[4 * 4 * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) dyv[j] = dy_lane[dy_off[j] + (int64_t)t * dy_t];
    };
    sa = sb = su = sr = sg = make_float4(0.f, 0.f, 0.f, 0.f);
    dyv[0] = dyv[1] = dyv[2] = dyv[3] = 0.f;
    load_step(0);
    float dbs0 = 0.f, dbs1 = 0.f, dbs2 = 0.f, dbs3 = 0.f;
    // MFMA phase: wave w owns the 16-row half rh and TPW column tiles from ct0 on
    const int rh = R == 32 ? w / (NWV / 2) : 0, ct0 = R == 32 ? (w % (NWV / 2)) * TPW : w * TPW;
    const float* arow = &gs[(rh * 16 + c16) * LDG + 4 * kg];
    const float* brow = &wl[(ct0 * 16 + c16) * LDK + 4 * kg];
    __syncthreads();

    COOP_PHASE_DECL();
    for (int step = 0; step < T; ++step) {
        const int fstep = T - 1 - step;
        const int t = S.reverse ? T - 1 - fstep : fstep;
        if (ew) {
            const float av[4] = {sa.x, sa.y, sa.z, sa.w}, bv[4] = {sb.x, sb.y, sb.z, sb.w}, uv[4] = {su.x, su.y, su.z, su.w},
                        rv[4] = {sr.x, sr.y, sr.z, sr.w}, gv[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = carry[j] + (dy_mask[j] != 0.f ? dyv[j] : 0.f);
                const float dan = d * av[j];
                const float dau = d * bv[j];
                const float dgh = dan * rv[j];
                const float dar = dgh * gv[j] * (1.0f - rv[j]);
                carry[j] = d * uv[j];                                // dh carried through the update gate
                float* gw = &gs[(erow + j) * LDG + li];
                gw[0] = dar; gw[32] = dau; gw[64] = dgh; gw[96] = dan;
                dbs0 += dar; dbs1 += dau; dbs2 += dan; dbs3 += dgh;
            }
        }
        load_step(step + 1 < T ? step + 1 : step);  // (the last step re-reads its own)
        COOP_PHASE(0);
        __syncthreads();
        COOP_PHASE(1);
        if (tid < R * 8 && prow < nvalid) {    // dG[b][t][da_r | da_z | dgi_n | dgh_n] columns C_m: LDS blocks (r, z, gh_n, gi_n) -> global (r, z, gi_n, gh_n)
            float* dgt = S.dg + ((int64_t)(row0 + prow) * T + t) * 4 * H + col0 + 4 * pc4;
            const float* src = &gs[prow * LDG + 4 * pc4];
            *reinterpret_cast<float4*>(dgt) = *reinterpret_cast<const float4*>(src);
            *reinterpret_cast<float4*>(dgt + H) = *reinterpret_cast<const float4*>(src + 32);
            *reinterpret_cast<float4*>(dgt + 3 * H) = *reinterpret_cast<const float4*>(src + 64);
            *reinterpret_cast<float4*>(dgt + 2 * H) = *reinterpret_cast<const float4*>(src + 96);
        }
        if (step + 1 == T && S.dh0 == nullptr) break;                     // nobody asks for dh before the first step
        // ---- partial dh_{t-1} = [da_r | da_z | dgh_n] W_hh[C_m rows], published straight from the accumulators (write-through)
        float* xs = xbuf + ((int64_t)(gidx * 2 + (step & 1)) * NM) * R * H;
        COOP_PHASE(2);
        {
            f32x4 acc[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto ldf = [](const float* p_) { return *reinterpret_cast<const f32x4*>(p_); };
            f32x4 a = ldf(arow), b[TPW];            // next chunk requested before this chunk's MFMAs (see the forward kernel)
#pragma unroll
            for (int i = 0; i < TPW; ++i) b[i] = ldf(brow + i * 16 * LDK);
#pragma unroll 2
            for (int c = 0; c < 6; ++c) {
                const int cn = c + 1 < 6 ? c + 1 : c;
                const f32x4 na = ldf(arow + 16 * cn);
                f32x4 nb[TPW];
#pragma unroll
                for (int i = 0; i < TPW; ++i) nb[i] = ldf(brow + i * 16 * LDK + 16 * cn);
                SCHED_FENCE();
#pragma unroll
                for (int e = 0; e < 4; ++e)           // element-major: consecutive MFMAs go to different accumulators
#pragma unroll
                    for (int i = 0; i < TPW; ++i) acc[i] = MFMA_16x16x4(a[e], b[i][e], acc[i]);
                SCHED_FENCE();
                a = na;
#pragma unroll
                for (int i = 0; i < TPW; ++i) b[i] = nb[i];
            }
            float* xm = xs + (int64_t)m * R * H + ((rh * 4 + kg) * 16 + c16) * 4;          // packet (column tile, 4-row group, column)
            COOP_MFMA_SETTLE();
            COOP_PHASE(3);
#pragma unroll
            for (int i = 0; i < TPW; ++i) COOP_STORE16(xm + (int64_t)(ct0 + i) * RG * 64, acc[i]);
        }
        COOP_DRAIN();
        COOP_PHASE(4);
        __syncthreads();
        COOP_PHASE(5);
        if (tid == 0) COOP_FLAG_STORE(&gflags[m], (int)((unsigned)base + (unsigned)step + 1u));
        if (tid < NM) {
            if (inject && step == 0 && tid == 0) atomicAdd(status, 1);          // fault injection (diagnostics): report, then wait normally
            int polls = COOP_FLAG_LOAD(status) != 0 && !inject ? max_polls : 0;
            while ((int)((unsigned)COOP_FLAG_LOAD(&gflags[tid]) - ((unsigned)base + (unsigned)step + 1u)) < 0) {   // wrap-safe
                COOP_BACKOFF();
                if (++polls > max_polls) { atomicAdd(status, 1); break; }
            }
        }
        __syncthreads();
        COOP_PHASE(6);
        if (ew) {       // the S partials of this lane's packet (column tile 2m + (li >> 4), row group rgl, column li & 15), in member order
            f32x4 v[8];
            const float* xr = xs + (((2 * m + (li >> 4)) * RG + rgl) * 16 + (li & 15)) * 4;
#pragma unroll
            for (int mm = 0; mm < 8; ++mm) {
                if (mm < NM) COOP_LOAD16(v[mm], xr + (int64_t)mm * R * H); else v[mm] = v[0];
            }
            COOP_WAIT_LOADS8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
            for (int mm = 0; mm < NM; ++mm) {
                carry[0] += v[mm][0]; carry[1] += v[mm][1]; carry[2] += v[mm][2]; carry[3] += v[mm][3];
            }
        }
        COOP_PHASE(7);
    }
    COOP_PHASE_END();
    if (S.dh0 && ew) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (erow + j < nvalid) S.dh0[(int64_t)(row0 + erow + j) * S.dh0_row + col0 + li] = carry[j];
    }
    if (S.dbias) {      // per-column sums over the tile's rows and all steps: the 4-row groups' partials meet in LDS, 16 rows at a time
        __syncthreads();                                                 // (the last step's dG pass has read gs)
        float* red = gs;                                                 // [RG][4][32]
        if (ew) {
            red[(rgl * 4 + 0) * 32 + li] = dbs0; red[(rgl * 4 + 1) * 32 + li] = dbs1;
            red[(rgl * 4 + 2) * 32 + li] = dbs2; red[(rgl * 4 + 3) * 32 + li] = dbs3;
        }
        __syncthreads();
        float s = 0.f;
        const int k = (tid >> 5) & 3, c = tid & 31;
        if (tid < 128) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) s += red[(sl * 4 + k) * 32 + c];
            if (R == 32) {
                float s2 = 0.f;
#pragma unroll
                for (int sl = 4; sl < 8; ++sl) s2 += red[((sl % RG) * 4 + k) * 32 + c];
                s += s2;
            }
        }
        if (R == 16) {
            // hand-over between the two 16-row groups of a tile: the upper one leaves its sums in the exchange slot it is not reading
            // any more and raises its flag once more; the lower one adds them to its own and writes the tile's row
            const bool partner = tile * 32 + 16 < B;
            const int steps_x = S.dh0 ? T : T - 1;                       // exchange steps of this launch; the last one used parity (steps_x - 1) & 1
            float* hx = xbuf + ((int64_t)((g * NH + 1) * 2 + (steps_x & 1)) * NM + m) * R * H;
            int* pflag = flags + ((int64_t)(g * NH + 1) * NM + m);
            const int done = (int)((unsigned)base + (unsigned)T + 1u);
            if (half == 1) {
                if (tid < 128) COOP_STORE4(hx + tid, s);
                COOP_DRAIN();
                __syncthreads();
                if (tid == 0) COOP_FLAG_STORE(pflag, done);
                return;
            }
            if (partner) {
                if (tid == 0) {
                    int polls = COOP_FLAG_LOAD(status) != 0 && !inject ? max_polls : 0;
                    while ((int)((unsigned)COOP_FLAG_LOAD(pflag) - (unsigned)done) < 0) {
                        COOP_BACKOFF();
                        if (++polls > max_polls) { atomicAdd(status, 1); break; }
                    }
                }
                __syncthreads();
                if (tid < 128) { float o; COOP_LOAD4(o, hx + tid); s += o; }
            }
        }
        if (tid < 128) S.dbias[(int64_t)tile * 4 * H + k * H + col0 + c] = s;
    }
}

template <int H, int R>
__global__ __launch_bounds__(COOP_NT) COOP_WAVES_PER_SIMD void gru_coop_bwd_kernel(GruBwdParams P, float* __restrict__ xbuf, int* __restrict__ flags, int* epoch,
                                                           int advance, int* __restrict__ status, int max_polls) {
    const int base = epoch[0];
    gru_coop_bwd_body<H, R>(P, xbuf, flags, base, status, max_polls);
    coop_advance_epoch(epoch, base, advance);
}

template <int H, int R>
static size_t coop_bwd_lds() { return (size_t)(H * 100 + R * 132) * 4; }

extern "C" int64_t vame_gru_coop_xbuf_floats(int nstreams, int B, int H) { return (int64_t)nstreams * cdiv64(B, 32) * 2 * 32 * H * (H / 32); }

extern "C" int vame_gru_coop_bwd_f32(const int64_t* desc, int nstreams, int B, int H, int row0, int nrows, float* xbuf, int* flags, int64_t flag_ints,
                                     int* epoch, int* status, void* stream) {
    VAME_CHECK_ARG(desc && xbuf && flags && epoch && status && nstreams >= 1 && nstreams <= 8 && B >= 1, VAME_E_BADARG, "gru_coop_bwd: bad arguments");
    int tile_off, ntiles;
    VAME_CHECK_ARG(coop_row_range(B, row0, nrows, tile_off, ntiles), VAME_E_SHAPE, "gru_coop_bwd: bad row range %d+%d of %d", row0, nrows, B);
    VAME_CHECK_ARG(vame_gru_coop_supported(nstreams, ntiles * 32, H), VAME_E_UNSUPPORTED,
                   "gru_coop_bwd: nstreams=%d rows=%d H=%d does not fit one workgroup per CU (or H not 128/256)", nstreams, ntiles * 32, H);
    VAME_CHECK_ARG((uintptr_t)xbuf % 16 == 0, VAME_E_SHAPE, "gru_coop_bwd: exchange buffer must be 16-byte aligned");
    VAME_CHECK_ARG(flag_ints >= COOP_LL_OFF, VAME_E_SHAPE, "gru_coop_bwd: flags holds %lld ints, BPTT needs the %d flag words", (long long)flag_ints, (int)COOP_LL_OFF);
    GruBwdParams P;
    if (int rc = gru_parse_bwd(desc, nstreams, B, P)) return rc;
    P.ntiles = ntiles; P.tile_off = tile_off;
    int advance = 0;
    for (int i = 0; i < nstreams; ++i) advance = (int)P.s[i].T > advance ? (int)P.s[i].T : advance;
    advance += 2;
    for (int i = 0; i < nstreams; ++i)
        VAME_CHECK_ARG((uintptr_t)P.s[i].dg % 16 == 0, VAME_E_SHAPE, "gru_coop_bwd: stream %d: dG must be 16-byte aligned", i);
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = nstreams * P.ntiles;
#ifdef VAME_EMU
    emu::g_coop = true;
#endif
    VAME_CHECK_ARG(P.kernel == VAME_GRU_KERNEL_AUTO || P.kernel == VAME_GRU_KERNEL_LOCKSTEP, VAME_E_UNSUPPORTED,
                   "gru_coop_bwd: kernel option %d (AUTO, or LOCKSTEP = 32-row groups)", P.kernel);
    const bool r16 = P.kernel == VAME_GRU_KERNEL_AUTO && coop_rows16(ngroups, H);
#define COOP_BWD_LAUNCH(HH, RR)                                                                                                  \
    do {                                                                                                                         \
        const size_t lds_ = coop_bwd_lds<HH, RR>();                                                                              \
        const int grid_ = coop_grid<HH / 32>(ngroups) * (32 / RR);                                                               \
        COOP_ALLOW_LDS((gru_coop_bwd_kernel<HH, RR>), lds_);                                                                     \
        hipLaunchKernelGGL((gru_coop_bwd_kernel<HH, RR>), dim3(grid_), dim3(COOP_NT), lds_, st, P, xbuf, flags, epoch, advance, status, g_coop_polls); \
    } while (0)
    if (H == 256) { if (r16) COOP_BWD_LAUNCH(256, 16); else COOP_BWD_LAUNCH(256, 32); }
    else          { if (r16) COOP_BWD_LAUNCH(128, 16); else COOP_BWD_LAUNCH(128, 32); }
#undef COOP_BWD_LAUNCH
#ifdef VAME_EMU
    emu::g_coop = false;
#endif
    VAME_LAUNCH_CHECK("gru_coop_bwd");
    return VAME_OK;
}
