// GO / NO-GO probe for a weight-stationary, column-split GRU forward on the bf16 matrix cores with an error-compensated split (bf16x6) at LARGE batch.
//
// Today (gru_seq.hip) a workgroup owns 32 batch rows and streams all of W_hh (786 KB at H = 256) from L2 every time step: 49.2 k cycles of
// f32-input MFMA + ~7 k of gate math that cannot overlap = 26-30 us per step.  Here a GROUP of 8 workgroups (one per CU, all on one XCD) shares 256
// rows; member c keeps the three bf16 planes of ITS 96 gate columns (hidden units [32c, 32c+32) of r, z, n) x 256 k in LDS for the whole sequence
// (144 KB) and nothing but h travels: every step each of a member's 8 waves (two per SIMD; wave w = row chunk w of 32 rows)
//   1. polls the 8 members' flags of ITS chunk, 2. contracts  W_slice (A operand, ds_read_b128 fragments) x h_{t-1}^T (B operand: the producers store
//   the three planes of h in the consumer's MFMA fragment order, one coalesced 1 KB load per fragment, L1 bypassed) = 288 v_mfma_f32_32x32x16_bf16,
//   3. does the gate math lane-locally (the operands are swapped -- D = W h^T -- so a lane holds r, z, n of 16 units of ONE batch row, and its own
//   16 h values ARE two B fragments of the next step under a fixed permutation of k that W's fragments are packed with), 4. splits h_t into three
//   bf16 planes (truncation, exact: x = x1 + x2 + x3), stores 6 x 16 B per lane + y (+ the BPTT stash), drains, raises its flag.
// MFMA time per step and SIMD: 2 waves x 288 x 32 cycles = 18.4 k cycles (7.7-9.2 us at 2.4-2.0 GHz); the partner wave's MFMAs cover a wave's gate
// math and hand-off (profiles/r05_mfma_valu_overlap_probe.txt: a bf16 MFMA stream keeps 32.0 cycles beside any partner).
//
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/rec_split_probe.hip -o tools/rec_split_probe
// Run:   tools/rec_split_probe [groups=32] [T=30] [store: 0 = sc1 write-through, 1 = plain (L2-resident, same-XCD groups)] [stash 0/1] [reps=5] [mode bits: 1 = no gi touch-ahead, 2 = s_setprio 1 on waves 0-3, 4 = plain (L2-resident) flag stores, 8 = longer sleep between polls, 16 = y / stash stores in front of the drain, 32 = no HBM streams]; reps < 0: that many launches back to back
// Output: us per launch and per step, cycles per phase (wave 0 and 4 of group 0 member 0), max |h - double reference| over group 0's first 64 rows, XCD census.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int H = 256, NM = 8, CH = 8, NT = 512, KS = 16;
constexpr int WFRAG_U4 = KS * 3 * 3 * 64;            // 16-byte units of one member's W slice: [ks][gate][plane][lane] = 147,456 B
constexpr unsigned HI16 = 0xffff0000u;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ inline float hash_unit(unsigned long long i, unsigned salt) {       // uniform in [-1, 1)
    unsigned long long x = i * 0x9E3779B97F4A7C15ull + salt * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return (float)((int)(x & 0xffffff) - 0x800000) * (1.0f / 8388608.0f);
}
// unit (0..255) that k-slot (ks, hh, j) of the permuted contraction order stands for: ks = 2 * member + s
__host__ __device__ inline int unit_of(int ks, int hh, int j) { return 32 * (ks >> 1) + 8 * (2 * (ks & 1) + (j >> 2)) + 4 * hh + (j & 3); }

struct Params {
    const u32x4* wfrag;      // [member][WFRAG_U4]
    u32x4* hx;               // [group][slot 2][chunk 8][ks 16][plane 3][lane 64]
    int* flags;              // [group][chunk 8][member 8]
    const float* gi;         // [row][T][3H]
    float* y;                // [row][T][H]
    float* stash;            // 3 x [row][T][H] or null
    const float* bhn;        // [H]
    long long* probe;        // [block][wave][8]
    int* xcc;                // [block]
    int T, plain, base, one_xcd, mode;
};

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ unsigned fb(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float uf(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) { return __builtin_amdgcn_perm(fb(hi), fb(lo), 0x07060302u); }
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

template <bool PLAIN, bool STASH>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void rec_kernel(Params P) {
    extern __shared__ u32x4 wl[];                                  // the member's W slice, fragment order
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block b runs on XCD b % 8 (observed; speed only): a group = 8 blocks of one XCD
    const int b = blockIdx.x, xcd = b & 7, slot_ = b >> 3, member = slot_ & 7;
    if (P.one_xcd && xcd != 0) return;                            // (runs with fewer than 8 groups: XCD 0's blocks only)
    const int group = P.one_xcd ? (slot_ >> 3) : (slot_ >> 3) * 8 + xcd;
    if (tid == 0) { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); P.xcc[b] = (int)(x & 15); }
    for (int i = tid; i < WFRAG_U4; i += NT) wl[i] = P.wfrag[(size_t)member * WFRAG_U4 + i];
    __syncthreads();
    const int T = P.T;
    const long long row = (long long)group * 256 + w * 32 + li;
    const float* gi_row = P.gi + row * T * 3 * H + 32 * member + 4 * hh;
    float* y_row = P.y + row * T * H + 32 * member + 4 * hh;
    const size_t plane_sz = (size_t)gridDim.x / 8 * 256 * T * H;
    __amdgpu_buffer_rsrc_t hx_rs = __builtin_amdgcn_make_buffer_rsrc(P.hx + (size_t)group * 2 * CH * KS * 3 * 64, 0, 2 * CH * KS * 3 * 64 * 16, 0x00020000);
    const int* fl = P.flags + (group * CH + w) * NM;
    float bhn[16], hprev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { bhn[r] = P.bhn[32 * member + 8 * (r >> 2) + 4 * hh + (r & 3)]; hprev[r] = 0.f; }
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = (long long)__builtin_amdgcn_s_memtime();
    const long long rt0 = (long long)__builtin_amdgcn_s_memrealtime(), c0 = t0;
#define PH(i) do { const long long t1_ = (long long)__builtin_amdgcn_s_memtime(); ph[i] += t1_ - t0; t0 = t1_; } while (0)
    constexpr int AUX = PLAIN ? 0 : 16;                            // 16 = sc1: stores written through, loads bypass L1 (always on the load side)
    const int mode = P.mode;                                       // bit 0: no touch-ahead of the next step's gi lines; bit 1: s_setprio 1 on waves 0-3
    if ((mode & 2) && w < 4) __builtin_amdgcn_s_setprio(1);
    // every VMEM load of a step is issued from inline asm in program order, so the vmcnt(n) waits below are exact (vmcnt retires in order)
// (MUBUF immediates are 12 bits: the k-step part of the offset travels in the scalar offset)
#define LD_H(dst, ks_, p_) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4 sc1" : "=v"(dst) : "v"(hoff), "s"(hx_rs), "s"((ks_) * 3072), "n"((p_) * 1024))
#define LD_GI(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(ptr))
#define WAIT3(n, a, b, c) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(n))
    for (int t = 0; t < T; ++t) {
        f32x4 gi[3][4];
        f32x16 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        const float* gp = gi_row + (long long)((mode & 32) ? 0 : t) * 3 * H;           // mode bit 32: no HBM streams (step 0's gi every step, no y / stash stores)
        if (t > 0) {
            // ---- wait for the 8 producers of this chunk's h_{t-1}
            const int want = P.base + t;
            int budget = 1 << 20;
            for (;;) {
                int f = want;
                if (lane < NM) f = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(f - want >= 0) || --budget == 0) break;
                if (mode & 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            PH(0);
            // ---- contraction: 16 k-steps x (3 planes of h from L2, 9 fragments of W from LDS, 18 MFMAs); h fragments run D k-steps ahead
            const int hoff = ((((t + 1) & 1) * CH + w) * KS * 3 * 64 + lane) * 16;       // byte offset of (slot, chunk, ks 0, plane 0, lane)
            constexpr int D = 4;
            u32x4 ring[D][3];
#pragma unroll
            for (int d = 0; d < D; ++d) { LD_H(ring[d][0], d, 0); LD_H(ring[d][1], d, 1); LD_H(ring[d][2], d, 2); }
            // this step's input projection behind the first h fragments (HBM latency: it lands during the contraction) and one touch per
            // 128-byte line of the NEXT step's (so that those loads find their lines in L2)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) LD_GI(gi[g][q], gp + g * H + 8 * q);
            constexpr int NTOUCH = 2;
            float touch[NTOUCH];
            {
                // chunk-step block of gi: 32 rows x 3 gates x 128 B; lane -> (row lane & 31, gate (lane >> 5) + 2 i) -- 96 lines, 2 loads of 64 lanes cover 128 (32 twice: harmless)
                const float* tp = P.gi + ((long long)group * 256 + w * 32 + li) * T * 3 * H + (long long)((mode & 32) ? 0 : (t + 1 < T && !(mode & 1)) ? t + 1 : t) * 3 * H + 32 * member;
                asm volatile("global_load_dword %0, %1, off" : "=v"(touch[0]) : "v"(tp + (hh ? H : 0)));
                asm volatile("global_load_dword %0, %1, off" : "=v"(touch[1]) : "v"(tp + 2 * H));
            }
            constexpr int AFTER = 3 * (D - 1) + 12 + NTOUCH;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < D) WAIT3(AFTER, ring[ks % D][0], ring[ks % D][1], ring[ks % D][2]);
                else if (KS - 1 - ks >= D - 1) WAIT3(3 * (D - 1), ring[ks % D][0], ring[ks % D][1], ring[ks % D][2]);
                else WAIT3(3 * (KS - 1 - ks), ring[ks % D][0], ring[ks % D][1], ring[ks % D][2]);     // (the last D - 1 k-steps: fewer loads behind the slot's)
                u32x4 a[3][3];
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[g][p] = wl[((ks * 3 + g) * 3 + p) * 64 + lane];
                const u32x4 b0 = ring[ks % D][0], b1 = ring[ks % D][1], b2 = ring[ks % D][2];
                // small products first, the leading one last; consecutive MFMAs go to different accumulators
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = MFMA(a[g][2], b0, acc[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = MFMA(a[g][0], b2, acc[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = MFMA(a[g][1], b1, acc[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = MFMA(a[g][1], b0, acc[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = MFMA(a[g][0], b1, acc[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = MFMA(a[g][0], b0, acc[g]);
                if (ks + D < KS) {
                    // (the asm statements keep their order; the operands tie them behind this k-step's last use of the slot)
                    asm volatile("" : "+v"(ring[ks % D][0]), "+v"(ring[ks % D][1]), "+v"(ring[ks % D][2]) : "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));
                    LD_H(ring[ks % D][0], ks + D, 0); LD_H(ring[ks % D][1], ks + D, 1); LD_H(ring[ks % D][2], ks + D, 2);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch[0]), "+v"(touch[1]));
            PH(1);
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) LD_GI(gi[g][q], gp + g * H + 8 * q);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(gi[0][0]), "+v"(gi[0][1]), "+v"(gi[0][2]), "+v"(gi[0][3]), "+v"(gi[1][0]), "+v"(gi[1][1]), "+v"(gi[1][2]), "+v"(gi[1][3]),
                     "+v"(gi[2][0]), "+v"(gi[2][1]), "+v"(gi[2][2]), "+v"(gi[2][3]));
        // ---- gates (lane: batch row li, units 8q + 4hh + e of this member)
        float hn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = r >> 2, e = r & 3;
            const float rr = fsig(gi[0][q][e] + acc[0][r]);
            const float zz = fsig(gi[1][q][e] + acc[1][r]);
            const float nn = ftanh(gi[2][q][e] + rr * (acc[2][r] + bhn[r]));
            hn[r] = nn + zz * (hprev[r] - nn);
            if (STASH) { acc[0][r] = rr; acc[1][r] = zz; acc[2][r] = nn; }
            hprev[r] = hn[r];
        }
        PH(2);
        // ---- split into three bf16 planes = two B fragments per plane of the next step, store, y, stash
        u32x4 pl[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float x0 = hn[8 * s + 2 * d], x1 = hn[8 * s + 2 * d + 1];
                const float r0 = x0 - uf(fb(x0) & HI16), r1 = x1 - uf(fb(x1) & HI16);
                const float s0 = r0 - uf(fb(r0) & HI16), s1 = r1 - uf(fb(r1) & HI16);
                pl[s][0][d] = pack_hi16(x0, x1); pl[s][1][d] = pack_hi16(r0, r1); pl[s][2][d] = pack_hi16(s0, s1);
            }
        const int obase = ((((t & 1) * CH + w) * KS + 2 * member) * 3 * 64 + lane) * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) __builtin_amdgcn_raw_buffer_store_b128(pl[s][p], hx_rs, obase + (s * 3 + p) * 1024, 0, AUX);
        // mode bit 16: the old order (y and the stash in front of the drain: their HBM write latency sits in the hand-off chain)
        auto out_streams = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {hn[4 * q], hn[4 * q + 1], hn[4 * q + 2], hn[4 * q + 3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(y_row + (long long)t * H + 8 * q));
                if (STASH) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const f32x4 sv = {acc[g][4 * q], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]};
                        __builtin_nontemporal_store(sv, reinterpret_cast<f32x4*>(P.stash + g * plane_sz + (y_row - P.y) + (long long)t * H + 8 * q));
                    }
                }
            }
        };
        if ((mode & 48) == 16) out_streams();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (mode & 4) asm volatile("global_store_dword %0, %1, off" : : "v"(const_cast<int*>(fl) + member), "v"(P.base + t + 1) : "memory");   // stays in this XCD's L2 (same-XCD groups only)
            else __hip_atomic_store(const_cast<int*>(fl) + member, P.base + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!(mode & 48)) out_streams();
        PH(3);
    }
    if (lane == 0 && P.probe) {
        long long* o = P.probe + ((long long)b * 8 + w) * 8;
        for (int i = 0; i < 4; ++i) o[i] = ph[i];
        o[4] = (long long)__builtin_amdgcn_s_memrealtime() - rt0; o[5] = (long long)__builtin_amdgcn_s_memtime() - c0; o[6] = rt0; o[7] = 0;
    }
}

__global__ void fill_gi(float* gi, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) gi[i] = 1.5f * hash_unit(i, 7);
}

static unsigned short hi16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float trunc16(float f) { unsigned u; memcpy(&u, &f, 4); u &= HI16; float r; memcpy(&r, &u, 4); return r; }

int main(int argc, char** argv) {
    const int groups = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 30, plain = argc > 3 ? atoi(argv[3]) : 0;
    const int stash = argc > 4 ? atoi(argv[4]) : 0, reps = argc > 5 ? atoi(argv[5]) : 5, mode = argc > 6 ? atoi(argv[6]) : 0;
    if (groups < 1 || groups > 32 || (groups > 4 && groups % 8)) { printf("groups: 1..4 (one XCD) or a multiple of 8 up to 32\n"); return 1; }
    // a group's 8 members are the blocks {xcd + 8 * (8 * gq + m)}: groups % 8 == 0 fills XCDs evenly; 1..4 groups: a 64 x groups grid of which only XCD 0's blocks work
    const int rows = groups * 256;
    std::vector<float> W((size_t)3 * H * H), bhn(H);
    for (size_t i = 0; i < W.size(); ++i) W[i] = 0.0625f * hash_unit(i, 1);
    for (int i = 0; i < H; ++i) bhn[i] = 0.0625f * hash_unit(i, 2);
    // fragments: [member][ks][gate][plane][lane][8 bf16]
    std::vector<unsigned short> wf((size_t)NM * WFRAG_U4 * 8);
    for (int c = 0; c < NM; ++c)
        for (int ks = 0; ks < KS; ++ks)
            for (int g = 0; g < 3; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int li = lane & 31, hh = lane >> 5;
                        const float x = W[(size_t)(g * H + 32 * c + li) * H + unit_of(ks, hh, j)];
                        const float x1 = trunc16(x), r = x - x1, x2 = trunc16(r), x3 = r - x2;
                        const float pv[3] = {x1, x2, x3};
                        for (int p = 0; p < 3; ++p)
                            wf[((((size_t)c * KS + ks) * 3 + g) * 3 + p) * 64 * 8 + lane * 8 + j] = hi16(pv[p]);
                    }
    Params P{};
    void *d_wf, *d_hx, *d_flags, *d_gi, *d_y, *d_stash = nullptr, *d_bhn, *d_probe, *d_xcc;
    const int nblocks = groups >= 8 ? groups * 8 : 64 * groups;
    const size_t gi_n = (size_t)rows * T * 3 * H, y_n = (size_t)rows * T * H;
    CK(hipMalloc(&d_wf, wf.size() * 2)); CK(hipMemcpy(d_wf, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
    const int ngroups_addr = groups;
    CK(hipMalloc(&d_hx, (size_t)ngroups_addr * 2 * CH * KS * 3 * 64 * 16)); CK(hipMemset(d_hx, 0, (size_t)ngroups_addr * 2 * CH * KS * 3 * 64 * 16));
    CK(hipMalloc(&d_flags, (size_t)ngroups_addr * CH * NM * 4)); CK(hipMemset(d_flags, 0, (size_t)ngroups_addr * CH * NM * 4));
    CK(hipMalloc(&d_gi, (size_t)ngroups_addr * 256 * T * 3 * H * 4)); CK(hipMalloc(&d_y, (size_t)ngroups_addr * 256 * T * H * 4));
    if (stash) CK(hipMalloc(&d_stash, (size_t)3 * ngroups_addr * 256 * T * H * 4));
    CK(hipMalloc(&d_bhn, H * 4)); CK(hipMemcpy(d_bhn, bhn.data(), H * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_probe, (size_t)nblocks * 8 * 8 * 8)); CK(hipMemset(d_probe, 0, (size_t)nblocks * 8 * 8 * 8));
    CK(hipMalloc(&d_xcc, nblocks * 4));
    hipLaunchKernelGGL(fill_gi, dim3(4096), dim3(256), 0, 0, (float*)d_gi, (size_t)ngroups_addr * 256 * T * 3 * H);
    P.wfrag = (const u32x4*)d_wf; P.hx = (u32x4*)d_hx; P.flags = (int*)d_flags; P.gi = (const float*)d_gi; P.y = (float*)d_y; P.stash = (float*)d_stash;
    P.bhn = (const float*)d_bhn; P.probe = (long long*)d_probe; P.xcc = (int*)d_xcc; P.T = T; P.plain = plain; P.one_xcd = groups < 8; P.mode = mode;
    const size_t lds = (size_t)WFRAG_U4 * 16;
    auto kern = plain ? (stash ? rec_kernel<true, true> : rec_kernel<true, false>) : (stash ? rec_kernel<false, true> : rec_kernel<false, false>);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0.f;
    if (reps < 0) {                                              // -n: n launches back to back (the clock governor sees sustained load), one event pair around all of them
        const int n = -reps;
        P.base = 0; hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), lds, 0, P);
        CK(hipEventRecord(e0));
        for (int r = 1; r <= n; ++r) { P.base = r * (T + 8); hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), lds, 0, P); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms / n; sum = ms; 
    }
    for (int r = 0; reps > 0 && r < reps + 1; ++r) {
        P.base = r * (T + 8);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), lds, 0, P);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) { best = ms < best ? ms : best; sum += ms; }
    }
    CK(hipDeviceSynchronize());
    printf("groups %d (rows %d, blocks %d), T %d, stores %s, stash %d, mode %d: launch %.1f us best, %.1f us mean -> %.2f / %.2f us per step\n", groups, rows, nblocks, T,
           plain ? "plain" : "sc1", stash, mode, best * 1e3, sum / (reps < 0 ? -reps : reps) * 1e3, best * 1e3 / T, sum / (reps < 0 ? -reps : reps) * 1e3 / T);
    const double hbm = (double)(gi_n + y_n * (stash ? 4 : 1)) * 4;
    printf("  HBM stream %.2f GB per launch -> %.2f TB/s at the best time; exchange %.1f MB per step through L2 / the fabric\n", hbm / 1e9, hbm / (best * 1e-3) / 1e12,
           (double)rows * H * 6 * 8 / 1e6);
    std::vector<long long> pr((size_t)nblocks * 64);
    CK(hipMemcpy(pr.data(), d_probe, pr.size() * 8, hipMemcpyDeviceToHost));
    {
        // every working wave: chain time (100 MHz real-time counter), shader clock (s_memtime ticks / real time), phases
        double ph[4] = {0, 0, 0, 0}, tmin = 1e30, tmax = 0, tsum = 0, clk = 0; int n = 0;
        for (int b = 0; b < nblocks; ++b) {
            if (groups < 8 && (b & 7)) continue;
            for (int wv = 0; wv < 8; ++wv) {
                const long long* o = pr.data() + ((size_t)b * 8 + wv) * 8;
                const double us = (double)o[4] / 100.0;
                tmin = us < tmin ? us : tmin; tmax = us > tmax ? us : tmax; tsum += us; clk += (double)o[5] / us / 1e3; ++n;
                for (int i = 0; i < 4; ++i) ph[i] += (double)o[i];
            }
        }
        printf("  %d waves: chain time min %.0f / mean %.0f / max %.0f us; shader clock %.2f GHz; mean cycles per step: poll %.0f, contraction %.0f (%.1f per MFMA), gates %.0f, split+publish %.0f\n",
               n, tmin, tsum / n, tmax, clk / n, ph[0] / n / (T - 1), ph[1] / n / (T - 1), ph[1] / n / (T - 1) / 288, ph[2] / n / T, ph[3] / n / T);
    }
    std::vector<int> xc(nblocks); CK(hipMemcpy(xc.data(), d_xcc, nblocks * 4, hipMemcpyDeviceToHost));
    int off = 0; for (int b = 0; b < nblocks; ++b) off += (xc[b] != (b & 7));
    printf("  XCD census: %d of %d blocks NOT on XCD (block %% 8)\n", off, nblocks);
    // ---- reference: group 0, chunk 0 and 1 (64 rows) in double
    const int RR = 64;
    std::vector<float> y((size_t)256 * T * H);
    CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> h((size_t)RR * H, 0.0), hn_((size_t)RR * H);
    double worst = 0, worst_last = 0;
    for (int t = 0; t < T; ++t) {
        for (int r = 0; r < RR; ++r)
            for (int u = 0; u < H; ++u) {
                double gh[3] = {0, 0, 0};
                for (int g = 0; g < 3; ++g) {
                    const float* wr = &W[(size_t)(g * H + u) * H];
                    double s = 0; for (int k = 0; k < H; ++k) s += (double)wr[k] * h[(size_t)r * H + k];
                    gh[g] = s;
                }
                const size_t gb = ((size_t)r * T + t) * 3 * H;
                const double gr = 1.5f * hash_unit(gb + u, 7), gz = 1.5f * hash_unit(gb + H + u, 7), gn = 1.5f * hash_unit(gb + 2 * H + u, 7);
                const double rr = 1 / (1 + exp(-(gr + gh[0]))), zz = 1 / (1 + exp(-(gz + gh[1]))), nn = tanh(gn + rr * (gh[2] + bhn[u]));
                hn_[(size_t)r * H + u] = nn + zz * (h[(size_t)r * H + u] - nn);
            }
        h.swap(hn_);
        for (int r = 0; r < RR; ++r)
            for (int u = 0; u < H; ++u) {
                const double d = fabs(h[(size_t)r * H + u] - (double)y[((size_t)r * T + t) * H + u]);
                worst = d > worst ? d : worst;
                if (t == T - 1) worst_last = d > worst_last ? d : worst_last;
            }
    }
    printf("  max |y - double reference| over %d rows x %d steps: %.3e (last step %.3e)\n", RR, T, worst, worst_last);
    return 0;
}
