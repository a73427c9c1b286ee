// Can a wave's vector-ALU / LDS work issue beside ANOTHER wave's bf16 MFMAs on the same SIMD?  Eight waves per workgroup, one workgroup per CU:
// waves 0-3 (one per SIMD) issue v_mfma_f32_32x32x16_bf16 back to back on four rotating accumulators (in arch VGPRs or in AGPRs), waves 4-7 (their
// SIMD partners) run a loop of one instruction kind.  Each side alone, then both: cycles per MFMA and per partner instruction (s_memtime, wave 0 / 4
// of workgroup 0).  hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap_probe.hip -o tools/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { V_IDLE = 0, V_AND, V_SUB, V_PKADD, V_PERM, V_DSW, V_DSR, V_MIX };

template <bool AGPR>
__device__ __forceinline__ void mfma_loop(int n, u32x4 a, u32x4 b, f32x16 (&acc)[4]) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
}

template <int KIND>
__device__ __forceinline__ void partner_loop(int n, float* lds, unsigned seed, float& sink) {
    unsigned r0 = seed, r1 = seed * 3, r2 = seed * 5, r3 = seed * 7, r4 = seed * 11, r5 = seed * 13, r6 = seed * 17, r7 = seed * 19;
    f32x2 p0 = {1.f + seed, 2.f}, p1 = {3.f, 4.f + seed}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f};
    u32x4 q = {seed, seed + 1, seed + 2, seed + 3};
    const unsigned l = (threadIdx.x & 255) * 16;      // byte offset in LDS (`lds` is the kernel's only LDS object: offset 0)
    for (int it = 0; it < n; ++it) {
        if (KIND == V_AND)
            asm volatile("v_and_b32 %0, 0xffff0000, %0\n v_and_b32 %1, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3\n"
                         "v_and_b32 %4, 0xffff0000, %4\n v_and_b32 %5, 0xffff0000, %5\n v_and_b32 %6, 0xffff0000, %6\n v_and_b32 %7, 0xffff0000, %7"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        if (KIND == V_SUB)
            asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %4\n"
                         "v_sub_f32 %4, %4, %5\n v_sub_f32 %5, %5, %6\n v_sub_f32 %6, %6, %7\n v_sub_f32 %7, %7, %0"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        if (KIND == V_PKADD)
            asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %0 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %0 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %1 neg_lo:[0,1] neg_hi:[0,1]"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (KIND == V_PERM)
            asm volatile("v_perm_b32 %0, %1, %0, %8\n v_perm_b32 %1, %2, %1, %8\n v_perm_b32 %2, %3, %2, %8\n v_perm_b32 %3, %4, %3, %8\n"
                         "v_perm_b32 %4, %5, %4, %8\n v_perm_b32 %5, %6, %5, %8\n v_perm_b32 %6, %7, %6, %8\n v_perm_b32 %7, %0, %7, %8"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(0x07060302u));
        if (KIND == V_DSW)
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:8192\n ds_write_b128 %0, %1 offset:12288\n"
                         "ds_write_b128 %0, %1 offset:16384\n ds_write_b128 %0, %1 offset:20480\n ds_write_b128 %0, %1 offset:24576\n ds_write_b128 %0, %1 offset:28672\n"
                         "s_waitcnt lgkmcnt(0)" : : "v"(l), "v"(q) : "memory");
        if (KIND == V_DSR) {
            u32x4 t0, t1, t2, t3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n"
                         "ds_read_b128 %0, %4 offset:16384\n ds_read_b128 %1, %4 offset:20480\n ds_read_b128 %2, %4 offset:24576\n ds_read_b128 %3, %4 offset:28672\n"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(l) : "memory");
            r0 ^= t0[0] ^ t1[1] ^ t2[2] ^ t3[3];
        }
        if (KIND == V_MIX)     // the split's own mix per 8 instructions: 4 and, 2 pk_add... as in split_octet (8 and : 4 pk_add : 6 perm per 18)
            asm volatile("v_and_b32 %0, 0xffff0000, %4\n v_and_b32 %1, 0xffff0000, %5\n v_pk_add_f32 %8, %8, %9 neg_lo:[0,1] neg_hi:[0,1]\n v_perm_b32 %2, %1, %0, %10\n"
                         "v_and_b32 %3, 0xffff0000, %6\n v_perm_b32 %4, %3, %2, %10\n v_and_b32 %5, 0xffff0000, %7\n v_pk_add_f32 %9, %9, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_perm_b32 %6, %5, %4, %10"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(p0), "+v"(p1) : "s"(0x07060302u));
    }
    sink = __builtin_bit_cast(float, r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) + p0[0] + p1[1] + p2[0] + p3[1];
}

template <int KIND, bool AGPR>
__global__ __launch_bounds__(512, 1) void probe(float* out, long long* cyc, int n_m, int n_v) {
    __shared__ __attribute__((aligned(16))) float lds[40960];       // 160 KB: one workgroup per CU
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960; i += 512) lds[i] = (float)i;
    __syncthreads();
    float res = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
        if (AGPR) mfma_loop<true>(n_m, a, b, acc); else mfma_loop<false>(n_m, a, b, acc);
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    } else if (KIND != V_IDLE) {
        partner_loop<KIND>(n_v, lds, threadIdx.x, res);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
}

template <int KIND, bool AGPR>
static void run(const char* name, int per_iter) {
    float* out; long long* cyc; long long h[8];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int NM = 20000, NV = KIND == V_DSW || KIND == V_DSR ? 20000 : 40000;
    double r[3][2];
    for (int mode = 0; mode < 3; ++mode) {       // 0: MFMA alone, 1: partner alone, 2: both
        const int nm = mode == 1 ? 0 : NM, nv = mode == 0 ? 0 : NV;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((probe<KIND, AGPR>), dim3(256), dim3(512), 0, 0, out, cyc, nm, nv); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        r[mode][0] = nm ? (double)h[0] / (4.0 * nm) : 0.0;
        r[mode][1] = nv ? (double)h[4] / ((double)per_iter * nv) : 0.0;
    }
    printf("%-34s acc in %s: MFMA alone %5.1f cyc/MFMA | partner alone %5.1f cyc/instr | together: %5.1f cyc/MFMA, %5.1f cyc/instr\n", name, AGPR ? "AGPR" : "VGPR",
           r[0][0], r[1][1], r[2][0], r[2][1]);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<V_AND, false>("v_and_b32", 8);        run<V_AND, true>("v_and_b32", 8);
    run<V_SUB, false>("v_sub_f32 (dependent chain)", 8); run<V_SUB, true>("v_sub_f32 (dependent chain)", 8);
    run<V_PKADD, false>("v_pk_add_f32", 8);   run<V_PKADD, true>("v_pk_add_f32", 8);
    run<V_PERM, false>("v_perm_b32", 8);      run<V_PERM, true>("v_perm_b32", 8);
    run<V_MIX, false>("split mix (and/pk_add/perm)", 9); run<V_MIX, true>("split mix (and/pk_add/perm)", 9);
    run<V_DSW, false>("ds_write_b128 (x8 + wait)", 8); run<V_DSW, true>("ds_write_b128 (x8 + wait)", 8);
    run<V_DSR, false>("ds_read_b128 (x8 + wait)", 8);  run<V_DSR, true>("ds_read_b128 (x8 + wait)", 8);
    return 0;
}
