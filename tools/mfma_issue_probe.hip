// Issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 streams as the cooperative GRU kernels issue them: NACC independent
// accumulators round-robin, one or two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_probe.hip -o tools/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int N>
__global__ void k16(float* out, long long* cyc, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int N>
__global__ void k32(float* out, long long* cyc, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    f32x16 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int q = 0; q < 16; ++q) r += s[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// the cooperative forward kernel's chunk: 1 A + 3 B fragments (ds_read_b128) per 12 MFMAs, requested one chunk ahead; STRIDE4 = distance of
// the four lane groups' float4 slots in a row (1: slot 4c + kg, 16: slot 16 kg + c)
template <int STRIDE4, int N>
__global__ __launch_bounds__(512) void k16lds(float* out, long long* cyc, float a_, float b_) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LD = 260;
    for (int i = threadIdx.x; i < 128 * LD; i += blockDim.x) lds[i] = a_ * (float)(i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, c16 = lane & 15, kg = lane >> 4, w = (threadIdx.x >> 6) & 3;
    const float* arow = lds + (96 + (w & 1) * 16 + c16) * LD + 4 * kg * STRIDE4;
    const float* brow = lds + ((w >> 1) * 16 + c16) * LD + 4 * kg * STRIDE4;
    f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto ldf = [](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
    const int cstep = STRIDE4 == 1 ? 16 : 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N / 8; ++it) {
        f32x4 a = ldf(arow), b[3] = {ldf(brow), ldf(brow + 32 * LD), ldf(brow + 64 * LD)};
#pragma unroll 2
        for (int c = 0; c < 8; ++c) {
            const int cn = c + 1 < 8 ? c + 1 : c;
            const f32x4 na = ldf(arow + cstep * cn);
            const f32x4 nb[3] = {ldf(brow + cstep * cn), ldf(brow + 32 * LD + cstep * cn), ldf(brow + 64 * LD + cstep * cn)};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[g][e], acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a = na; b[0] = nb[0]; b[1] = nb[1]; b[2] = nb[2];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const f32x4 s = acc[0] + acc[1] + acc[2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename K>
static void runlds(const char* name, K kern, int nmfma, int threads = 256) {
    float* out; long long* cyc; long long h = 0;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 260 * 4);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 128 * 260 * 4, 0, out, cyc, 1.0f, 0.5f); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 128 * 260 * 4, 0, out, cyc, 1.0f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %d wave/SIMD : %6.1f s_memtime ticks per MFMA per wave; wall clock (one workgroup per CU, incl. the LDS fill) %6.1f TFLOP/s (%s)\n", name,
           threads / 256, (double)h / nmfma, 10.0 * 256 * (threads / 64) * (double)nmfma * 2048.0 / ms * 1e-9, hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(cyc);
}
template <typename K>
static void run(const char* name, K kern, int threads, int nmfma, double flop_per) {
    float* out; long long* cyc; long long h = 0;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f, 0.5f); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 10.0 * 256 * (threads / 64) * (double)nmfma * flop_per;
    printf("%-44s %d waves/SIMD: %6.1f s_memtime ticks per MFMA per wave; wall clock %7.1f TFLOP/s\n", name, threads / 256, (double)h / nmfma, flops / ms * 1e-9);
    hipFree(out); hipFree(cyc);
}
int main() {
    constexpr int N = 20000;
    run("16x16x4  3 accumulators", k16<3, N>, 256, 3 * N, 2048.0);  run("16x16x4  3 accumulators", k16<3, N>, 512, 3 * N, 2048.0);
    run("16x16x4  6 accumulators", k16<6, N>, 256, 6 * N, 2048.0);  run("16x16x4  6 accumulators", k16<6, N>, 512, 6 * N, 2048.0);
    run("16x16x4  8 accumulators", k16<8, N>, 256, 8 * N, 2048.0);  run("16x16x4  1 accumulator (dependent)", k16<1, N>, 256, N, 2048.0);
    run("32x32x2  1 accumulator (dependent)", k32<1, N>, 256, N, 4096.0); run("32x32x2  3 accumulators", k32<3, N>, 256, 3 * N, 4096.0);
    run("32x32x2  3 accumulators", k32<3, N>, 512, 3 * N, 4096.0);
    run("16x16x4  3 accumulators", k16<3, N>, 1024, 3 * N, 2048.0);
    run("32x32x2  3 accumulators", k32<3, N>, 1024, 3 * N, 4096.0);
    runlds("16x16x4 + LDS fragments, slot 4c + kg", k16lds<1, 19200>, 19200 / 8 * 96);
    runlds("16x16x4 + LDS fragments, slot 16kg + c", k16lds<16, 19200>, 19200 / 8 * 96);
    runlds("16x16x4 + LDS fragments, slot 16kg + c", k16lds<16, 19200>, 19200 / 8 * 96, 512);
    return 0;
}
