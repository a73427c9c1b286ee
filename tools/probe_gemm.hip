// In-kernel phase timing of the product GEMM kernel (build: make probe; run on the MI355X: tools/probe_gemm M N K akm bkm splitk).
// Compiles vame_amd/csrc/gemm.hip with -DVAME_PROBE: every wave accumulates s_memtime ticks spent in (a) register->LDS
// staging incl. the wait for the prefetched global loads, (b) barrier 1, (c) the MFMA phase, (d) barrier 2, and stamps
// its begin/end with s_memtime and the constant-rate s_memrealtime, which also gives the shader clock under load.
#include "../vame_amd/csrc/gemm.hip"
#include <algorithm>
#include <string.h>
#include <vector>


// calibration: what s_memtime counts, and the MFMA rate / clock the chip sustains with no memory traffic at all
__global__ void spin_kernel(long long* out, int iters) {
    const long long t0 = PROBE_T(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    int x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525 + 1013904223;
    if (threadIdx.x == 0) { out[0] = PROBE_T() - t0; out[1] = (long long)__builtin_amdgcn_s_memrealtime() - r0; out[2] = x; }
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma_only_kernel(long long* out, float* sink, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f - a;
    const long long t0 = PROBE_T(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = MFMA_32x32x2(a, b, acc[j]);
    }
    const long long t1 = PROBE_T(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) v += acc[j][0] + acc[j][7];
    if (v == 123.456f) sink[0] = v;
    if ((threadIdx.x & 63) == 0) {
        long long* o = out + ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = r0; o[3] = r1;
    }
}
static void calib() {
    long long* d; float* sink; hipMalloc(&d, 1 << 22); hipMalloc(&sink, 64);
    long long h[4];
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 400000); hipDeviceSynchronize();
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 400000); hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("spin (1 wave, idle chip): %lld memtime ticks / %lld realtime ticks (100 MHz) -> ratio %.3f\n", h[0], h[1], (double)h[0] / h[1]);
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 40000 / wgs_per_cu, NACC = 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_only_kernel<4>, dim3(grid), dim3(256), 0, 0, d, sink, iters); hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_only_kernel<4>, dim3(grid), dim3(256), 0, 0, d, sink, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> v((size_t)grid * 4 * 4); hipMemcpy(v.data(), d, v.size() * 8, hipMemcpyDeviceToHost);
        double t = 0, r = 0; long long rmin = v[2], rmax = v[3];
        for (size_t w = 0; w < (size_t)grid * 4; ++w) { t += v[w * 4]; r += v[w * 4 + 1]; rmin = std::min(rmin, v[w * 4 + 2]); rmax = std::max(rmax, v[w * 4 + 3]); }
        const double flops = (double)grid * 4 * iters * NACC * 4096.0;
        printf("mfma-only, %d WG/CU (4 waves, %d accumulators each): %.1f us, %.1f TF (event) / %.1f TF (in-kernel span); clock ratio %.3f; cycles per MFMA per SIMD %.1f\n",
               wgs_per_cu, NACC, ms * 1e3, flops / ms / 1e9, flops / ((double)(rmax - rmin) * 10.0) / 1e3, t / r,
               t / ((double)grid * 4) / ((double)iters * NACC) / wgs_per_cu);
    }
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "calib")) { calib(); return 0; }
    if (argc < 7) { fprintf(stderr, "usage: probe_gemm M N K akm bkm splitk\n"); return 2; }
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), akm = atoi(argv[4]), bkm = atoi(argv[5]), sk = atoi(argv[6]);
    const size_t na = (size_t)M * K, nb = (size_t)N * K;
    float *A, *B, *C, *ws;
    hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&ws, (size_t)std::max(sk, 1) * M * N * 4);
    std::vector<float> h(std::max(na, nb));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
    const int maxwg = 1 << 16, waves = 4;
    long long* probe; hipMalloc(&probe, (size_t)maxwg * waves * PROBE_SLOTS * 8); hipMemset(probe, 0, (size_t)maxwg * waves * PROBE_SLOTS * 8);
    vame_probe_set_gemm(probe);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return vame_gemm_f32(M, N, K, A, akm ? M : K, akm, 0, 0, B, bkm ? N : K, bkm, 0, 0, nullptr, C, N, 0, sk, ws, 0, 0, nullptr); };
    for (int i = 0; i < 3; ++i) if (run()) return 1;
    hipDeviceSynchronize();
    float ms = 0; const int reps = 5;
    hipEventRecord(e0, 0); for (int i = 0; i < reps; ++i) run(); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("M=%d N=%d K=%d akm=%d bkm=%d sk=%d: %.1f us per call (incl. split-K reduce), %.1f TF\n", M, N, K, akm, bkm, sk, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    std::vector<long long> p((size_t)maxwg * waves * PROBE_SLOTS);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    long long tmin = -1, tmax = 0; double s[4] = {0, 0, 0, 0}, dur = 0, rdur = 0; size_t n = 0;
    std::vector<double> starts, ends, durs;
    for (size_t w = 0; w < (size_t)maxwg * waves; ++w) {
        const long long* o = &p[w * PROBE_SLOTS];
        if (!o[1]) continue;
        if (tmin < 0 || o[2] < tmin) tmin = o[2];
        tmax = std::max(tmax, o[3]);
    }
    for (size_t w = 0; w < (size_t)maxwg * waves; ++w) {
        const long long* o = &p[w * PROBE_SLOTS];
        if (!o[1]) continue;
        ++n; dur += (double)(o[1] - o[0]); rdur += (double)(o[3] - o[2]);
        for (int i = 0; i < 4; ++i) s[i] += (double)o[4 + i];
        starts.push_back((double)(o[2] - tmin)); ends.push_back((double)(o[3] - tmin)); durs.push_back((double)(o[3] - o[2]));
    }
    if (!n) { printf("no probe data\n"); return 1; }
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end()); std::sort(durs.begin(), durs.end());
    auto q = [&](std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))] / 100.0; };   // s_memrealtime = 100 MHz -> us
    printf("waves %zu; kernel span (realtime) %.1f us; memtime ticks / realtime tick = %.3f (x100 MHz = clock in MHz if s_memtime counts shader clocks)\n",
           n, (double)(tmax - tmin) / 100.0, dur / rdur);
    printf("wave start  us: min %.1f p50 %.1f p90 %.1f max %.1f\n", q(starts, 0), q(starts, .5), q(starts, .9), q(starts, 1));
    printf("wave end    us: min %.1f p10 %.1f p50 %.1f max %.1f\n", q(ends, 0), q(ends, .1), q(ends, .5), q(ends, 1));
    printf("wave length us: min %.1f p50 %.1f max %.1f\n", q(durs, 0), q(durs, .5), q(durs, 1));
    const double tot = s[0] + s[1] + s[2] + s[3];
    printf("loop time split: stage(regs->LDS, incl. vmcnt wait) %.1f%%  barrier1 %.1f%%  mfma phase %.1f%%  barrier2 %.1f%%;  loop = %.1f%% of wave length\n",
           100 * s[0] / tot, 100 * s[1] / tot, 100 * s[2] / tot, 100 * s[3] / tot, 100 * tot / dur);
    {   // whole wave lifetime (kernel entry -> stores acknowledged) vs the loop: what the slots are held for, and how full they are
        double pro = 0, epi = 0, life = 0; long long emin = -1, emax = 0;
        std::vector<double> lifes;
        for (size_t w = 0; w < (size_t)maxwg * waves; ++w) {
            const long long* o = &p[w * PROBE_SLOTS];
            if (!o[1]) continue;
            pro += (double)(o[2] - o[8]); epi += (double)(o[9] - o[3]); life += (double)(o[9] - o[8]); lifes.push_back((double)(o[9] - o[8]));
            if (emin < 0 || o[8] < emin) emin = o[8];
            emax = std::max(emax, o[9]);
        }
        std::sort(lifes.begin(), lifes.end());
        printf("wave lifetime us: mean %.1f p50 %.1f  = prologue (entry -> loop) %.1f + loop %.1f + epilogue (loop end -> stores acknowledged) %.1f\n",
               life / n / 100.0, q(lifes, .5), pro / n / 100.0, rdur / n / 100.0, epi / n / 100.0);
        printf("kernel span entry->exit %.1f us; mean resident waves %.0f of %d slots (3 workgroups x 4 waves x 256 CUs)\n", (double)(emax - emin) / 100.0,
               life / (double)(emax - emin), 3072);
    }
    const double ktiles = (double)((K + sk - 1) / sk + 31) / 32;
    printf("memtime ticks per k-tile per wave: %.0f (mfma phase %.0f; 64 MFMAs x 64 cycles = 4096 if alone on the SIMD)\n", tot / n / ktiles, s[2] / n / ktiles);
    return 0;
}
