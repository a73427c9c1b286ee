// In-kernel phase timing of the product GEMM kernel (build: make probe; run on the MI355X: tools/probe_gemm M N K akm bkm splitk).
// Compiles vame_amd/csrc/gemm.hip with -DVAME_PROBE: every wave accumulates s_memtime ticks spent in (a) register->LDS
// staging incl. the wait for the prefetched global loads, (b) barrier 1, (c) the MFMA phase, (d) barrier 2, and stamps
// its begin/end with s_memtime and the constant-rate s_memrealtime, which also gives the shader clock under load.
#include "../vame_amd/csrc/gemm.hip"
#include <algorithm>
#include <vector>


int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: probe_gemm M N K akm bkm splitk\n"); return 2; }
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), akm = atoi(argv[4]), bkm = atoi(argv[5]), sk = atoi(argv[6]);
    const size_t na = (size_t)M * K, nb = (size_t)N * K;
    float *A, *B, *C, *ws;
    hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&ws, (size_t)std::max(sk, 1) * M * N * 4);
    std::vector<float> h(std::max(na, nb));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
    const int maxwg = 1 << 16, waves = 4;
    long long* probe; hipMalloc(&probe, (size_t)maxwg * waves * 8 * 8); hipMemset(probe, 0, (size_t)maxwg * waves * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_probe), &probe, sizeof(probe));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return vame_gemm_f32(M, N, K, A, akm ? M : K, akm, 0, 0, B, bkm ? N : K, bkm, 0, 0, nullptr, C, N, 0, sk, ws, 0, 0, nullptr); };
    for (int i = 0; i < 3; ++i) if (run()) return 1;
    hipDeviceSynchronize();
    float ms = 0; const int reps = 5;
    hipEventRecord(e0, 0); for (int i = 0; i < reps; ++i) run(); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("M=%d N=%d K=%d akm=%d bkm=%d sk=%d: %.1f us per call (incl. split-K reduce), %.1f TF\n", M, N, K, akm, bkm, sk, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    std::vector<long long> p((size_t)maxwg * waves * 8);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    long long tmin = -1, tmax = 0; double s[4] = {0, 0, 0, 0}, dur = 0, rdur = 0; size_t n = 0;
    std::vector<double> starts, ends, durs;
    for (size_t w = 0; w < (size_t)maxwg * waves; ++w) {
        const long long* o = &p[w * 8];
        if (!o[1]) continue;
        if (tmin < 0 || o[2] < tmin) tmin = o[2];
        tmax = std::max(tmax, o[3]);
    }
    for (size_t w = 0; w < (size_t)maxwg * waves; ++w) {
        const long long* o = &p[w * 8];
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
    const double ktiles = (double)((K + sk - 1) / sk + 31) / 32;
    printf("memtime ticks per k-tile per wave: %.0f (mfma phase %.0f; 64 MFMAs x 64 cycles = 4096 if alone on the SIMD)\n", tot / n / ktiles, s[2] / n / ktiles);
    return 0;
}
