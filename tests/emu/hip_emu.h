// Host emulation of the HIP device-language subset used by vame_amd/csrc/*.hip.
// TEST INFRASTRUCTURE ONLY: lets `pytest -m "not gpu"` execute the very same kernel sources on
// the CPU (cooperative fibers, one per HIP thread; wave64 collectives incl. MFMA emulated with
// the documented gfx950 fragment layouts) so index math is checked without a GPU.  The product
// library (libvame_hip.so) never contains or loads this code.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3 };

namespace emu {
constexpr int WAVE = 64;
struct Fiber { void* sp; char* stack; bool done; };
struct WaveState { int count = 0; int gen = 0; float buf[4][WAVE]; unsigned long long ubuf[WAVE]; unsigned mbuf[2][WAVE][4]; };
struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    int nthreads = 0, cur = 0, bar_count = 0, bar_gen = 0;
    void* sched_sp = nullptr;
    std::function<void()> body;
    std::vector<char> dyn_smem;
};
extern thread_local BlockCtx* g_blk;
extern bool g_coop;
}  // namespace emu
extern thread_local uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

extern "C" void emu_switch(void** save_sp, void* new_sp);

namespace emu {
inline void yield() { BlockCtx* b = g_blk; emu_switch(&b->fibers[b->cur].sp, b->sched_sp); }
inline int lane_id() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) % WAVE; }
inline int flat_tid() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }
inline WaveState& wave() { return g_blk->waves[flat_tid() / WAVE]; }
inline void wave_sync() {
    WaveState& w = wave();
    int g = w.gen;
    if (++w.count == WAVE) { w.count = 0; w.gen++; } else { while (w.gen == g) yield(); }
}
inline void block_sync() {
    BlockCtx* b = g_blk;
    int g = b->bar_gen;
    if (++b->bar_count == b->nthreads) { b->bar_count = 0; b->bar_gen++; } else { while (b->bar_gen == g) yield(); }
}
inline char* dyn_smem() { return g_blk->dyn_smem.data(); }
void launch_impl(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);
template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
    launch_impl(grid, block, shmem, [=]() { kernel(args...); });
}
}  // namespace emu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, ##__VA_ARGS__)

static inline void __syncthreads() { emu::block_sync(); }

template <class T> static inline T emu_shfl_src(T v, int src) {
    static_assert(sizeof(T) == 4, "4-byte shuffles only");
    emu::WaveState& w = emu::wave();
    int l = emu::lane_id();
    memcpy(&w.buf[0][l], &v, 4);
    emu::wave_sync();
    T r;
    memcpy(&r, &w.buf[0][src & 63], 4);
    emu::wave_sync();
    return r;
}
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { return emu_shfl_src(v, emu::lane_id() ^ m); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) { int l = emu::lane_id(); return emu_shfl_src(v, l + d < 64 ? l + d : l); }
template <class T> static inline T __shfl(T v, int s, int = 64) { return emu_shfl_src(v, s); }

typedef float f32x16_emu __attribute__((ext_vector_type(16)));
typedef float f32x4_emu __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; C/D col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5)
static inline f32x16_emu emu_mfma_32x32x2(float a, float b, f32x16_emu c) {
    emu::WaveState& w = emu::wave();
    int l = emu::lane_id();
    w.buf[0][l] = a; w.buf[1][l] = b;
    emu::wave_sync();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.buf[0][k * 32 + row], w.buf[1][k * 32 + col], acc);
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; C/D col=l&15,row=(l>>4)*4+r
static inline f32x4_emu emu_mfma_16x16x4(float a, float b, f32x4_emu c) {
    emu::WaveState& w = emu::wave();
    int l = emu::lane_id();
    w.buf[0][l] = a; w.buf[1][l] = b;
    emu::wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.buf[0][k * 16 + row], w.buf[1][k * 16 + col], acc);
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}
// v_mfma_f32_32x32x16_bf16: lane l supplies 8 bf16 (four dwords, low half first) of A row / B column (l & 31) for k = 8 * (l >> 5) .. + 7;
// C/D layout as the other 32x32 forms.  Products of two bf16 are exact in fp32; the 16 of a row-column pair are summed exactly here
// (double) and rounded once into the fp32 accumulator -- the device's internal order is not documented, the tests compare with a tolerance.
typedef unsigned u32x4_emu __attribute__((ext_vector_type(4)));
static inline float emu_bf16_at(const unsigned (&w)[4], int e) {
    const unsigned u = (e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline f32x16_emu emu_mfma_32x32x16_bf16(u32x4_emu a, u32x4_emu b, f32x16_emu c) {
    emu::WaveState& w = emu::wave();
    int l = emu::lane_id();
    for (int d = 0; d < 4; ++d) { w.mbuf[0][l][d] = a[d]; w.mbuf[1][l][d] = b[d]; }
    emu::wave_sync();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double s = 0.0;
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) s += (double)emu_bf16_at(w.mbuf[0][h * 32 + row], e) * (double)emu_bf16_at(w.mbuf[1][h * 32 + col], e);
        c[r] = (float)((double)c[r] + s);
    }
    emu::wave_sync();
    return c;
}
static inline float atomicAdd(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned o = __atomic_load_n(u, __ATOMIC_RELAXED), n;
    float f;
    do { __builtin_memcpy(&f, &o, 4); f += v; __builtin_memcpy(&n, &f, 4); } while (!__atomic_compare_exchange_n(u, &o, n, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    __builtin_memcpy(&f, &o, 4);
    return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline long long atomicMin(long long* p, long long v) {
    long long o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline long long atomicMax(long long* p, long long v) {
    long long o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
