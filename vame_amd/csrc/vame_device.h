// Common device-side definitions for the gfx950 kernels of the VAME RNN-VAE hot path.
// Written for MI355X (CDNA4, wave64) only.  VAME_EMU selects the host emulator used by the CPU
// test-suite (tests/emu); the shipped library is always built with hipcc for gfx950.
#pragma once
#ifdef VAME_EMU
#include "hip_emu.h"
typedef f32x16_emu f32x16;
typedef f32x4_emu f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA_32x32x2(a, b, c) emu_mfma_32x32x2((a), (b), (c))
#define MFMA_16x16x4(a, b, c) emu_mfma_16x16x4((a), (b), (c))
#define VAME_DYN_SMEM(name) char* name = emu::dyn_smem()
#define SETPRIO(n)
#define VAME_SLEEP4()
#define LDS_BARRIER() __syncthreads()
#define STREAM_STORE16(ptr, a, b, c, d) (*reinterpret_cast<float4*>(ptr) = make_float4((a), (b), (c), (d)))
#define WAVE_SYNC() emu::wave_sync()      /* lanes are fibers on the host: a wave-private LDS exchange needs an explicit rendezvous */
#define SCHED_FENCE()
#define RING_FENCE()
#define RING_LOAD(dst, ptr, idx) (dst) = *reinterpret_cast<const f32x4*>((ptr) + (idx))
#define RING_LOAD_U(dst, ubase, voff, imm) (dst) = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ubase) + (voff) + (imm))
#define RING_WAIT2(n, a, b)
#define RING_WAIT3(n, a, b, c)
#define RING_WAIT4(n, a, b, c, d)
#define VAME_EXPF(x) expf(x)
#define VAME_RCP(x) (1.0f / (x))
typedef u32x4_emu u32x4;
#define MFMA_BF16_32x32x16(a, b, c) emu_mfma_32x32x16_bf16((a), (b), (c))
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));      /* pairs for the packed fp32 VALU ops (v_pk_mul_f32, v_pk_add_f32) */
// f32-in/f32-acc matrix FMA: exact f32 (k-ordered fmaf chain) at the 157 TF rate on gfx950
#define MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define VAME_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#define VAME_SLEEP4() __builtin_amdgcn_s_sleep(4)      /* ~256 cycles off the issue ports */
// Workgroup barrier that orders LDS traffic only: this wave's LDS operations have completed (lgkmcnt(0)), then s_barrier.  Unlike
// __syncthreads() it does not drain the vector-memory queue (hipcc puts s_waitcnt vmcnt(0) in front of that barrier whenever it knows
// of outstanding global loads / stores, and the in-order counter then also waits for the inline-asm weight ring), so global stores and
// ring loads stay in flight across it.  For barriers that publish LDS data to the other waves of the workgroup and nothing else.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// 16-byte store of write-once streaming output (the BPTT stash) with the sc1 policy: written through and NOT kept in the XCD's L2 (plain
// and nt stores leave their lines there, MI355X_MICROARCH.md "stores of each flavour").  The GRU kernels re-read their W_hh fragments from
// L2 every time step; 10-16 MB of stash lines per step and XCD passing through a 4 MB L2 evict them several times per step
// (profiles/r04_cfg4_pmc_hbm_traffic.json: 6-11 GB of re-fetched weights per H = 512 launch).  asm: HIP has no cache-policy store.
#define STREAM_STORE16(ptr, a, b, c, d)                                                                        \
    do {                                                                                                       \
        const f32x4 v_ = {(a), (b), (c), (d)};                                                                 \
        /* s_nop: a VALU write of the data registers needs 2 wait states behind a > 8-byte store; hipcc does not see a store here */ \
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(v_) : "memory");      \
    } while (0)
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()   /* lanes of a wave run in lock step and its LDS accesses complete in order: ordering hint only */
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   /* keep the scheduler from merging phases (register pressure) */
#define RING_FENCE() __builtin_amdgcn_sched_barrier(0)    /* prefetch-ring refills stay behind the MFMAs that read the slot */
// Register prefetch ring for the L2-resident weight fragments.  hipcc's own s_waitcnt insertion drains the whole vector-memory
// queue (vmcnt(0)) at the back edge of a loop that carries in-flight loads, which turns a ring into "load, wait, use".  So the
// ring loads are issued from inline asm (invisible to that pass) and the consumer waits with an explicit vmcnt(n), n = number
// of ring loads issued after the slot's own.  Memory operations the compiler issues in between only make the wait more
// conservative (vmcnt counts in order), never unsafe.  The asm wait takes the slot registers as in/out operands so that no
// consumer can be scheduled above it.  (ptr: float4 pointer, idx: float4 index; offsets stay < 4 KiB of a per-lane base.)
#define RING_LOAD(dst, ptr, idx) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"((ptr) + (idx)))
// the same with a UNIFORM base in an SGPR pair + a per-lane byte offset + an immediate: the address arithmetic stays on the scalar ALU
// (a 64-bit VGPR address costs a v_lshl_add_u64 per load -- vector-ALU time the f32 MFMAs need)
#define RING_LOAD_U(dst, ubase, voff, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(ubase), "n"(imm))
#define RING_WAIT2(n, a, b) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(n))
#define RING_WAIT3(n, a, b, c) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(n))
#define RING_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n))
#define VAME_EXPF(x) __expf(x)
#define VAME_RCP(x) __builtin_amdgcn_rcpf(x)   /* v_rcp_f32, 1 ulp */
// bf16-input matrix FMA (fp32 accumulate, 16x the f32-input rate): a lane supplies 8 bf16 = four dwords of A row / B column (lane & 31),
// k = 8 * (lane >> 5) .. + 7; used by the error-compensated split contraction of gemm.hip only
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 vame_bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16_32x32x16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vame_bf16x8, (a)), __builtin_bit_cast(vame_bf16x8, (b)), (c), 0, 0, 0)
#endif
#include <stdint.h>

#define VAME_WAVE 64

// ---- buffer-resource addressing: uniform 48-bit base + byte range in four scalar registers, a 32-bit per-lane byte offset,
// a scalar byte offset and a <= 4095 B immediate.  Used where a kernel walks rows of a tile with compile-time strides: the flat
// (64-bit per-lane pointer) form makes hipcc materialise one address pair per row and spill them.  Out-of-range lanes load 0 and
// their stores are dropped, which is how ragged tiles and absent (null, 0-byte) operands are handled without branches.
// `soff` must be wave-uniform.  A range covers at most 4 GiB - 1: rebase per tile.
#ifdef VAME_EMU
struct BufRange { const char* base; uint32_t bytes; };
static inline BufRange buf_range(const void* base, uint64_t bytes) {
    return BufRange{reinterpret_cast<const char*>(base), base ? (uint32_t)(bytes > 0xffffffffull ? 0xffffffffull : bytes) : 0u};
}
static inline float buf_load_f32(BufRange r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    return o + 4 <= r.bytes ? *reinterpret_cast<const float*>(r.base + o) : 0.0f;
}
static inline float4 buf_load_f32x4(BufRange r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    return o + 16 <= r.bytes ? *reinterpret_cast<const float4*>(r.base + o) : float4{0.f, 0.f, 0.f, 0.f};
}
static inline f32x2 buf_load_f32x2(BufRange r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    return o + 8 <= r.bytes ? *reinterpret_cast<const f32x2*>(r.base + o) : f32x2{0.f, 0.f};
}
static inline void buf_store_f32(BufRange r, uint32_t voff, uint32_t soff, float v) {
    const uint64_t o = (uint64_t)voff + soff;
    if (o + 4 <= r.bytes) *reinterpret_cast<float*>(const_cast<char*>(r.base) + o) = v;
}
static inline void buf_store_f32x4(BufRange r, uint32_t voff, uint32_t soff, float4 v) {
    const uint64_t o = (uint64_t)voff + soff;
    if (o + 16 <= r.bytes) *reinterpret_cast<float4*>(const_cast<char*>(r.base) + o) = v;
}
static inline void buf_stream_store_f32(BufRange r, uint32_t voff, uint32_t soff, float v) { buf_store_f32(r, voff, soff, v); }
#else
typedef __amdgpu_buffer_rsrc_t BufRange;
__device__ __forceinline__ BufRange buf_range(const void* base, uint64_t bytes) {
    const uint32_t n = base ? (uint32_t)(bytes > 0xffffffffull ? 0xffffffffull : bytes) : 0u;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);      // gfx9 raw buffer, 32-bit data format
}
__device__ __forceinline__ float buf_load_f32(BufRange r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float4 buf_load_f32x4(BufRange r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x2 buf_load_f32x2(BufRange r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store_f32(BufRange r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
// write-once streaming output with the sc1 policy (aux bit 4): written through, not kept in the XCD's L2 (see STREAM_STORE16)
__device__ __forceinline__ void buf_stream_store_f32(BufRange r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 16);
}
typedef unsigned vame_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_store_f32x4(BufRange r, uint32_t voff, uint32_t soff, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vame_u32x4, v), r, (int)voff, (int)soff, 0);
}
#endif

// ---- 32x32 MFMA accumulator fragment (C/D) layout on gfx950:
//   lane l holds column (l & 31); register r holds row  (r&3) + 8*(r>>2) + 4*(l>>5)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float fast_sigmoid(float x) { return VAME_RCP(1.0f + VAME_EXPF(-x)); }
// tanh(x) = 1 - 2/(exp(2x)+1): absolute error ~1e-7, saturates correctly for |x| large
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * VAME_RCP(VAME_EXPF(2.0f * x) + 1.0f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
