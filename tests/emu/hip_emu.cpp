// Fiber scheduler for tests/emu/hip_emu.h (test infrastructure).
#include "hip_emu.h"
#include <memory>

thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
namespace emu { thread_local BlockCtx* g_blk = nullptr; }

asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace emu {
static constexpr size_t STACK = 96 * 1024;

static void fiber_entry() {
    BlockCtx* b = g_blk;
    b->body();
    b->fibers[b->cur].done = true;
    emu_switch(&b->fibers[b->cur].sp, b->sched_sp);
    abort();
}

static void run_block(BlockCtx& b, dim3 grid, dim3 block, uint3 bid) {
    g_blk = &b;
    gridDim = grid; blockDim = block; blockIdx = bid;
    b.bar_count = 0; b.bar_gen = 0;
    for (auto& w : b.waves) { w.count = 0; w.gen = 0; }
    for (int t = 0; t < b.nthreads; ++t) {
        Fiber& f = b.fibers[t];
        f.done = false;
        uintptr_t top = ((uintptr_t)(f.stack + STACK)) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address (keeps rsp = 8 mod 16 at entry)
        *--sp = (void*)&fiber_entry;     // popped by `ret`
        for (int i = 0; i < 6; ++i) *--sp = nullptr;
        f.sp = sp;
    }
    int alive = b.nthreads;
    while (alive > 0) {
        for (int t = 0; t < b.nthreads; ++t) {
            Fiber& f = b.fibers[t];
            if (f.done) continue;
            b.cur = t;
            threadIdx.x = t % block.x;
            threadIdx.y = (t / block.x) % block.y;
            threadIdx.z = t / (block.x * block.y);
            emu_switch(&b.sched_sp, f.sp);
            if (f.done) --alive;
        }
    }
}

bool g_coop = false;      // cooperative kernels (cross-block flags): one OS thread per block so every block makes progress

void launch_impl(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    const int nthreads = block.x * block.y * block.z;
    if (nthreads % WAVE != 0) { fprintf(stderr, "emu: block size %d not a multiple of 64\n", nthreads); abort(); }
    const long nblocks = (long)grid.x * grid.y * grid.z;
    int nworkers = (int)std::min<long>(nblocks, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("VAME_EMU_THREADS")) nworkers = std::max(1, std::min(nworkers, atoi(e)));
    if (g_coop) nworkers = (int)nblocks;
    std::atomic<long> next{0};
    auto worker = [&]() {
        BlockCtx b;
        b.nthreads = nthreads;
        b.body = body;
        b.fibers.resize(nthreads);
        b.waves.resize(nthreads / WAVE);
        b.dyn_smem.assign(shmem + 64, 0);
        std::unique_ptr<char[]> stacks(new char[(size_t)nthreads * STACK]);      // not zeroed: only touched pages get committed
        for (int t = 0; t < nthreads; ++t) b.fibers[t].stack = stacks.get() + (size_t)t * STACK;
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= nblocks) break;
            uint3 bid{(unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long)grid.x * grid.y))};
            run_block(b, grid, block, bid);
        }
        g_blk = nullptr;
    };
    if (nworkers <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nworkers; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
}
}  // namespace emu
