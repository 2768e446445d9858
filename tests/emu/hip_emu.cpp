// tests/emu/hip_emu.cpp -- runtime of the CPU SIMT emulation (TEST INFRASTRUCTURE ONLY, see hip/hip_runtime.h)
//
// The lanes of a block are fibers on the launching thread, switched in user space: a wave operation (ballot, shuffle) or __syncthreads() is a
// barrier among 64 / all fibers, and a fiber that waits hands the processor to the next one.  (One operating-system thread per lane with
// pthread barriers, the first form of this file, spent nearly all of its time in futex calls: a ballot cost a hundred microseconds.)
// Blocks run one after the other, in order (static __shared__ storage; look-backs find their predecessors finished).  A wave operation
// inside divergent control flow, which hangs the GPU, is reported here: the scheduler sees that no fiber can move.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <mutex>

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#define EMU_ASAN 1
#else
#define EMU_ASAN 0
#endif

// void emu_switch(void **save_sp, void *load_sp): callee-saved registers of the System V x86-64 ABI on the old stack, then the new stack's
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch, @function
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
    .size emu_switch, .-emu_switch
)");

namespace hipemu {
thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {
constexpr size_t STACK = (size_t)512 << 10, GUARD = 4096;
struct Fiber { void *sp; bool done; };
struct Barrier { unsigned arrived = 0, gen = 0; };
struct Run {
    Fiber fib[1024]; unsigned nt = 0, cur = 0; void *sched_sp = nullptr;
    Barrier block_bar, wave_bar[16];
    uint64_t wave_buf[16][64];
    std::vector<unsigned char> dyn;
    const std::function<void()> *fn = nullptr;
    bool progress = false;
    unsigned char *stacks = nullptr;      // 1024 stacks, mapped once
    const void *sched_bottom = nullptr; size_t sched_size = 0;      // (address sanitizer builds: the scheduler's stack, learnt at the first switch)
};
Run g;      // (one launch at a time: the mutex in launch())

unsigned char *stack_of(unsigned t) { return g.stacks + (size_t)t * (STACK + GUARD) + GUARD; }

void to_scheduler()
{
    Fiber &f = g.fib[g.cur];
#if EMU_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(f.done ? nullptr : &fake, g.sched_bottom, g.sched_size);
#endif
    emu_switch(&f.sp, g.sched_sp);
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

void fiber_main()
{
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g.sched_bottom, &g.sched_size);
#endif
    (*g.fn)();
    g.fib[g.cur].done = true; g.progress = true;
    to_scheduler();
    abort();
}

void wait_at(Barrier &b, unsigned count)
{
    const unsigned gen = b.gen;
    if (++b.arrived == count) { b.arrived = 0; ++b.gen; g.progress = true; return; }
    while (b.gen == gen) to_scheduler();
}
} // namespace

void sync_block() { wait_at(g.block_bar, g.nt); }
void sync_wave() { wait_at(g.wave_bar[t_threadIdx.x >> 6], 64); }
void yield() { to_scheduler(); }
uint64_t *wave_buf() { return g.wave_buf[t_threadIdx.x >> 6]; }
void *dyn_shared() { return g.dyn.data(); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &fn, const char *kernel_name)
{
    const unsigned nt = block.x;
    if (nt == 0 || grid.x == 0) return;
    if (nt % 64 != 0 || nt > 1024) { fprintf(stderr, "hipemu: block size %u unsupported\n", nt); abort(); }
    static std::mutex one_launch;                 // static __shared__ storage, global barriers: one kernel at a time, whichever host thread launches
    std::lock_guard<std::mutex> lock(one_launch);
    if (!g.stacks) {
        g.stacks = (unsigned char *)mmap(nullptr, 1024 * (STACK + GUARD), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g.stacks == MAP_FAILED) { perror("hipemu: mmap"); abort(); }
        for (unsigned t = 0; t < 1024; ++t) mprotect(g.stacks + (size_t)t * (STACK + GUARD), GUARD, PROT_NONE);
    }
    g.dyn.assign(shmem + 64, 0);
    g.nt = nt; g.fn = &fn;
    // HIPEMU_REVERSE=<kernel name>[,<kernel name>...]: the blocks of those kernels run from the last to the first, and so do the lanes of a
    // block.  Threads that are meant to be independent (one cluster of the left-justification each) must not care; run in index order only,
    // an overlap between two of them looks like the sequential algorithm and stays hidden (it did: dw_walk.hip reach_del, found on the GPU).
    // Not for kernels whose blocks wait for their predecessors (look-backs).
    // HIPEMU_REVERSE_LANES=<kernel names>: only the lanes (blocks stay in order): for kernels with look-backs; shows code that counts on the
    // lock step of a wave where no wave operation enforces it.
    auto named_in = [&](const char *var) {
        const char *e = getenv(var);
        if (!e) return false;
        const char *nm = kernel_name ? kernel_name : "";
        const char *colon = strrchr(nm, ':'); if (colon) nm = colon + 1;                  // dw::k_jrun -> k_jrun
        const size_t n = strcspn(nm, "<( ");
        for (const char *q = e; *q;) { const size_t m = strcspn(q, ","); if ((m == n && !strncmp(q, nm, n)) || (m == 3 && !strncmp(q, "all", 3))) return true; q += m; if (*q == ',') ++q; }
        return false;
    };
    const bool reverse_blocks = named_in("HIPEMU_REVERSE");
    const bool reverse_lanes = reverse_blocks || named_in("HIPEMU_REVERSE_LANES");
    const unsigned gy = grid.y ? grid.y : 1;
    const Idx saved[4] = {t_threadIdx, t_blockIdx, t_blockDim, t_gridDim};
    t_blockDim = Idx{nt, 1, 1}; t_gridDim = Idx{grid.x, gy, 1};
    for (unsigned y = 0; y < gy; ++y)
        for (unsigned bb = 0; bb < grid.x; ++bb) {      // blocks run one after the other (static __shared__ storage)
            const unsigned b = reverse_blocks ? grid.x - 1 - bb : bb;
            t_blockIdx = Idx{b, y, 0};
            g.block_bar = Barrier();
            for (unsigned w = 0; w < nt / 64; ++w) g.wave_bar[w] = Barrier();
            for (unsigned t = 0; t < nt; ++t) {
                void **top = (void **)(stack_of(t) + STACK);
                top[-2] = (void *)&fiber_main;            // return address of the first switch; the stack pointer is 8 mod 16 behind it, as after a call
                for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
                g.fib[t].sp = (void *)(top - 8); g.fib[t].done = false;
            }
            unsigned left = nt, idle_rounds = 0;
            while (left) {
                g.progress = false;
                for (unsigned tt = 0; tt < nt; ++tt) {
                    const unsigned t = reverse_lanes ? nt - 1 - tt : tt;
                    if (g.fib[t].done) continue;
                    g.cur = t; t_threadIdx = Idx{t, 0, 0};
#if EMU_ASAN
                    void *fake = nullptr;
                    __sanitizer_start_switch_fiber(&fake, stack_of(t), STACK);
#endif
                    emu_switch(&g.sched_sp, g.fib[t].sp);
#if EMU_ASAN
                    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
                    if (g.fib[t].done) --left;
                }
                // no barrier completed, no lane finished: either lanes poll memory for one another (s_sleep loops), or some lanes wait at a wave
                // operation the others will never reach
                idle_rounds = g.progress ? 0 : idle_rounds + 1;
                if (idle_rounds > 100000) { fprintf(stderr, "hipemu: block %u: no lane can move -- a wave operation (ballot, shuffle, barrier) inside divergent control flow?\n", b); abort(); }
            }
        }
    t_threadIdx = saved[0]; t_blockIdx = saved[1]; t_blockDim = saved[2]; t_gridDim = saved[3];
}
} // namespace hipemu
