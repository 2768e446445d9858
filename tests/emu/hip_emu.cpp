// tests/emu/hip_emu.cpp -- runtime of the CPU SIMT emulation (TEST INFRASTRUCTURE ONLY, see hip/hip_runtime.h)
#include <hip/hip_runtime.h>
#include <mutex>

namespace hipemu {
thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
static pthread_barrier_t g_block_bar;
static pthread_barrier_t g_wave_bar[16];
static uint64_t g_wave_buf[16][64];
static std::vector<unsigned char> g_dyn;

void sync_block() { pthread_barrier_wait(&g_block_bar); }
void sync_wave() { pthread_barrier_wait(&g_wave_bar[t_threadIdx.x >> 6]); }
uint64_t *wave_buf() { return g_wave_buf[t_threadIdx.x >> 6]; }
void *dyn_shared() { return g_dyn.data(); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &fn)
{
    const unsigned nt = block.x;
    if (nt == 0 || grid.x == 0) return;
    if (nt % 64 != 0 || nt > 1024) { fprintf(stderr, "hipemu: block size %u unsupported\n", nt); abort(); }
    static std::mutex one_launch;                 // static __shared__ storage, global barriers: one kernel at a time, whichever host thread launches
    std::lock_guard<std::mutex> lock(one_launch);
    g_dyn.assign(shmem + 64, 0);
    pthread_barrier_init(&g_block_bar, nullptr, nt);
    for (unsigned w = 0; w < nt / 64; ++w) pthread_barrier_init(&g_wave_bar[w], nullptr, 64);
    std::vector<std::thread> th;
    th.reserve(nt);
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([=, &fn]() {
            const unsigned gy = grid.y ? grid.y : 1;
            t_threadIdx = Idx{t, 0, 0}; t_blockDim = Idx{nt, 1, 1}; t_gridDim = Idx{grid.x, gy, 1};
            for (unsigned y = 0; y < gy; ++y)
                for (unsigned b = 0; b < grid.x; ++b) {      // blocks run one after the other (static __shared__ storage)
                    t_blockIdx = Idx{b, y, 0};
                    fn();
                    pthread_barrier_wait(&g_block_bar);
                }
        });
    for (auto &x : th) x.join();
    pthread_barrier_destroy(&g_block_bar);
    for (unsigned w = 0; w < nt / 64; ++w) pthread_barrier_destroy(&g_wave_bar[w]);
}
} // namespace hipemu
