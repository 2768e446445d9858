// tools/ubench_pin.hip -- analysis only: ways of getting page-locked host memory and what they cost: hipHostMalloc, hipHostRegister of malloc'ed memory with and
// without transparent huge pages, several threads at once; D2H copy rate into each; time for the process to go away afterwards.
//   ubench_pin <mode> <mb> <n> [threads]     mode: malloc | register | register_thp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double mono() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double epoch() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "malloc"; const size_t mb = argc > 2 ? (size_t)atol(argv[2]) : 100; const int n = argc > 3 ? atoi(argv[3]) : 4, nt = argc > 4 ? atoi(argv[4]) : 1;
    (void)hipFree(nullptr);
    std::vector<void *> bufs((size_t)n, nullptr);
    const size_t bytes = mb << 20;
    auto get = [&](int i) {
        void *p = nullptr;
        if (!strcmp(mode, "malloc")) { if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) p = nullptr; }
        else {
            p = mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            p = (void *)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
            if (!strcmp(mode, "register_thp")) madvise(p, bytes, MADV_HUGEPAGE);
            memset(p, 0, bytes);
            if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); p = nullptr; }
        }
        bufs[(size_t)i] = p;
    };
    double t = mono();
    if (nt <= 1) for (int i = 0; i < n; ++i) get(i);
    else { std::vector<std::thread> th; for (int k = 0; k < nt; ++k) th.emplace_back([&, k]() { for (int i = k; i < n; i += nt) get(i); }); for (auto &x : th) x.join(); }
    const double dt = mono() - t;
    printf("%s: %d x %zu MB with %d thread(s): %.4f s = %.2f ms per 100 MB\n", mode, n, mb, nt, dt, dt / (n * mb / 100.0) * 1e3);
    void *d = nullptr; (void)hipMalloc(&d, bytes); (void)hipMemset(d, 3, bytes); (void)hipDeviceSynchronize();
    if (bufs[0]) {
        t = mono();
        for (int r = 0; r < 4; ++r) for (int i = 0; i < n; ++i) if (bufs[(size_t)i]) (void)hipMemcpyAsync(bufs[(size_t)i], d, bytes, hipMemcpyDeviceToHost, 0);
        (void)hipDeviceSynchronize();
        printf("D2H into them: %.1f GB/s\n", 4.0 * n * bytes / (mono() - t) / 1e9);
    }
    if (FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r")) { char b[128]; if (fgets(b, sizeof b, f)) printf("THP: %s", b); fclose(f); }
    if (FILE *f = fopen("/proc/self/smaps_rollup", "r")) { char b[256]; while (fgets(b, sizeof b, f)) if (strstr(b, "AnonHuge") || strstr(b, "Rss:")) printf("%s", b); fclose(f); }
    printf("EXIT %.6f\n", epoch()); fflush(stdout);
    _exit(0);
}
