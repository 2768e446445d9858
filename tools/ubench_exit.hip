// tools/ubench_exit.hip -- analysis only: what a process that holds page-locked host memory, device memory and a mapped file costs to START
// (hipHostMalloc / hipMalloc times) and to END (time from _exit to the parent's wait returning: the kernel tearing the address space down).
//   ubench_exit <n_host_bufs> <host_mb_each> <dev_mb> [file to map]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
static double mono() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double epoch() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 0; const size_t hmb = argc > 2 ? (size_t)atol(argv[2]) : 0, dmb = argc > 3 ? (size_t)atol(argv[3]) : 0;
    double t = mono();
    hipFree(nullptr);
    printf("runtime init %.3f s\n", mono() - t);
    for (int i = 0; i < nb; ++i) {
        void *p = nullptr; t = mono();
        if (hipHostMalloc(&p, hmb << 20, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
        const double a = mono() - t; t = mono();
        memset(p, 1, hmb << 20);
        printf("hipHostMalloc %zu MB: %.4f s, first touch %.4f s\n", hmb, a, mono() - t);
    }
    if (dmb) {
        void *d = nullptr; t = mono();
        if (hipMalloc(&d, dmb << 20) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        const double a = mono() - t; t = mono();
        hipMemset(d, 1, dmb << 20); hipDeviceSynchronize();
        printf("hipMalloc %zu MB: %.4f s, memset %.4f s\n", dmb, a, mono() - t);
    }
    if (argc > 4) {
        const int fd = open(argv[4], O_RDONLY); struct stat st; fstat(fd, &st);
        t = mono();
        const unsigned char *m = (const unsigned char *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        unsigned long s = 0; for (off_t o = 0; o < st.st_size; o += 4096) s += m[o];
        printf("mapped + touched %.0f MB: %.4f s (%lu)\n", st.st_size / 1e6, mono() - t, s);
    }
    printf("EXIT %.6f\n", epoch()); fflush(stdout);
    _exit(0);
}
