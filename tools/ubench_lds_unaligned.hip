// tools/ubench_lds_unaligned.hip -- analysis only: do unaligned ds_write_b64 / ds_write_b32 / ds_read_b32 (what hipcc emits for align(1) LDS
// accesses on gfx950) really land the right bytes at every byte offset, and what do they cost next to aligned ones?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint16_t __attribute__((aligned(1))) u16u;
constexpr int STRIDE = 80;
__global__ void k_check(uint8_t *out, int nrec)     // every lane appends nrec pseudo-random-length pieces to its 80-byte ring, flushing 64-byte chunks to out
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *mine = lds + threadIdx.x * STRIDE;
    uint8_t *g = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4096;
    uint32_t n = 0, x = threadIdx.x * 2654435761u + 12345u + blockIdx.x; uint64_t ctr = 0;
    for (int r = 0; r < nrec; ++r) {
        x = x * 1664525u + 1013904223u;
        const uint32_t cnt = 1 + (x >> 29);                 // 1..8 bytes
        uint64_t v = 0; for (uint32_t b = 0; b < cnt; ++b) v |= (uint64_t)((ctr + b) & 0xff) << (8 * b);
        ctr += cnt;
        if (cnt == 1) mine[n] = (unsigned char)v; else if (cnt <= 4 && (x & 1)) *(u32u *)(mine + n) = (uint32_t)v; else *(u64u *)(mine + n) = v;
        n += cnt;
        if (n >= 64) {
            for (int q = 0; q < 4; ++q) reinterpret_cast<uint4 *>(g)[q] = reinterpret_cast<const uint4 *>(mine)[q];
            *reinterpret_cast<uint4 *>(mine) = *reinterpret_cast<const uint4 *>(mine + 64);
            g += 64; n -= 64;
        }
    }
    for (uint32_t q = 0; q < n; ++q) g[q] = mine[q];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 * 64 * gridDim.x] = 1;
}
template <int MODE>
__global__ void k_time(uint32_t *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // MODE 0: aligned b64, per-lane ring position advancing by 8; 1: unaligned b64 (odd start), advancing by 8; 2: unaligned b64 advancing by 5;
    // 3: aligned b32 advancing by 4; 4: unaligned b32 advancing by 3; 5: b8 advancing by 1; 6: aligned b128 read of the ring (4 per iteration)
    const uint32_t base = (uint32_t)(uintptr_t)(lds + threadIdx.x * STRIDE) & 0xffffu;      // LDS byte address of this lane's ring
    uint32_t n = MODE == 0 || MODE == 3 || MODE == 6 ? (threadIdx.x * 8) & 63 : (threadIdx.x * 7 + 1) & 63; uint64_t v = threadIdx.x; uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t addr = base + n;
            if (MODE <= 2) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory");
            else if (MODE <= 4) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"((uint32_t)v) : "memory");
            else if (MODE == 5) asm volatile("ds_write_b8 %0, %1" :: "v"(addr), "v"((uint32_t)v) : "memory");
            else { uint4 r; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base + (n & 48)) : "memory"); acc += r.x; }
            n = (n + (MODE == 0 || MODE == 1 ? 8 : MODE == 2 ? 5 : MODE == 3 ? 4 : MODE == 4 ? 3 : MODE == 5 ? 1 : 16)) & 63;
            v += 0x0101010101010101ull;
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = *(uint32_t *)(lds + threadIdx.x * STRIDE) + (uint32_t)n + acc;
}
int main()
{
    const int blocks = 64, nrec = 700;
    uint8_t *d; const size_t sz = (size_t)4096 * 64 * blocks + 64;
    hipMalloc(&d, sz); hipMemset(d, 0, sz);
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(64), 64 * STRIDE, 0, d, nrec);
    std::vector<uint8_t> h(sz); hipMemcpy(h.data(), d, sz, hipMemcpyDeviceToHost);
    long bad = 0, total = 0;
    for (int b = 0; b < blocks; ++b) for (int t = 0; t < 64; ++t) {
        uint32_t x = (uint32_t)t * 2654435761u + 12345u + (uint32_t)b; uint64_t ctr = 0;
        for (int r = 0; r < nrec; ++r) { x = x * 1664525u + 1013904223u; ctr += 1 + (x >> 29); }
        const uint8_t *g = h.data() + (size_t)(b * 64 + t) * 4096;
        for (uint64_t q = 0; q < ctr; ++q) { bad += g[q] != (uint8_t)(q & 0xff); ++total; }
    }
    printf("unaligned LDS ring check: %ld bytes, %ld wrong\n", total, bad);
    uint32_t *o; hipMalloc(&o, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[7] = {"aligned b64 (+8)", "unaligned b64 (+8)", "unaligned b64 (+5)", "aligned b32 (+4)", "unaligned b32 (+3)", "b8 (+1)", "aligned b128 read"};
    for (int mode = 0; mode < 7; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_time<0>, dim3(2048), dim3(256), 256 * STRIDE, 0, o, 500); break;
            case 1: hipLaunchKernelGGL(k_time<1>, dim3(2048), dim3(256), 256 * STRIDE, 0, o, 500); break;
            case 2: hipLaunchKernelGGL(k_time<2>, dim3(2048), dim3(256), 256 * STRIDE, 0, o, 500); break;
            case 3: hipLaunchKernelGGL(k_time<3>, dim3(2048), dim3(256), 256 * STRIDE, 0, o, 500); break;
            case 4: hipLaunchKernelGGL(k_time<4>, dim3(2048), dim3(256), 256 * STRIDE, 0, o, 500); break;
            case 5: hipLaunchKernelGGL(k_time<5>, dim3(2048), dim3(256), 256 * STRIDE, 0, o, 500); break;
            default: hipLaunchKernelGGL(k_time<6>, dim3(2048), dim3(256), 256 * STRIDE, 0, o, 500); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = 2048.0 * 4 * 500 * 16 / 256;     // wave-level LDS instructions per CU
        printf("%-22s %8.3f ms  %6.1f cycles (2.4 GHz) per wave-level instruction per CU\n", names[mode], ms, ms * 1e-3 * 2.4e9 / ops);
    }
    return bad != 0;
}
