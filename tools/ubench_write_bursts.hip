// tools/ubench_write_bursts.hip -- analysis only: how the L2 of gfx950 merges per-lane sequential output streams.
// Every lane writes its own contiguous region (like a FASTQ record) in pieces of B bytes, with ALU work between the pieces so that the
// whole grid writes at ~1 TB/s; rocprofv3 --pmc WRITE_SIZE tells how many bytes leave the L2 for each variant:
//   k<16,0>  16-byte aligned pieces      k<32,0>  32-byte aligned bursts      k<64,0>  64-byte aligned bursts     k<16,1> 16-byte pieces at odd addresses
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_write_bursts.hip -o ubench_write_bursts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed, aligned(1))) U16 { uint32_t a, b, c, d; };
template <int B, int ODD>
__global__ void __launch_bounds__(256) k(uint8_t *out, uint32_t region, uint32_t spin, uint32_t *sink)
{
    const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint8_t *p = out + lane * region + (ODD ? 5 : 0);
    uint32_t x = (uint32_t)lane * 2654435761u + 1u;
    for (uint32_t off = 0; off + B + 16 <= region; off += B) {
        for (uint32_t s = 0; s < spin * (B / 16); ++s) x = x * 1664525u + 1013904223u;      // the "normals" between the pieces
#pragma unroll
        for (int q = 0; q < B / 16; ++q) { U16 v; v.a = x; v.b = x ^ off; v.c = q; v.d = off; *reinterpret_cast<U16 *>(p + off + 16 * q) = v; }
    }
    if (x == 12345u) *sink = x;
}
int main(int argc, char **argv)
{
    const uint32_t region = 384, spin = argc > 1 ? atoi(argv[1]) : 200;
    const uint32_t blocks = 256 * 5 * 8;                    // 8 rounds of a full chip at 5 blocks per CU
    const size_t bytes = (size_t)blocks * 256 * region;
    uint8_t *out; uint32_t *sink;
    hipMalloc((void **)&out, bytes + 64); hipMalloc((void **)&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, void (*kern)(uint8_t *, uint32_t, uint32_t, uint32_t *)) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, region, spin, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) printf("%-10s %8.3f ms  %7.1f GB/s of text\n", name, ms, bytes / ms * 1e-6);
        }
        fflush(stdout);
    };
    run("a16", k<16, 0>); run("a32", k<32, 0>); run("a64", k<64, 0>); run("odd16", k<16, 1>); run("odd64", k<64, 1>);
    return 0;
}
