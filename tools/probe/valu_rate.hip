// tools/probe/valu_rate.hip -- analysis only: what one VALU instruction of a wave64 costs on gfx950, as k_simulate issues them -- integer ops with
// dependent and independent chains, at 1 .. 8 waves per SIMD.  Answers two questions DESIGN.md 5 needs: (1) how many cycles of a SIMD one such instruction
// takes (is a wave64 op 2 or 4 cycles of the SIMD-32?), (2) how SQ_ACTIVE_INST_VALU (quad-cycles) relates to that -- run under rocprofv3 --pmc.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define N_ITER 4096
// KIND 0: eight independent chains of v_xor / v_add (ILP 8); 1: one dependent chain; 2: v_mad_u64_u32 x 4 independent; 3: v_log_f32 x 4 independent;
// 4: v_bitop3 dependent pairs as in Philox (mad -> xor3 -> mad)
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint32_t *out, uint64_t *cycles, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    float f[4] = {1.5f + seed, 2.5f + seed, 3.5f + seed, 4.5f + seed};
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_ITER; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = (a[i] ^ 0x9E3779B9u) + (uint32_t)it;      // 2 VALU each: 64 per iteration
        } else if (KIND == 1) {
#pragma unroll
            for (int r = 0; r < 32; ++r) a[0] = (a[0] ^ 0x9E3779B9u) + (uint32_t)it;          // 64 dependent VALU per iteration
        } else if (KIND == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const uint64_t p = (uint64_t)a[i] * 0xD2511F53u; a[i] = (uint32_t)p; a[i + 4] ^= (uint32_t)(p >> 32); }      // 16 mad + 16 xor
        } else if (KIND == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = __builtin_amdgcn_logf(f[i]) + 3.0f;         // 16 log + 16 add
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint64_t p0 = (uint64_t)0xD2511F53u * a[0], p1 = (uint64_t)0xCD9E8D57u * a[2];
                const uint32_t n0 = (uint32_t)(p1 >> 32) ^ a[1] ^ seed, n2 = (uint32_t)(p0 >> 32) ^ a[3] ^ (seed + r);
                a[1] = (uint32_t)p1; a[3] = (uint32_t)p0; a[0] = n0; a[2] = n2;                  // one Philox round: 2 mad + 2 xor3
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    s ^= __float_as_uint(f[0] + f[1] + f[2] + f[3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *what, int valu_per_iter)
{
    int dev = 0; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    const int cus = pr.multiProcessorCount;
    uint32_t *out; uint64_t *cyc; hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8); hipMalloc(&cyc, sizeof(uint64_t) * cus * 8);
    for (int wps = 1; wps <= 8; wps *= 2) {                 // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
        const int blocks = cus * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_rate<KIND><<<blocks, 256>>>(out, cyc, 1);           // warm-up
        hipEventRecord(e0);
        k_rate<KIND><<<blocks, 256>>>(out, cyc, 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        uint64_t c0 = 0; hipMemcpy(&c0, cyc, sizeof c0, hipMemcpyDeviceToHost);
        const double insts = (double)N_ITER * valu_per_iter;             // per wave
        const double clk_mhz = pr.clockRate / 1e3;
        // time-based: cycles of the SIMD per VALU instruction = ms * clk / (insts * waves per SIMD)
        printf("%-44s waves/SIMD %d: %8.3f ms   s_memtime ticks per wave %10llu (%.2f per VALU)   SIMD cycles per VALU at %.0f MHz: %.2f\n", what, wps, ms,
               (unsigned long long)c0, (double)c0 / insts, clk_mhz, ms * 1e-3 * clk_mhz * 1e6 / (insts * wps));
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("8 independent chains of xor + add", 64);
    run<1>("one dependent chain of xor + add", 64);
    run<2>("v_mad_u64_u32 + xor, 4 independent", 32);
    run<3>("v_log_f32 + add, 4 independent", 32);
    run<4>("Philox round (2 mad + 2 xor3), dependent", 32);
    return 0;
}
