// tools/ubench_select.hip -- analysis only (not part of the product): what a select costs on gfx950.  tools/ubench_valu.hip measured
// v_cndmask_b32 at 22.7 SIMD cycles per wave64 instruction (everything else: 2.5 - 4.5).  This looks closer: VCC or an SGPR pair as the
// condition, next to other instructions, fed by a compare, and the alternatives (v_bfi_b32 with a mask register, arithmetic).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_select.hip -o /tmp/ubench_select && /tmp/ubench_select
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define KERNEL(name, body)                                                                \
    __global__ void __launch_bounds__(256) name(uint32_t *out, int iters, uint32_t seed)  \
    {                                                                                     \
        uint32_t a = seed + threadIdx.x, b = a * 3 + 1, c = a ^ 0x5555, d = a + 7, e = a * 5, f = a + 11, g = a ^ 99, h = a + 123; \
        for (int i = 0; i < iters; ++i) { REP16(body) }                                   \
        if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345678) out[threadIdx.x] = a;          \
    }

KERNEL(k_cnd_vcc, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_cnd_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");)
// independent destinations (no read-modify-write chain on the same register)
KERNEL(k_cnd_indep, asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
// compare feeding a select (the usual pair)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc" : "+v"(a), "+v"(b) : "v"(e), "v"(f) : "vcc");)
KERNEL(k_cmp_cnd_sgpr, asm volatile("v_cmp_lt_u32 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cmp_lt_u32 s[22:23], %1, %4\n v_cndmask_b32_e64 %1, %1, %5, s[22:23]" : "+v"(a), "+v"(b) : "v"(e), "v"(f) : "s20", "s21", "s22", "s23");)
// two selects between two adds
KERNEL(k_cnd_add, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
// arithmetic stand-ins: the borrow of a subtraction as a mask (sub, ashr, and-or as bfi)
KERNEL(k_arith, asm volatile("v_sub_u32 %2, %0, %4\n v_ashrrev_i32 %2, 31, %2\n v_bfi_b32 %0, %2, %5, %0\n v_add_u32 %1, %1, %4" : "+v"(a), "+v"(b), "+v"(c) : "v"(e), "v"(f));)
KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
// what the compiler makes of `x = c ? y : x` chains
__global__ void __launch_bounds__(256) k_cxx(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = a * 3 + 1, c = a ^ 0x5555, d = a + 7, e = a * 5, f = a + 11;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { a = a < e ? a + f : a; b = b < e ? b ^ f : b; c = c < e ? c + 3 : c; d = d < e ? d ^ 5 : d; e += 7; }
    }
    if ((a ^ b ^ c ^ d ^ e) == 0x12345678) out[threadIdx.x] = a;
}

typedef void (*kern_t)(uint32_t *, int, uint32_t);
struct T { const char *name; kern_t k; int per_body; };
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double clk_ghz = pr.clockRate / 1e6; const int ncu = pr.multiProcessorCount;
    printf("device %s, %d CUs, %.3f GHz\n", pr.name, ncu, clk_ghz);
    uint32_t *out; hipMalloc(&out, 4096);
    T tests[] = {{"v_add_u32 x4", k_add, 4}, {"cndmask vcc (rmw) x4", k_cnd_vcc, 4}, {"cndmask s[20:21] x4", k_cnd_sgpr, 4}, {"cndmask vcc, new dst x4", k_cnd_indep, 4}, {"v_bfi_b32 x4", k_bfi, 4},
                 {"cmp+cndmask vcc x2", k_cmp_cnd, 4}, {"cmp+cndmask sgpr x2", k_cmp_cnd_sgpr, 4}, {"cndmask,add,cndmask,add", k_cnd_add, 4}, {"sub,ashr,bfi,add", k_arith, 4},
                 {"C++ a<e?a+f:a (x4 per k)", k_cxx, 4}};
    const int iters = 2000;
    for (int wps : {8, 2}) {
        printf("---- %d wave(s) per SIMD: SIMD cycles per wave instruction ----\n", wps);
        for (auto &t : tests) {
            const int blocks = ncu * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 10, 1u); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-28s %8.3f ms   %6.2f\n", t.name, ms, ms * 1e-3 * clk_ghz * 1e9 / ((double)wps * iters * 16 * t.per_body));
        }
    }
    return 0;
}
