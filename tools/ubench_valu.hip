// tools/ubench_valu.hip -- analysis only (not part of the product): issue cost of the VALU instructions the k_simulate hot loops
// are made of, in SIMD cycles per wave64 instruction, measured with all SIMDs full (8 waves each) so latencies are hidden.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define KERNEL(name, decl, body, sink)                                                   \
    __global__ void __launch_bounds__(256) name(uint32_t *out, int iters, uint32_t seed) \
    {                                                                                    \
        decl;                                                                            \
        for (int i = 0; i < iters; ++i) { REP16(body) }                                  \
        sink;                                                                            \
    }

#define DECL32 uint32_t a = seed + threadIdx.x, b = a * 3 + 1, c = a ^ 0x5555, d = a + 7, e = a * 5, f = a + 11, g = a ^ 99, h = a + 123
#define SINK32 if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345678) out[threadIdx.x] = a
#define DECLF float a = seed + threadIdx.x, b = a * 3 + 1, c = a + 0.5f, d = a + 7, e = a * 5, f = a + 11, g = a + 99, h = a + 123
#define SINKF if ((a + b + c + d + e + f + g + h) == 0.12345f) out[threadIdx.x] = 1
#define DECLD double a = seed + threadIdx.x, b = a * 3 + 1, c = a + 0.5, d = a + 7, e = a * 5, f = a + 11, g = a + 99, h = a + 123
#define SINKD if ((a + b + c + d + e + f + g + h) == 0.12345) out[threadIdx.x] = 1
#define DECL64 uint64_t a = seed + threadIdx.x, b = a * 3 + 1, c = a ^ 0x5555, d = a + 7; uint32_t e = (uint32_t)a * 5, f = (uint32_t)a + 11, g = (uint32_t)a ^ 99, h = (uint32_t)a + 123
#define SINK64 if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345678) out[threadIdx.x] = (uint32_t)a

// 4 independent instructions per body => 64 per loop iteration
KERNEL(k_fma_f32, DECLF, asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));, SINKF)
KERNEL(k_mul_f32, DECLF, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINKF)
KERNEL(k_fma_f64, DECLD, asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));, SINKD)
KERNEL(k_mul_f64, DECLD, asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINKD)
KERNEL(k_add_f64, DECLD, asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINKD)
KERNEL(k_rcp_f64, DECLD, asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, SINKD)
KERNEL(k_rsq_f64, DECLD, asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, SINKD)
KERNEL(k_cvt_f64_u32, DECL64, asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));, SINK64)
KERNEL(k_cvt_f32_f64, DECL64, asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b), "v"(c), "v"(d));, SINK64)
KERNEL(k_cvt_i32_f64, DECL64, asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7" : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b), "v"(c), "v"(d));, SINK64)
KERNEL(k_log_f32, DECLF, asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, SINKF)
KERNEL(k_rcp_f32, DECLF, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, SINKF)
KERNEL(k_sqrt_f32, DECLF, asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, SINKF)
KERNEL(k_add_u32, DECL32, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINK32)
KERNEL(k_xor_b32, DECL32, asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINK32)
KERNEL(k_bitop3, DECL32, asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));, SINK32)
KERNEL(k_mul_lo_u32, DECL32, asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINK32)
KERNEL(k_mul_hi_u32, DECL32, asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINK32)
KERNEL(k_mad_u64_u32, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %5, %6, 0\n v_mad_u64_u32 %2, vcc, %6, %7, 0\n v_mad_u64_u32 %3, vcc, %7, %4, 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");, SINK64)
KERNEL(k_mul_u32_u24, DECL32, asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINK32)
KERNEL(k_mul_hi_u32_u24, DECL32, asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINK32)
KERNEL(k_mad_u32_u24, DECL32, asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));, SINK32)
KERNEL(k_perm_b32, DECL32, asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));, SINK32)
KERNEL(k_lshl_add, DECL32, asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, SINK32)
KERNEL(k_lshlrev_b64, DECL64, asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, SINK64)
KERNEL(k_cndmask, DECL32, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : );, SINK32)
KERNEL(k_cmp_lt_u32, DECL32, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e) : "vcc");, SINK32)
KERNEL(k_pk_fma_f32, DECL64, asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(a), "v"(b));, SINK64)
KERNEL(k_s_add, DECL32, asm volatile("s_add_u32 s20, s20, 5\n s_add_u32 s21, s21, 5\n s_add_u32 s22, s22, 5\n s_add_u32 s23, s23, 5" : : : "s20", "s21", "s22", "s23", "scc");, SINK32)
// mixed: one VALU + one SALU (do they co-issue from different waves?)
KERNEL(k_mix_valu_salu, DECL32, asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 5\n v_add_u32 %1, %1, %4\n s_add_u32 s21, s21, 5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21", "scc");, SINK32)

typedef void (*kern_t)(uint32_t *, int, uint32_t);
struct T { const char *name; kern_t k; int per_body; };

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double clk_ghz = pr.clockRate / 1e6;
    const int ncu = pr.multiProcessorCount;
    printf("device %s, %d CUs, %.3f GHz\n", pr.name, ncu, clk_ghz);
    uint32_t *out; hipMalloc(&out, 4096);
    T tests[] = {
        {"v_fma_f32", k_fma_f32, 4}, {"v_mul_f32", k_mul_f32, 4}, {"v_pk_fma_f32", k_pk_fma_f32, 4}, {"v_fma_f64", k_fma_f64, 4}, {"v_mul_f64", k_mul_f64, 4}, {"v_add_f64", k_add_f64, 4},
        {"v_rcp_f64", k_rcp_f64, 4}, {"v_rsq_f64", k_rsq_f64, 4}, {"v_cvt_f64_u32", k_cvt_f64_u32, 4}, {"v_cvt_f32_f64", k_cvt_f32_f64, 4}, {"v_cvt_i32_f64", k_cvt_i32_f64, 4},
        {"v_log_f32", k_log_f32, 4}, {"v_rcp_f32", k_rcp_f32, 4}, {"v_sqrt_f32", k_sqrt_f32, 4},
        {"v_add_u32", k_add_u32, 4}, {"v_xor_b32", k_xor_b32, 4}, {"v_bitop3_b32", k_bitop3, 4}, {"v_mul_lo_u32", k_mul_lo_u32, 4}, {"v_mul_hi_u32", k_mul_hi_u32, 4},
        {"v_mad_u64_u32", k_mad_u64_u32, 4}, {"v_mul_u32_u24", k_mul_u32_u24, 4}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 4}, {"v_mad_u32_u24", k_mad_u32_u24, 4},
        {"v_perm_b32", k_perm_b32, 4}, {"v_lshl_add_u32", k_lshl_add, 4}, {"v_lshlrev_b64", k_lshlrev_b64, 4}, {"v_cndmask_b32", k_cndmask, 4}, {"v_cmp_lt_u32", k_cmp_lt_u32, 4},
        {"s_add_u32", k_s_add, 4}, {"v_add+s_add", k_mix_valu_salu, 4},
    };
    const int iters = 2000;
    for (int wps : {8, 2, 1}) {      // waves per SIMD
        printf("---- %d wave(s) per SIMD ----\n", wps);
        for (auto &t : tests) {
            const int blocks = ncu * wps;          // 256-thread blocks = 4 waves = one per SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 10, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)wps * iters * 16 * t.per_body;     // wave-instructions per SIMD
            printf("%-18s %8.3f ms   %6.2f cycles / wave-instruction / SIMD\n", t.name, ms, ms * 1e-3 * clk_ghz * 1e9 / instr_per_simd);
        }
    }
    return 0;
}
