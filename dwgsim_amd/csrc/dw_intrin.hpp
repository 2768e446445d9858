// dw_intrin.hpp -- the gfx950 (CDNA4) instructions, address spaces and hardware-operation sequences the kernels are written with, one small function
// each.  This is the product's header; tests/emu/dw_intrin.hpp is its stand-in for the test-only CPU emulation of the kernels (plain C++ with the
// same results), found first by that build's include path.  Nothing else in dwgsim_amd/csrc knows about the emulation.
#pragma once
#include <stdint.h>

#ifndef DW_DEV
#define DW_DEV __device__ __forceinline__
#endif
// The rare paths of k_simulate (the ragged first / last unit of a record, the exact fp64 quality try).  Inlined at every use they make up a third of
// the kernel's code (94 KB against 57 KB when they are called); measured on one box the called form is 1 % SLOWER at 2 x 150 -o 1 and 0.5 % faster
// at -o 0 (profiles/r05_bench_lines_final.txt): the instruction cache is not what holds the kernel back.  -DDW_CALL_RARE builds the called form.
#ifdef DW_CALL_RARE
#define DW_DEV_NOINLINE __device__ __attribute__((noinline))
#else
#define DW_DEV_NOINLINE __device__ __forceinline__
#endif
// dynamic LDS of a kernel
#define DW_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
// A pointer into memory that nothing writes while the kernel runs (tables the host uploaded before the launch), in the constant address space:
// loads through it at a wave-uniform address are SCALAR loads (s_load), their results live in scalar registers.  Through a plain global pointer
// the compiler must assume that the kernel's own stores and atomics could have clobbered the table and falls back to vector loads.
#define DW_CONST_AS __attribute__((address_space(4)))

namespace dw {

// a ^ b ^ c in one instruction: v_bitop3_b32 (truth table 0x96)
DW_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// a value that is the same in every lane of the wave, moved to a scalar register (what depends on it -- table lookups, loop bounds -- stays scalar)
DW_DEV uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// byte-wise table lookup: byte i of the result = byte sel.byte[i] (0..7) of the eight-byte table {hi, lo}; one v_perm_b32
DW_DEV uint32_t lut8(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// two wave-uniform values pinned to scalar registers at this point (an optimisation barrier: the compiler may not hoist what depends on them)
DW_DEV void keep_scalar(uint32_t &a, uint32_t &b) { asm volatile("" : "+s"(a), "+s"(b)); }

// the XCD (0..7) this wave runs on: s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4).  The eight XCDs have an L2 each and the L2s are not coherent with each
// other, so memory that is handed from workgroup to workgroup INSIDE a launch (the Ion Torrent scratch slots) stays within one XCD
DW_DEV uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }
// every store this wave has issued has been acknowledged by the L2 (a workgroup ends without waiting for its stores)
DW_DEV void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// issue priority of this wave among the waves of its SIMD (s_setprio 0 .. 3)
DW_DEV void wave_priority(int p) { if (p) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }

// IEEE-754 binary64 x / y and sqrt(x) for operands far from the ends of the exponent range and without special values:
// exactly the Newton-Raphson + correction sequences the compiler emits for `/` and sqrt() on gfx950 (LLVM AMDGPU LowerFDIV64 /
// lowerFSQRTF64) minus their v_div_scale / v_div_fixup / ldexp / class-test range handling, which is the identity on such
// operands.  Used where the operand range is known (quality normals: y, x in [2^-62, 2^70]); dwgsim_hip_selftest_fp64 compares
// them bit for bit with the compiler's own `/` and sqrt() (tests/test_gpu_parity.py).
DW_DEV double div_mid(double x, double y)
{
    const double r0 = __builtin_amdgcn_rcp(y);
    const double r1 = __builtin_fma(r0, __builtin_fma(-y, r0, 1.0), r0);
    const double r2 = __builtin_fma(r1, __builtin_fma(-y, r1, 1.0), r1);
    const double q0 = x * r2;
    return __builtin_fma(__builtin_fma(-y, q0, x), r2, q0);
}
DW_DEV double sqrt_mid(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double g2 = __builtin_fma(__builtin_fma(-g1, g1, x), h1, g1);
    return __builtin_fma(__builtin_fma(-g2, g2, x), h1, g2);
}

} // namespace dw
