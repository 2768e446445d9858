// tests/emu/dw_intrin.hpp -- TEST INFRASTRUCTURE ONLY: the stand-in for dwgsim_amd/csrc/dw_intrin.hpp in the CPU emulation build of the kernels
// (tests/emu/build.sh puts this directory first on the include path).  Plain C++ with the same results as the gfx950 instructions.
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef DW_DEV
#define DW_DEV __device__ __forceinline__
#endif
#define DW_DEV_NOINLINE __device__ __forceinline__
#define DW_DYN_SHARED(type, name) type *name = (type *)hipemu::dyn_shared()
#define DW_CONST_AS

namespace dw {
DW_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
DW_DEV uint32_t uniform_u32(uint32_t v) { return v; }
DW_DEV uint32_t lut8(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t t = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((t >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
DW_DEV void keep_scalar(uint32_t &, uint32_t &) {}
DW_DEV uint32_t xcc_id() { return (uint32_t)blockIdx.x & 7u; }      // block b runs on XCD b % 8 (what the hardware is observed to do)
DW_DEV void wait_stores() {}
DW_DEV void wave_priority(int) {}
DW_DEV double div_mid(double x, double y) { return x / y; }
DW_DEV double sqrt_mid(double x) { return sqrt(x); }
} // namespace dw
