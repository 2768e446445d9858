// dw_common.hpp -- device-side primitives of the MI355X dwgsim hot path:
// Philox4x32-10 counter RNG, the 53-bit uniform, the pure-IEEE natural log, agent-scope
// status words and wave64 helpers.  gfx950 only (wave = 64 lanes).
//
// RNG layout (DESIGN.md "RNG layout"; this is the product's own definition of the
// "matched RNG" stream that replaces the reference's sequential drand48):
//   key     = (uint32 seed, uint32 contig_index)
//   counter = ( index[31:0],  index[47:32] | retry << 16,  domain << 24 | attempt,  block )
//   block b yields two uniforms: slot 2b = u53(w0,w1), slot 2b+1 = u53(w2,w3),
//   u53(hi,lo) = ((hi << 21) | (lo >> 11)) * 2^-53   (exact in fp64).
#pragma once
#include <stdint.h>

#define DW_DEV __device__ __forceinline__
// the gfx950 instructions and address spaces this code is written for -- the only header with a stand-in elsewhere (tests/emu/dw_intrin.hpp, found
// first by the include path of the test-only CPU emulation build: test infrastructure, never part of the product)
#include <dw_intrin.hpp>
#include <dw_probe.hpp>

namespace dw {

// domains
enum : uint32_t {
    D_WALK = 1,         // index = position.  slot 0 deletion-extend (mut.c:611), 1 mutate (mut.c:618),
                        // 2 substitution-vs-indel (:619), 3 new base (:620) / deletion-vs-insertion (:628),
                        // 4 hom test (:622,:629, mut.c:301), 5 het haplotype (:625,:633, mut.c:303)
    D_WALK_INSLEN = 2,  // slot k = k-th insertion length-extension test (mut.c:292)
    D_WALK_INSBASE = 3, // slot k = k-th inserted-base draw (mut.c:314 / :351)
    D_WALK_SITE = 7,    // index = window q of 256 positions of the contig.  NARROW words: word m = the m-th GAP between candidate sites ("mutate this position?",
                        // mut.c:618) inside the window: candidates at 256 q + S_m, S_0 = G_0, S_(m+1) = S_m + 1 + G_(m+1) (geom_gap below), while S_m < 256
    D_WALK_SITE_REF = 26, // (rounds 2-5: the low halves of per-position site draws; unused)
    D_PAIR = 4,         // index = ii.  slot 0 random-read test (dwgsim.c:649), 1 haplotype (:716), 2 strand (:723)
    D_PLACE = 5,        // slot t = position uniform of placement try t (dwgsim.c:671)
    D_PLACE_NORM = 6,   // block t, retry r = polar tries of the insert-size normal of try t (dwgsim.c:657)
    // NARROW domains: a draw is one 32-bit word w of a block, u = w * 2^-32, four draws per Philox block
    D_BASE0 = 8,        // +read end.  Genomic reads: NARROW words, word m = the m-th GAP between error sites of the read end (dwgsim.c:237 `drand48() < e[i]`): sites
                        // S_0 = G_0, S_(m+1) = S_m + 1 + G_(m+1) < s, the chain at the LARGEST threshold of the ramp (geom_gap below; dw_simulate.hip).
                        // Random reads (:1000): base i = the 2-bit field i of the stream (bits 2 (i & 15) of word (i >> 4) & 3 of block i >> 6)
    D_QUAL0 = 10,       // +read end.  the sequential stream of polar tries of the read's quality normals (dwgsim.c:912, :156-175), 16-BIT uniforms: try t =
                        // the two halves of word t & 3 of block t >> 2 (low half v1, high half v2); every accepted try delivers two normals (v2*fac,
                        // then the cached v1*fac)
    D_FLOW0 = 12,       // +read end.  generate_errors_flows (dwgsim.c:246-417): NARROW words, word m = the m-th GAP between scoring FIRST draws of pass 1 (the
                        // first draws of the homopolymer starts, in the order they are examined; + D_FLOW_PASS2: of the empty flows of pass 2) -- geom_gap
                        // below; every further draw of an event: its private stream + D_FLOW_EV
    D_FLOW_REF = 32,    // (rounds 2-5: the low halves of per-event first uniforms; unused)
    D_FLOW_EV = 64,     // added to a flow-model domain: draw s of event h = word s & 3 of the block (retry s >> 2, block h)
    D_FLOW_PASS2 = 8,   // added to D_FLOW0 / D_CALIB (+read end) for the second pass of the flow model (domains 20-23)
    D_CALIB = 14,       // +read end.  -B calibration (dwgsim_opt.c:415-457): index = random read; attempt 0 = its bases, attempt 1 = its flow-model stream
    D_SUB0 = 16,        // +read end.  word m: substituted-base draw of error site m of the read end's chain (dwgsim.c:238): drawn with the chain, a block per four sites
    D_BASE_REF0 = 24    // +read end.  word m = the THINNING draw of error site m where the position's threshold thr_i is below the ramp's largest: the site is kept
                        // iff w * thr_max < thr_i * 2^32 (probability thr_i / thr_max); not drawn for a constant error rate
};

struct U4 { uint32_t x, y, z, w; };

template <class T> DW_DEV const DW_CONST_AS T *as_constant(const T *p) { return (const DW_CONST_AS T *)p; }      // (DW_CONST_AS: dw_intrin.hpp)

// the four nibbles of v[15:0] spread into the four bytes of the result
DW_DEV uint32_t spread4(uint32_t v)
{
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    return (v | (v << 4)) & 0x0F0F0F0Fu;
}

// UNIFORM_KEY: the key is the same in every lane of the wave (scalar registers); the read kernels' keys are, the walk kernels' per-candidate keys
// (a contig of a group per thread) are not
#ifndef DW_PHILOX_ROUNDS
#define DW_PHILOX_ROUNDS 10      // (analysis builds only: the cost of the generator's rounds)
#endif
template <bool UNIFORM_KEY = true>
DW_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < DW_PHILOX_ROUNDS; ++r) {
        if (UNIFORM_KEY) keep_scalar(k0, k1);      // keep the round keys out of 20 hoisted SGPRs: two SALU adds per round instead
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

struct RngKey { uint32_t seed, contig; };

template <bool UNIFORM_KEY = true>
DW_DEV U4 rng_block(RngKey k, uint32_t dom, uint64_t idx, uint32_t att, uint32_t retry, uint32_t block)
{
    return philox4x32_10<UNIFORM_KEY>((uint32_t)idx, (uint32_t)((idx >> 32) & 0xFFFFu) | (retry << 16), (dom << 24) | (att & 0xFFFFFFu), block, k.seed, k.contig);
}

DW_DEV double u53(uint32_t hi, uint32_t lo)
{
    return (double)(((uint64_t)hi << 21) | (uint64_t)(lo >> 11)) * 0x1p-53;
}
DW_DEV double u_lo(const U4 &b) { return u53(b.x, b.y); }   // even slot of the block
DW_DEV double u_hi(const U4 &b) { return u53(b.z, b.w); }   // odd slot of the block

template <bool UNIFORM_KEY = true>
DW_DEV double rng_slot(RngKey k, uint32_t dom, uint64_t idx, uint32_t att, uint32_t slot)
{
    const U4 b = rng_block<UNIFORM_KEY>(k, dom, idx, att, 0, slot >> 1);
    return (slot & 1) ? u_hi(b) : u_lo(b);
}

DW_DEV uint64_t dbl_bits(double x) { union { double d; uint64_t u; } c; c.d = x; return c.u; }
DW_DEV double bits_dbl(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }

// IEEE-754 binary64 x / y and sqrt(x) for operands far from the ends of the exponent range and without special values:
// exactly the Newton-Raphson + correction sequences the compiler emits for `/` and sqrt() on gfx950 (LLVM AMDGPU LowerFDIV64 /
// lowerFSQRTF64) minus their v_div_scale / v_div_fixup / ldexp / class-test range handling, which is the identity on such
// operands.  Used where the operand range is known (quality normals: y, x in [2^-62, 2^70]); dwgsim_hip_selftest_fp64 compares
// them bit for bit with the compiler's own `/` and sqrt() (tests/test_gpu_parity.py).
// (div_mid, sqrt_mid: dw_intrin.hpp)

// Natural log for finite x > 0 with only IEEE-754 fp64 + - * / (compile with -ffp-contract=off):
// argument reduction x = 2^k (1+f), s = f/(2+f), even polynomial in s -- the classic fdlibm
// e_log algorithm ("(c) 1993 Sun Microsystems, Inc. Permission to use, copy, modify, and
// distribute this software is freely granted, provided that this notice is preserved").
// Bit-identical on gfx950 and x86-64; the libm/ocml logs are not.
// NORMAL = true: the caller guarantees a normal (not subnormal) argument, the 2^54 pre-scaling test is dropped (same bits for such x)
template <bool NORMAL = false>
DW_DEV double det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t b = dbl_bits(x);
    int32_t hx = (int32_t)(b >> 32), k = 0;
    if (!NORMAL && hx < 0x00100000) { x *= 0x1p54; k = -54; b = dbl_bits(x); hx = (int32_t)(b >> 32); }
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = bits_dbl(((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (b & 0xFFFFFFFFull));
    k += (i >> 20);
    const double f = x - 1.0, dk = (double)k;
    const double s = NORMAL ? div_mid(f, 2.0 + f) : f / (2.0 + f);       // 2 + f in [1.7, 2.42], f = 0 or |f| >= 2^-53
    const double z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    i = (hx - 0x6147a) | (0x6b851 - hx);
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// ---- geometric gaps (round 6).  A sequence of independent "does this one score?" draws with probability e' = thr / 2^32 is, in law, its gaps: the quiet
// draws in front of each scoring one are Geometric(e').  G = floor(-log2(U) / -log2(1 - e')), U = (2 w + 1) / 2^33, in integer arithmetic only, so that every
// compiler agrees: lg = the host's 257-entry table floor(2^32 log2(1 + i / 256)) (linear interpolation), -log2(U) in Q8.56, one 64 x 64 -> 128 multiplication by
// R = 2^127 / (normalised -log2(1 - e')) and a shift (dw_kernels.hpp flow_gap_params).  Used by the flow model's first draws and the walk's site draws. ----
DW_DEV uint32_t geom_gap(uint32_t w, const uint32_t *lg, uint64_t R, int sR)
{
    const uint64_t X = ((uint64_t)w << 1) | 1ull;
    const int p = 63 - __clzll((long long)X);
    const uint64_t M = X << (63 - p);
    const uint32_t idx = (uint32_t)(M >> 55) & 0xFFu, r16 = (uint32_t)(M >> 39) & 0xFFFFu;
    const uint32_t t0 = lg[idx], t1 = lg[idx + 1];
    const uint32_t f = t0 + (uint32_t)(((uint64_t)(t1 - t0) * r16) >> 16);
    const uint64_t Lu = ((uint64_t)(33 - p) << 56) - ((uint64_t)f << 24);
    const uint64_t G = __umul64hi(Lu, R) >> sR;
    return G > 0x3FFFFFFFull ? 0x3FFFFFFFu : (uint32_t)G;
}

// ---- wave64 helpers (all 64 lanes must call) ----
DW_DEV int lane_id() { return (int)(threadIdx.x & 63); }

DW_DEV uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d); if (lane_id() >= d) v += o; }
    return v;
}
DW_DEV uint64_t shfl_down_u64(uint64_t v, int d)
{
    uint32_t lo = __shfl_down((uint32_t)v, d), hi = __shfl_down((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
DW_DEV uint64_t wave_sum_u64(uint64_t v)   // result valid in lane 0, broadcast to all
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += shfl_down_u64(v, d);
    uint32_t lo = __shfl((uint32_t)v, 0), hi = __shfl((uint32_t)(v >> 32), 0);
    return ((uint64_t)hi << 32) | lo;
}
DW_DEV uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    return __shfl(v, 0);
}

// ---- decoupled look-back status words: one naturally aligned 8-byte {flag:2, value:62},
// written and read with relaxed agent-scope atomics (no payload besides the word itself) ----
constexpr uint64_t ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_VAL = (1ull << 62) - 1;

DW_DEV void status_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DW_DEV uint64_t status_load(uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Exclusive prefix of `aggregate` over logical blocks 0..t-1 (+ base).  Called by one whole wave;
// logical block ids come from an atomic ticket, so every predecessor has already started.
// A hop looks at LB_W x 64 predecessors.  Round 6 tried wider hops on the theory that the 0.8 ms a 2 x 150 launch spends at its look-backs (a look-back that
// never waits: 5.34 -> 4.54 ms, also without a single retry in the batch) are the dependent round trips of the hops: 128 / 256 / 512 predecessors per hop
// make the launch SLOWER (1 238 -> 1 201 / 1 143 / 1 070 M pairs/s, profiles/r06_bench_lines_final.txt): more polling traffic on the same words, no fewer
// waits -- the time is spent waiting for predecessors to publish, not hopping.  LB_W stays 1.
#ifndef DW_LB_SLEEP
#define DW_LB_SLEEP 2      // (s_sleep units of 64 cycles between two polls of a predecessor that has not published)
#endif
#ifndef DW_LB_W
#define DW_LB_W 1
#endif
constexpr int LB_W = DW_LB_W;
DW_DEV uint64_t lookback_excl(uint64_t *status, uint32_t t, uint64_t aggregate, uint64_t base)
{
    const int lane = lane_id();
    if (t == 0) { if (lane == 0) status_store(&status[0], ST_PREFIX | ((base + aggregate) & ST_VAL)); return base; }
    if (lane == 0) status_store(&status[t], ST_AGG | (aggregate & ST_VAL));
    uint64_t excl = 0;
    int64_t k = (int64_t)t - 1;
    for (;;) {
        // this lane's LB_W predecessors, nearest first: k - lane * LB_W - w
        uint64_t v[LB_W];
#pragma unroll
        for (int w = 0; w < LB_W; ++w) {
            const int64_t idx = k - (int64_t)lane * LB_W - w;
            v[w] = idx >= 0 ? status_load(&status[idx]) : ST_PREFIX;                      // below block 0: an empty prefix
        }
#pragma unroll
        for (int w = 0; w < LB_W; ++w) {
            const int64_t idx = k - (int64_t)lane * LB_W - w;
            if (probe::off(2048) && (v[w] >> 62) == 0) v[w] = ST_PREFIX;                  // (probe 2048: a look-back that never waits -- garbage offsets, analysis only)
            while ((v[w] >> 62) == 0) { __builtin_amdgcn_s_sleep(DW_LB_SLEEP); v[w] = status_load(&status[idx]); }
        }
        uint64_t mine = 0; bool have = false;                                            // the values up to and including this lane's nearest PREFIX
#pragma unroll
        for (int w = 0; w < LB_W; ++w) { if (!have) mine += v[w] & ST_VAL; have = have || (v[w] >> 62) == 2; }
        const uint64_t pm = __ballot(have);
        const int first = pm ? (__ffsll((unsigned long long)pm) - 1) : 64;
        excl += wave_sum_u64(lane <= first ? mine : 0);
        if (pm) break;
        k -= 64 * LB_W;
    }
    if (lane == 0) status_store(&status[t], ST_PREFIX | ((excl + aggregate) & ST_VAL));
    return excl;
}

} // namespace dw
