// dw_device.hpp -- device helpers shared by the mutation-walk kernels (dw_walk.hip) and the read-simulation kernels
// (dw_simulate.hip): base codes, block-wide scans, insertion-table lookup.
#pragma once
#include <hip/hip_runtime.h>
#include "dw_common.hpp"
#include "dw_kernels.hpp"

namespace dw {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
DW_DEV uint32_t code_of_ascii(uint32_t ch)          // dwgsim.c:56-73 nst_nt4_table
{
    const uint32_t u = ch | 0x20u;
    return u == 'a' ? 0u : u == 'c' ? 1u : u == 'g' ? 2u : u == 't' ? 3u : (ch == '-' ? 5u : 4u);
}

// block-wide exclusive scan of one uint32 per thread (blockDim multiple of 64, <= 1024); returns the
// exclusive prefix, *total gets the block sum.  `sm` = 17 words of LDS scratch.
DW_DEV uint32_t block_excl_scan(uint32_t v, uint32_t *sm, uint32_t *total)
{
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    const uint32_t inc = wave_incl_scan(v);
    if (nw == 1) { *total = (uint32_t)__shfl((int)inc, 63); return inc - v; }      // single-wave block: no LDS, no barrier
    __syncthreads();                       // protect sm from a previous use
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int w = 0; w < nw; ++w) { uint32_t t = sm[w]; sm[w] = run; run += t; } sm[16] = run; }
    __syncthreads();
    *total = sm[16];
    return inc - v + sm[wave];
}

// N independent block-wide exclusive scans behind ONE barrier.  sm = N x 16 words of LDS that nothing else in the kernel touches
// (no protective barrier before the write, no serial pass: every thread adds up the wave totals below its own wave).
template <int N>
DW_DEV void block_excl_scan_n(const uint32_t (&v)[N], uint32_t (*sm)[16], uint32_t (&excl)[N], uint32_t (&total)[N])
{
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    uint32_t inc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { inc[n] = wave_incl_scan(v[n]); if (lane == 63) sm[n][wave] = inc[n]; }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        uint32_t base = 0, tot = 0;
        for (int w = 0; w < nw; ++w) { const uint32_t t = sm[n][w]; tot += t; base += w < wave ? t : 0u; }
        excl[n] = inc[n] - v[n] + base; total[n] = tot;
    }
}

// the first entry of the (sorted) insertion table at or behind pos.  A read that crosses an INSERT cell looks its bases up here while every block behind its own waits for
// that block's sizes (dw_simulate.hip, "one look-back"): a bisection over a few thousand entries is a dozen DEPENDENT loads.  Insertions lie about evenly over a
// contig, so the search interpolates (every other step bisects: the bound of a bisection, twice over, whatever the table looks like): four or five loads, the
// table's two ends fetched together first.  Which entries it looks at changes nothing about what it finds.
DW_DEV uint32_t ins_find(const HapDev &h, int64_t pos)
{
    pos += h.pos_off;                      // the table holds group coordinates
    const uint32_t n = h.n_ins;
    if (n == 0) return 0;
    int64_t vlo = (int64_t)h.ins_pos[0], vhi = (int64_t)h.ins_pos[n - 1];
    if (pos <= vlo) return 0;
    if (pos > vhi) return n;
    uint32_t lo = 0, hi = n - 1;           // invariant: ins_pos[lo] < pos <= ins_pos[hi]
    bool bisect = false;
    while (hi - lo > 1) {
        uint32_t g = lo + ((hi - lo) >> 1);
        if (!bisect) {
            const float f = (float)(pos - vlo) / (float)(vhi - vlo);
            g = lo + (uint32_t)((float)(hi - lo) * f);
            g = g <= lo ? lo + 1 : g >= hi ? hi - 1 : g;
        }
        const int64_t v = (int64_t)h.ins_pos[g];
        if (v < pos) { lo = g; vlo = v; } else { hi = g; vhi = v; }
        bisect = !bisect;
    }
    return hi;
}

// which contig of the group holds group coordinate g (the last k with start[k] <= g; a position in the padding behind a contig belongs to it)
DW_DEV uint32_t seg_of(const SegTab &t, int64_t g)
{
    uint32_t lo = 0, hi = (uint32_t)t.n;   // invariant: start[lo] <= g < start[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)t.start[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}

DW_DEV uint32_t count_of(Count c) { if (!c.dev) return c.host; const uint64_t n = *c.dev; return n < (uint64_t)c.host ? (uint32_t)n : c.host; }
// walk kernels: the insertion-table sizes come from the device while the host has not read them back yet
DW_DEV void adopt_device_sizes(ContigDev &c) { if (c.tot4) { c.hap[0].n_ins = c.tot4[0]; c.hap[1].n_ins = c.tot4[2]; } }

[[maybe_unused]] static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

} // namespace dw
