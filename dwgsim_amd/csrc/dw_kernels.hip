// dw_kernels.hip -- hand-written HIP kernels (gfx950 / MI355X) for the dwgsim hot path.
//
//   mutation walk (replaces src/mut.c:591-643 + :481-589 of the reference)
//     k_pack          ASCII -> base codes, initialises both haplotypes            HBM: 1 B in, 3 B out / base
//     k_site_scan     one Philox draw per position: candidate sites (bitmask)     HBM: 1 B in / base; ALU (Philox)
//     k_scan_excl     single-block exclusive scan of per-block counts
//     k_compact       ordered compaction of a bitmask into a position list
//     k_events        one thread per candidate: speculative event (type, ploidy, lengths)
//     k_resolve       liveness of candidates (deletion runs swallow later candidates)
//     k_apply         writes live events into the haplotype cells + insertion tables
//     k_jreach / k_sufmin / k_jbound / k_jrun   left-justification of indels: exact sequential
//                     semantics inside independent clusters, clusters in parallel
//     k_collect_mask / k_gather   list of mutated cells for the host's txt/vcf writer
//   read simulation (replaces the loop body src/dwgsim.c:636-1099)
//     k_place         per pair: attempts until the N filter / geometry accept; random-read flag
//     k_simulate      per read end: base extraction through indels, errors, qualities,
//                     FASTQ formatting, decoupled look-back for record offsets, packed stores
//
// Everything is byte/integer work plus fp64 for the normals: no MFMA.  fp64 expressions mirror the
// reference's evaluation order; compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include "dw_common.hpp"
#include "dw_kernels.hpp"

// The file is compiled in parts so the twelve k_simulate variants build in parallel (csrc/Makefile):
//   DW_PART 0: mutation-walk kernels, k_place, host launchers and the k_simulate dispatcher
//   DW_PART 1..6: k_simulate<LPP, *, DT> for (LPP, DT) = (2,0) (1,0) (2,2) (1,2) (2,1) (1,1)
//   DW_PART -1 (default): everything in one translation unit
#ifndef DW_PART
#define DW_PART -1
#endif
#define DW_HAS(part) (DW_PART == -1 || DW_PART == (part))
#ifndef DW_SIM_WAVES
#define DW_SIM_WAVES 5       // minimum waves per SIMD requested for the Illumina variants (one less when both output families are written):
                             // the kernel sits 1-2 VGPRs above these occupancy steps without the hint; measured +4 % at 5 vs 4 waves, 6 spills (so do the SOLiD variants with any hint)
#endif
#ifndef DW_SIM_WAVES_BOTH
#define DW_SIM_WAVES_BOTH 4  // ... when both output families are written (-o 0)
#endif
#ifndef DW_ION_WAVES
#define DW_ION_WAVES 1       // minimum waves per SIMD requested for the (latency-bound) Ion Torrent variants
#endif

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

DW_DEV uint32_t ins_find(const HapDev &h, int64_t pos)
{
    uint32_t lo = 0, hi = h.n_ins;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)h.ins_pos[mid] < pos) lo = mid + 1; else hi = mid; }
    return lo;
}

#if DW_HAS(0)
// ------------------------------------------------------------------------------------------------
// K0: ASCII -> codes; both haplotypes start as the reference (mut.c:609)
// ------------------------------------------------------------------------------------------------
__global__ void k_pack(const uint8_t *__restrict__ ascii, uint8_t *__restrict__ ref, uint8_t *__restrict__ h0,
                       uint8_t *__restrict__ h1, int64_t l)
{
    const int64_t nchunk = (l + 15) >> 4;
    for (int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunk; ch += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p0 = ch << 4;
        uint32_t w[4] = {0, 0, 0, 0};
        if (p0 + 16 <= l) {
            const uint4 v = *reinterpret_cast<const uint4 *>(ascii + p0);
            const uint32_t in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int b = 0; b < 4; ++b) w[q] |= code_of_ascii((in[q] >> (8 * b)) & 0xff) << (8 * b);
        } else {
            for (int b = 0; b < 16; ++b) { const uint32_t c = (p0 + b < l) ? code_of_ascii(ascii[p0 + b]) : 4u; w[b >> 2] |= c << (8 * (b & 3)); }
        }
        const uint4 o = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4 *>(ref + p0) = o;
        *reinterpret_cast<uint4 *>(h0 + p0) = o;
        *reinterpret_cast<uint4 *>(h1 + p0) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// K1: candidate sites.  mut.c:618 `c < 4 && drand48() < opt->mut_rate` with the draw taken from
// (D_WALK, position, slot 1).  16 positions per thread, 4096 per block; emits a bitmask (uint16 per
// thread) and the per-block candidate count.
// ------------------------------------------------------------------------------------------------
__global__ void k_site_scan(const uint8_t *__restrict__ ref, int64_t l, WalkParams wp, uint32_t contig_index,
                            uint16_t *__restrict__ mask, uint32_t *__restrict__ block_count)
{
    __shared__ uint32_t sm[17];
    const RngKey key{wp.seed, contig_index};
    const int64_t p0 = ((int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_POS_PER_THREAD;
    uint32_t bits = 0;
    if (p0 < l) {
        const uint4 v = *reinterpret_cast<const uint4 *>(ref + p0);      // ref is padded: reading past l is safe
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint32_t c = (in[b >> 2] >> (8 * (b & 3))) & 0xff;
            if (c < 4 && p0 + b < l) {
                const U4 blk = rng_block(key, D_WALK, (uint64_t)(p0 + b), 0, 0, 0);
                if ((((uint64_t)blk.z << 21) | (uint64_t)(blk.w >> 11)) < wp.mut_thr53) bits |= 1u << b;   // u53(w2,w3) < mut_rate
            }
        }
    }
    mask[(int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x] = (uint16_t)bits;
    uint32_t total;
    (void)block_excl_scan((uint32_t)__popc(bits), sm, &total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

// single-block exclusive scan (in place) of n uint32; optional 64-bit total
__global__ void k_scan_excl(uint32_t *data, uint32_t n, uint64_t *total_out)
{
    __shared__ uint32_t sm[17];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    uint64_t grand = 0;
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? data[i] : 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, sm, &total);
        const uint32_t carry = carry_s;
        if (i < n) data[i] = ex + carry;
        grand += total;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (total_out && threadIdx.x == 0) *total_out = grand;
}

// ordered compaction: positions of set bits of `mask` (one uint16 per thread of the producing kernel)
__global__ void k_compact(const uint16_t *__restrict__ mask, const uint32_t *__restrict__ block_base, int32_t *__restrict__ out)
{
    __shared__ uint32_t sm[17];
    uint32_t bits = mask[(int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x];
    uint32_t total;
    uint32_t off = block_base[blockIdx.x] + block_excl_scan((uint32_t)__popc(bits), sm, &total);
    const int64_t p0 = ((int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_POS_PER_THREAD;
    while (bits) { const int b = __ffs((int)bits) - 1; bits &= bits - 1; out[off++] = (int32_t)(p0 + b); }
}

// ------------------------------------------------------------------------------------------------
// K2a: one thread per candidate site: the event it would be if it is live.  mut.c:619-640, 287-308.
// ------------------------------------------------------------------------------------------------
__global__ void k_events(const int32_t *__restrict__ cand, uint32_t n_cand, const uint8_t *__restrict__ ref, int64_t l,
                         WalkParams wp, uint32_t contig_index, Event *__restrict__ ev, uint32_t *__restrict__ max_del)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_cand) {
        const RngKey key{wp.seed, contig_index};
        const int64_t p = cand[k];
        const uint32_t c = ref[p];
        const U4 b1 = rng_block(key, D_WALK, (uint64_t)p, 0, 0, 1);   // slots 2,3
        const U4 b2 = rng_block(key, D_WALK, (uint64_t)p, 0, 0, 2);   // slots 4,5
        Event e; e.pos = (int32_t)p; e.live = 1; e.len = 1; e.base = (uint8_t)c;
        if (u_lo(b1) >= wp.indel_frac) {                 // substitution (mut.c:619-626)
            e.type = 1;
            e.base = (uint8_t)((c + (uint32_t)(uint64_t)(u_hi(b1) * 3.0 + 1)) & 3);
            e.hap = (wp.is_hap || u_lo(b2) < 0.333333) ? 3 : (u_hi(b2) < 0.5 ? 1 : 2);
        } else if (u_hi(b1) < 0.5) {                     // deletion (mut.c:628-636) + its run (mut.c:610-617)
            e.type = 2;
            e.hap = (wp.is_hap || u_lo(b2) < 0.3333333) ? 3 : (u_hi(b2) < 0.5 ? 1 : 2);
            uint32_t len = 1;
            for (int64_t q = p + 1; q < l; ++q) {
                if ((int64_t)len < wp.indel_min || rng_slot(key, D_WALK, (uint64_t)q, 0, 0) < wp.indel_extend) ++len; else break;
            }
            e.len = len;
            atomicMax(max_del, len);
        } else {                                         // insertion (mut.c:637-639 -> :287-308)
            e.type = 3;
            uint64_t num = 0; uint32_t kk = 0;
            do { ++num; } while (num < 0xFFFFFFFFull && ((int64_t)num < wp.indel_min || rng_slot(key, D_WALK_INSLEN, (uint64_t)p, 0, kk++) < wp.indel_extend));
            e.len = (uint32_t)num;
            e.hap = (wp.is_hap || u_lo(b2) < 0.333333) ? 3 : (u_hi(b2) < 0.5 ? 1 : 2);
        }
        ev[k] = e;
    }
}

// K2b: liveness.  In the sequential walk a candidate inside an active deletion run is never tested
// (mut.c:610-615 `continue`).  Candidate k is dead iff a LIVE earlier deletion covers it.  Exact
// parallel evaluation: a candidate no earlier deletion reaches at all is live; otherwise replay the
// (tiny) chain from the nearest such anchor.
DW_DEV bool reached_by_any(const Event *ev, uint32_t k, uint32_t max_del)
{
    const int64_t pk = ev[k].pos;
    for (int64_t j = (int64_t)k - 1; j >= 0; --j) {
        const int64_t pj = ev[j].pos;
        if (pk - pj >= (int64_t)max_del) break;
        if (ev[j].type == 2 && pj + (int64_t)ev[j].len - 1 >= pk) return true;
    }
    return false;
}
__global__ void k_resolve(Event *ev, uint32_t n_cand, const uint32_t *max_del_p, uint4 *flags)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    const uint32_t max_del = *max_del_p;
    bool live = true;
    if (max_del > 1 && reached_by_any(ev, k, max_del)) {
        int64_t a = (int64_t)k - 1;
        while (a > 0 && reached_by_any(ev, (uint32_t)a, max_del)) --a;     // anchor: definitely live (or first candidate)
        int64_t reach = -1;
        for (int64_t m = a; m <= (int64_t)k; ++m) {
            const bool lv = ev[m].pos > reach;
            if (lv && ev[m].type == 2) reach = (int64_t)ev[m].pos + ev[m].len - 1;
            live = lv;
        }
    }
    // (the live flag is written to a side array so that neighbours still read the speculative events)
    const Event e = ev[k];
    uint4 f;
    f.x = (live && e.type == 3 && (e.hap & 1)) ? 1u : 0u;       // insertion count hap 1
    f.y = (live && e.type == 3 && (e.hap & 1)) ? e.len : 0u;    // inserted bases hap 1
    f.z = (live && e.type == 3 && (e.hap & 2)) ? 1u : 0u;
    f.w = (live && e.type == 3 && (e.hap & 2)) ? e.len : 0u;
    if (!live) f.x |= 0x80000000u;                               // dead marker
    flags[k] = f;
}

// exclusive scan of the four insertion-allocation columns (single block); totals -> tot[0..3]
__global__ void k_scan4(uint4 *flags, uint32_t n, uint32_t *tot)
{
    __shared__ uint32_t sm[17];
    __shared__ uint32_t carry_s[4];
    if (threadIdx.x < 4) carry_s[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i < n) v = flags[i];
        const uint32_t dead = v.x & 0x80000000u;
        uint32_t in[4] = {v.x & 0x7fffffffu, v.y, v.z, v.w}, ex[4], total[4];
        for (int c = 0; c < 4; ++c) ex[c] = block_excl_scan(in[c], sm, &total[c]) + carry_s[c];
        if (i < n) flags[i] = make_uint4(ex[0] | dead, ex[1], ex[2], ex[3]);
        __syncthreads();
        if (threadIdx.x < 4) carry_s[threadIdx.x] += total[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x < 4) tot[threadIdx.x] = carry_s[threadIdx.x];
}

// K3: write live events into the cells and the insertion tables.
__global__ void k_apply(Event *ev, uint32_t n_cand, const uint4 *flags, ContigDev c, WalkParams wp)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    const uint4 f = flags[k];
    Event e = ev[k];
    if (f.x & 0x80000000u) { ev[k].live = 0; return; }
    const int64_t p = e.pos;
    if (e.type == 1) {
        if (e.hap & 1) c.hap[0].cells[p] = T_SUB | e.base;
        if (e.hap & 2) c.hap[1].cells[p] = T_SUB | e.base;
    } else if (e.type == 2) {
        for (uint32_t q = 0; q < e.len; ++q) {
            const uint8_t v = T_DEL | c.ref[p + q];
            if (e.hap & 1) c.hap[0].cells[p + q] = v;
            if (e.hap & 2) c.hap[1].cells[p + q] = v;
        }
    } else {
        const RngKey key{wp.seed, c.contig_index};
        const uint32_t idx[2] = {f.x & 0x7fffffffu, f.z}, off[2] = {f.y, f.w};
        for (int h = 0; h < 2; ++h) if (e.hap & (1 << h)) {
            c.hap[h].cells[p] = T_INS | e.base;
            c.hap[h].ins_pos[idx[h]] = (int32_t)p;
            c.hap[h].ins_len[idx[h]] = e.len;
            c.hap[h].ins_off[idx[h]] = off[h];
        }
        for (uint32_t j = 0; j < e.len; ++j) {      // draw j lands at printed index len-1-j (mut.c:313-315, :347-365 read by :249-279)
            const uint8_t b = (uint8_t)(uint64_t)(rng_slot(key, D_WALK_INSBASE, (uint64_t)p, 0, j) * 4.0);
            if (e.hap & 1) c.hap[0].ins_bases[off[0] + e.len - 1 - j] = b;
            if (e.hap & 2) c.hap[1].ins_bases[off[1] + e.len - 1 - j] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4: left-justification (mut.c:427-589).  The reference scans every position; only mutated cells act
// and unmutated (non-N) positions merely reset prev_del, so the walk visits the live events'
// original footprints in order and treats the gaps between them in O(1).
// ------------------------------------------------------------------------------------------------
DW_DEV void justify_ins(HapDev &h, int64_t i)          // mut.c:427-478
{
    const uint32_t idx = ins_find(h, i);
    const uint32_t n = h.ins_len[idx];
    uint8_t *P = h.ins_bases + h.ins_off[idx];
    int64_t j = i;
    while (j > 0 && (h.cells[j - 1] & TMASK) == T_NONE && P[n - 1] == (h.cells[j - 1] & 3)) {
        for (uint32_t t = n - 1; t > 0; --t) P[t] = P[t - 1];
        P[0] = h.cells[j - 1] & 3;
        h.cells[j] = h.cells[j] & 3;
        --j;
    }
    h.cells[j] = T_INS | (h.cells[j] & 3);
    h.ins_pos[idx] = (int32_t)j;
}
DW_DEV void del_swap(HapDev &h, int64_t j, int64_t dl)   // mut.c:515-516
{
    const uint8_t t = h.cells[j]; h.cells[j] = h.cells[j + dl]; h.cells[j + dl] = (uint8_t)((t | TMASK) ^ TMASK);
}
DW_DEV int64_t del_run(const HapDev &h, int64_t i, int64_t l)
{
    int64_t dl = 1;
    for (int64_t j = i + 1; j < l && (h.cells[j] & TMASK) == T_DEL; ++j) ++dl;
    return dl;
}
DW_DEV void justify_visit(ContigDev &c, int64_t i, int *prev_del)
{
    if (c.ref[i] >= 4) return;
    HapDev &h0 = c.hap[0], &h1 = c.hap[1];
    const uint8_t c1 = h0.cells[i], c2 = h1.cells[i];
    if ((c1 & TMASK) == T_NONE && (c2 & TMASK) == T_NONE) { prev_del[0] = prev_del[1] = 0; return; }
    if ((c1 & BTMASK) == (c2 & BTMASK)) {
        if ((c1 & TMASK) == T_SUB) { prev_del[0] = prev_del[1] = 0; }
        else if ((c1 & TMASK) == T_DEL) {
            if (prev_del[0] == 1 || prev_del[1] == 1) return;
            prev_del[0] = prev_del[1] = 1;
            const int64_t dl = del_run(h0, i, c.l);
            if (c.l <= i + dl) return;
            if (i > 0) for (int64_t j = i - 1;; --j) {
                const uint8_t a = h0.cells[j], b = h1.cells[j];
                if ((a & TMASK) != T_INS && (b & TMASK) != T_INS && (a & TMASK) != T_DEL && (b & TMASK) != T_DEL
                    && (a & 3) == (h0.cells[j + dl] & 3) && (b & 3) == (h1.cells[j + dl] & 3)) { del_swap(h0, j, dl); del_swap(h1, j, dl); }
                else break;
                if (j == 0) break;
            }
        } else { prev_del[0] = prev_del[1] = 0; justify_ins(h0, i); justify_ins(h1, i); }
    } else {
        if ((c1 & TMASK) == T_SUB || (c2 & TMASK) == T_SUB) { prev_del[0] = prev_del[1] = 0; }
        else if ((c1 & TMASK) == T_DEL || (c2 & TMASK) == T_DEL) {
            const int x = ((c1 & TMASK) == T_DEL) ? 0 : 1;
            if (prev_del[x] == 1) return;
            prev_del[x] = 1;
            HapDev &h = c.hap[x];
            const int64_t dl = del_run(h, i, c.l);
            if (c.l <= i + dl) return;
            if (i > 0) for (int64_t j = i - 1;; --j) {
                const uint8_t a = h.cells[j];
                if ((a & TMASK) == T_NONE && (a & 3) == (h.cells[j + dl] & 3)) del_swap(h, j, dl); else break;
                if (j == 0) break;
            }
        } else if ((c1 & TMASK) == T_INS) { prev_del[0] = prev_del[1] = 0; justify_ins(h0, i); }
        else { prev_del[0] = prev_del[1] = 0; justify_ins(h1, i); }
    }
}
// sequential cross-check (DWGSIM_HIP_JUSTIFY=seq): one thread walks every live event of the contig
__global__ void k_justify_seq(const Event *ev, uint32_t n_cand, ContigDev c)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int prev_del[2] = {0, 0};
    int64_t last = -1;
    for (uint32_t k = 0; k < n_cand; ++k) {
        const Event e = ev[k];
        if (!e.live) continue;
        const int64_t p = e.pos, right = p + (e.type == 2 ? (int64_t)e.len - 1 : 0);
        if (prev_del[0] | prev_del[1])
            for (int64_t q = last + 1; q < p; ++q) if (c.ref[q] < 4) { prev_del[0] = prev_del[1] = 0; break; }
        for (int64_t i = p; i <= right; ++i) justify_visit(c, i, prev_del);
        last = right;
    }
}

// ---- parallel left-justification --------------------------------------------------------------
// The reference's pass is sequential, but an indel only interacts with what its leftward scan can
// touch.  (1) k_jreach: per live event, a conservative lower bound `lo` of every cell its scan can
// read or write: continue while the cell is mutated on either haplotype (pre-justify state) or the
// reference is periodic there (deletion: ref[j]&3 == ref[j+L]&3, insertion: the rotated copy keeps
// matching).  Shifts preserve (cell & 3) at every position, so the true scan never goes further.
// (2) k_sufmin: suffix minimum of lo.  (3) k_jbound: event b starts a new cluster iff no event >= b
// can reach the previous live event's footprint and an unmutated non-N position separates them
// (prev_del is then 0, mut.c:585-587).  (4) k_jrun: one thread per cluster replays the exact
// sequential semantics (justify_visit) over its events; clusters touch disjoint cells.
DW_DEV int64_t reach_del(const ContigDev &c, int h, int64_t p)
{
    // period = the run of DELETE cells the sequential pass would measure at p (adjacent runs merge,
    // mut.c:503 / :535 / :557) on haplotype h
    const int64_t L = del_run(c.hap[h], p, c.l);
    int64_t j = p - 1;
    for (; j >= 0; --j) {
        const bool mutated = ((c.hap[0].cells[j] | c.hap[1].cells[j]) & TMASK) != 0;
        if (!(mutated || (p + L < c.l && (c.ref[j] & 3) == (c.ref[j + L] & 3)))) break;   // run at the contig end never moves (mut.c:506)
    }
    return j < 0 ? 0 : j;                              // last cell read
}
DW_DEV int64_t reach_ins(const ContigDev &c, int h, int64_t p)
{
    const uint32_t idx = ins_find(c.hap[h], p);
    const uint32_t n = c.hap[h].ins_len[idx];
    const uint8_t *P = c.hap[h].ins_bases + c.hap[h].ins_off[idx];
    // rotating left by one makes the cell's base the new first base: after r rotations the last
    // inserted base is P[n-1-r] while r < n, then ref[p-1-(r-n)] & 3 (bases rotated in earlier)
    int64_t j = p - 1; int64_t r = 0;
    for (; j >= 0; --j, ++r) {
        const bool mutated = ((c.hap[0].cells[j] | c.hap[1].cells[j]) & TMASK) != 0;
        const uint32_t last = r < (int64_t)n ? (uint32_t)P[n - 1 - r] : (uint32_t)(c.ref[p - 1 - (r - n)] & 3);
        if (!(mutated || last == (uint32_t)(c.ref[j] & 3))) break;
    }
    return j < 0 ? 0 : j;
}
__global__ void k_jreach(const Event *__restrict__ ev, uint32_t n_cand, ContigDev c, int32_t *__restrict__ lo)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    const Event e = ev[k];
    int64_t reach = 0x7fffffff;
    if (e.live) {
        const int64_t p = e.pos;
        reach = p;                                      // substitution: no scan; it only matters as a neighbour (prev_del)
        if (e.type == 2) reach = reach_del(c, (e.hap & 1) ? 0 : 1, p);         // haplotype 1 for hom and hap-1 events
        else if (e.type == 3) reach = reach_ins(c, (e.hap & 1) ? 0 : 1, p);    // both copies carry the same bases before justification
        else if (e.type == 4) {                         // cell patched from a mutation-input file: whatever the two cells hold
            for (int h = 0; h < 2; ++h) {
                const uint8_t t = c.hap[h].cells[p] & TMASK;
                int64_t r = p;
                if (t == T_DEL) r = reach_del(c, h, p); else if (t == T_INS) r = reach_ins(c, h, p);
                if (r < reach) reach = r;
            }
        }
    }
    lo[k] = (int32_t)reach;
}
// single block: sufmin[k] = min(lo[k..n))
__global__ void k_sufmin(const int32_t *__restrict__ lo, uint32_t n, int32_t *__restrict__ sufmin)
{
    __shared__ int32_t sm[16];
    __shared__ int32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0x7fffffff;
    __syncthreads();
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    const uint32_t nchunk = (n + blockDim.x - 1) / blockDim.x;
    for (uint32_t ch = 0; ch < nchunk; ++ch) {
        // walk the array from its end: thread t handles element (n-1) - (ch*blockDim + t)
        const int64_t i = (int64_t)n - 1 - ((int64_t)ch * blockDim.x + threadIdx.x);
        int32_t v = i >= 0 ? lo[i] : 0x7fffffff;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(v, d); if (lane >= d && o < v) v = o; }   // inclusive min-scan
        if (lane == 63) sm[wave] = v;
        __syncthreads();
        int32_t pre = carry_s;
        for (int w = 0; w < wave; ++w) if (sm[w] < pre) pre = sm[w];
        if (pre < v) v = pre;
        if (i >= 0) sufmin[i] = v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry_s = v;
        (void)nw;
        __syncthreads();
    }
}
__global__ void k_jbound(const Event *__restrict__ ev, uint32_t n_cand, ContigDev c, const int32_t *__restrict__ sufmin, uint8_t *__restrict__ bound)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    uint8_t b = 0;
    if (ev[k].live) {
        int64_t a = (int64_t)k - 1;
        while (a >= 0 && !ev[a].live) --a;
        if (a < 0) b = 1;
        else {
            const int64_t right_a = (int64_t)ev[a].pos + (ev[a].type == 2 ? (int64_t)ev[a].len - 1 : 0);
            if ((int64_t)sufmin[k] > right_a) {
                for (int64_t q = right_a + 1; q < (int64_t)ev[k].pos; ++q) if (c.ref[q] < 4) { b = 1; break; }
            }
        }
    }
    bound[k] = b;
}
__global__ void k_jrun(const Event *__restrict__ ev, uint32_t n_cand, ContigDev c, const uint8_t *__restrict__ bound)
{
    const uint32_t k0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k0 >= n_cand || !bound[k0]) return;
    int prev_del[2] = {0, 0};
    int64_t last = -1;
    for (uint32_t k = k0; k < n_cand; ++k) {
        const Event e = ev[k];
        if (!e.live) continue;
        if (k > k0 && bound[k]) break;
        const int64_t p = e.pos, right = p + (e.type == 2 ? (int64_t)e.len - 1 : 0);
        if (k > k0 && (prev_del[0] | prev_del[1]))      // an unmutated non-N position in the gap resets prev_del (mut.c:585-587)
            for (int64_t q = last + 1; q < p; ++q) if (c.ref[q] < 4) { prev_del[0] = prev_del[1] = 0; break; }
        for (int64_t i = p; i <= right; ++i) justify_visit(c, i, prev_del);
        last = right;
    }
}

// mutation-input files: the host resolved the file's entries into final cell values (dw_mutin.cpp); scatter them
__global__ void k_apply_patches(const int32_t *__restrict__ pos, const uint16_t *__restrict__ cells, uint32_t n, uint8_t *h0, uint8_t *h1)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { h0[pos[k]] = (uint8_t)(cells[k] & 0xff); h1[pos[k]] = (uint8_t)(cells[k] >> 8); }
}

// mutated cells for the host's mutations.txt / .vcf writer
__global__ void k_collect_mask(const uint8_t *__restrict__ h0, const uint8_t *__restrict__ h1, int64_t l,
                               uint16_t *__restrict__ mask, uint32_t *__restrict__ block_count)
{
    __shared__ uint32_t sm[17];
    const int64_t p0 = ((int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_POS_PER_THREAD;
    uint32_t bits = 0;
    if (p0 < l) {
        const uint4 a = *reinterpret_cast<const uint4 *>(h0 + p0), b = *reinterpret_cast<const uint4 *>(h1 + p0);
        const uint32_t m[4] = {(a.x | b.x) & 0x30303030u, (a.y | b.y) & 0x30303030u, (a.z | b.z) & 0x30303030u, (a.w | b.w) & 0x30303030u};
#pragma unroll
        for (int q = 0; q < 16; ++q) if (((m[q >> 2] >> (8 * (q & 3))) & 0xff) && p0 + q < l) bits |= 1u << q;
    }
    mask[(int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x] = (uint16_t)bits;
    uint32_t total;
    (void)block_excl_scan((uint32_t)__popc(bits), sm, &total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}
__global__ void k_gather(const int32_t *__restrict__ pos, uint32_t n, const uint8_t *__restrict__ h0, const uint8_t *__restrict__ h1, uint16_t *__restrict__ cells)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) cells[k] = (uint16_t)(h0[pos[k]] | (h1[pos[k]] << 8));
}

#endif // DW_HAS(0): mutation walk

// ------------------------------------------------------------------------------------------------
// Read simulation
// ------------------------------------------------------------------------------------------------
struct ReadRes { int32_t ext_coor, n_sub, n_indel, num_n, n_ins; };   // n_ins: INSERT cells crossed (the reference's n_indel_first, dwgsim.c:98)

// dwgsim.c:75-153 __gen_read.  STORE: packed 4-bit bases go to lds[word * stride].
// The haplotype is read in 16-byte chunks with the next chunk prefetched; runs of up to 8 cells that
// hold no INSERT/DELETE cell (bit 4 clear) are handled at once with byte-parallel arithmetic, any other
// cell goes through the reference's per-cell logic.
template <bool STORE>
DW_DEV ReadRes gen_read(const HapDev &h, int64_t l, int64_t start, int step, int s, int strand, uint32_t *lds, int stride)
{
    ReadRes r{-10, 0, 0, 0, 0};
    int k = 0, kw = 0; uint64_t acc = 0; uint32_t nacc = 0;       // nacc nibbles pending in acc
    auto push = [&](uint64_t nibs, uint32_t cnt) {                 // append cnt packed nibbles
        if (STORE) {
            acc |= nibs << (4 * nacc); nacc += cnt;
            if (nacc >= 8) { lds[kw * stride] = (uint32_t)acc; acc >>= 32; nacc -= 8; ++kw; }
        }
        k += (int)cnt;
    };
    auto emit = [&](uint32_t v) {
        if (strand) v = v < 4 ? 3 - v : 4;                 // dwgsim.c:150-152
        r.num_n += (v == 4);                                // dwgsim.c:824-831
        push(v, 1);
    };
    const int64_t last_chunk = (l - 1) >> 4;
    const int dirc = step > 0 ? 1 : -1;
    int64_t cb = -1, pb = -1; uint64_t clo = 0, chi = 0, plo = 0, phi = 0;
    if (start >= 0 && start < l) {
        cb = start >> 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(h.cells + (cb << 4));
        clo = (uint64_t)v.x | ((uint64_t)v.y << 32); chi = (uint64_t)v.z | ((uint64_t)v.w << 32);
        pb = cb + dirc;
        if (pb >= 0 && pb <= last_chunk) { const uint4 w = *reinterpret_cast<const uint4 *>(h.cells + (pb << 4)); plo = (uint64_t)w.x | ((uint64_t)w.y << 32); phi = (uint64_t)w.z | ((uint64_t)w.w << 32); }
#if !defined(DW_EMU) && !defined(DW_NO_TOUCH)
        // touch the following 128-byte lines of the read's window now (results unused): their HBM latency overlaps with the first
        // chunks instead of being met one line at a time by the chunk loop
        for (int t = 1; t <= 3 && t * 128 < s; ++t) {                    // only addresses the read is sure to reach: no over-fetch
            const int64_t pa = start + (int64_t)dirc * 128 * t;
            if (pa >= 0 && pa < l) (void)*reinterpret_cast<const volatile uint32_t *>(h.cells + (pa & ~(int64_t)3));
        }
#endif
    }
    int64_t i = start;
    while (i >= 0 && i < l && k < s) {
        if ((i >> 4) != cb) {
            cb = i >> 4;
            if (cb == pb) { clo = plo; chi = phi; }
            else { const uint4 v = *reinterpret_cast<const uint4 *>(h.cells + (cb << 4)); clo = (uint64_t)v.x | ((uint64_t)v.y << 32); chi = (uint64_t)v.z | ((uint64_t)v.w << 32); }
            pb = cb + dirc;
            if (pb >= 0 && pb <= last_chunk) { const uint4 w = *reinterpret_cast<const uint4 *>(h.cells + (pb << 4)); plo = (uint64_t)w.x | ((uint64_t)w.y << 32); phi = (uint64_t)w.z | ((uint64_t)w.w << 32); }
        }
        const uint64_t half = (i & 8) ? chi : clo;
        const uint32_t off = (uint32_t)(i & 7);
        // cells of this 8-byte half in travel order, limited by the half, the read and the contig end
        uint32_t want = step > 0 ? 8 - off : off + 1;
        const uint32_t left = (uint32_t)(s - k);
        if (want > left) want = left;
        if (step > 0) { const int64_t room = l - i; if ((int64_t)want > room) want = (uint32_t)room; }
        uint64_t cells = step > 0 ? half >> (8 * off) : __builtin_bswap64(half << (8 * (7 - off)));
        if (want < 8) cells &= (1ull << (8 * want)) - 1;
        if ((cells & 0x1010101010101010ull) == 0) {           // only NOCHANGE / SUBSTITUTE cells: one base each
            if (r.ext_coor < 0) { r.ext_coor = (int32_t)i; if (strand) r.ext_coor -= s - 1; }
            r.n_sub += __popcll(cells & 0x2020202020202020ull);
            uint64_t codes = cells & 0x0f0f0f0f0f0f0f0full;
            const uint64_t ge4 = (codes >> 2) & 0x0101010101010101ull;
            if (strand) {
                codes = ((codes ^ 0x0303030303030303ull) & ~(ge4 * 0x0f)) | (ge4 << 2);
                if (want < 8) codes &= (1ull << (8 * want)) - 1;
                r.num_n += __popcll(ge4);
            } else r.num_n += __popcll(ge4 & ~codes);          // exactly code 4 (code 5, '-', is not counted on this strand)
            uint64_t x = codes;
            x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
            x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
            x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
            push(x, want);
            i += (int64_t)step * want;
            continue;
        }
        const uint32_t c = (uint32_t)(half >> (8 * off)) & 0xffu, mt = c & TMASK;
        if (r.ext_coor < 0) {
            if (mt != T_NONE && mt != T_SUB) { i += step; continue; }
            r.ext_coor = (int32_t)i;
            if (strand) r.ext_coor -= s - 1;
        }
        if (mt == T_DEL) { ++r.n_indel; if (strand) r.ext_coor--; }
        else if (mt == T_NONE || mt == T_SUB) { emit(c & 0xf); if (mt == T_SUB) ++r.n_sub; }
        else {
            ++r.n_indel; ++r.n_ins;
            const uint32_t idx = ins_find(h, i);
            uint32_t n = h.ins_len[idx];
            const uint8_t *P = h.ins_bases + h.ins_off[idx];
            if (!strand) {
                if (k < s) emit(c & 0xf);
                for (uint32_t t = 0; t < n && k < s; ++t) emit(P[t] & 3u);
            } else {
                while (n > 0 && k < s) { r.ext_coor++; emit(P[n - 1] & 3u); --n; }
                if (k < s) emit(c & 0xf);
            }
        }
        i += step;
    }
    if (STORE && nacc) lds[kw * stride] = (uint32_t)acc;
    if (k != s) r.ext_coor = -10;
    return r;
}

struct PairDraw { bool is_rand; int32_t pos, d; int hap, strand0, strand1; };

// select-by-value accessors: dynamic indexing into the by-value kernel argument block would force a
// private copy of the whole struct (promoted to LDS by the backend)
DW_DEV HapDev sel_hap(const SimArgs &a, int h)
{
    HapDev r;
    r.cells = h ? a.c.hap[1].cells : a.c.hap[0].cells;
    r.ins_pos = h ? a.c.hap[1].ins_pos : a.c.hap[0].ins_pos;
    r.ins_len = h ? a.c.hap[1].ins_len : a.c.hap[0].ins_len;
    r.ins_off = h ? a.c.hap[1].ins_off : a.c.hap[0].ins_off;
    r.ins_bases = h ? a.c.hap[1].ins_bases : a.c.hap[0].ins_bases;
    r.n_ins = h ? a.c.hap[1].n_ins : a.c.hap[0].n_ins;
    return r;
}
DW_DEV int sel_len(const SimArgs &a, int j) { return j ? a.p.len[1] : a.p.len[0]; }

// dwgsim.c:649-742: random-read test, fragment size + position, haplotype, strands
DW_DEV PairDraw draw_pair(const SimArgs &a, RngKey key, uint64_t ii, uint32_t att)
{
    PairDraw pd; pd.pos = 0; pd.d = 0; pd.hap = 0; pd.strand0 = pd.strand1 = 0;
    const U4 b0 = rng_block(key, D_PAIR, ii, att, 0, 0);
    pd.is_rand = !(a.p.rand_read < u_lo(b0));
    if (pd.is_rand) return pd;
    const int s0 = a.p.len[0], s1 = a.p.len[1];
    const int64_t l = a.l_place, sl = a.c.l;          // placement length (region length with -x) vs contig length
    if (a.p.amplicons) { pd.pos = 0; pd.d = (int32_t)sl; }
    else {
        uint32_t t = 0; int32_t pos, d; bool continue_flag = false;
        do {
            if (s1 > 0) {
                double v1, v2, rsq; uint32_t r = 0;
                do {
                    const U4 b = rng_block(key, D_PLACE_NORM, ii, att, r, t);
                    v1 = 2.0 * u_lo(b) - 1.0; v2 = 2.0 * u_hi(b) - 1.0;
                    rsq = v1 * v1 + v2 * v2; ++r;
                } while (rsq >= 1.0 || rsq == 0.0);
                const double fac = sqrt(-2.0 * det_log(rsq) / rsq);
                double ran = v2 * fac;
                ran = ran * a.p.std_dev + a.p.dist;
                d = (int32_t)(ran + 0.5);
                const int32_t min_dist = s0 + s1;
                if (d < min_dist) d = min_dist;
                if ((int64_t)d > l) d = (int32_t)l;
            } else d = 0;
            const int64_t range = l - d + 1;
            pos = (int32_t)((double)range * rng_slot(key, D_PLACE, ii, att, t));
            bool inside = true;
            if (a.have_regions) {                         // dwgsim.c:696-707: region coordinate -> contig coordinate, then regions_bed_query (:712)
                for (int q = 0; q < a.n_reg; ++q) {
                    const int32_t jl = a.reg_end[q] - a.reg_start[q];
                    if (pos < jl) { pos = a.reg_start[q] + pos - 1; break; }
                    pos -= jl;
                }
                inside = false;                           // regions are sorted and disjoint: "some region contains [pos, pos + d)" (regions_bed.c:130-156)
                int lo = 0, hi = a.n_reg - 1;
                const uint32_t qs = (uint32_t)pos, qe = (uint32_t)(pos + d);
                while (lo <= hi) {
                    const int mid = lo + (hi - lo) / 2;
                    if (qs < (uint32_t)a.reg_start[mid]) hi = mid - 1;
                    else if ((uint32_t)a.reg_end[mid] < qe) lo = mid + 1;
                    else { inside = true; break; }
                }
            }
            ++t;
            if (t > (1u << 20)) { pd.hap = -1; break; }   // the reference would never terminate here; reported as an error by the caller
            continue_flag = !inside;
        } while (continue_flag || pos < 0 || pos >= sl || (int64_t)pos + d - 1 >= sl
                 || (s1 > 0 && !a.p.is_inner && ((s0 > 0 && d <= s1) || (d <= s0 && s1 > 0))));
        pd.pos = pos; pd.d = d;
    }
    if (pd.hap < 0) { pd.hap = 0; pd.pos = 0; pd.d = (int32_t)(s0 + s1 < sl ? s0 + s1 : sl); atomicOr((unsigned long long *)&a.counters[2], 4ull); return pd; }
    pd.hap = u_hi(b0) < a.p.mut_freq ? 0 : 1;
    switch (a.p.read_one_strand) {
    case 0: pd.strand0 = rng_slot(key, D_PAIR, ii, att, 2) < 0.5 ? 1 : 0; break;
    case 1: pd.strand0 = 0; break;
    default: pd.strand0 = 1; break;
    }
    switch (a.p.strandedness) {
    case 0: pd.strand1 = (a.p.data_type == 0) ? 1 - pd.strand0 : pd.strand0; break;
    case 1: pd.strand1 = pd.strand0; break;
    default: pd.strand1 = 1 - pd.strand0; break;
    }
    return pd;
}

// dwgsim.c:745-821: first cell and direction of read end j
DW_DEV void read_geom(const SimArgs &a, const PairDraw &pd, int j, int64_t *start, int *step)
{
    const int64_t pos = pd.pos, d = pd.d, s0 = a.p.len[0], s1 = a.p.len[1], sl = a.c.l;
    const bool amp = a.p.amplicons != 0, inner = a.p.is_inner != 0;
    if (s1 > 0) {
        const int64_t far_outer = pos + d - 1;
        if (pd.strand0 == pd.strand1) {
            if (pd.strand0 == 0) {
                if (j == 0) { *start = amp ? sl - 1 : (inner ? pos + s1 + d - 1 : pos + d - s0); *step = 1; }
                else { *start = pos; *step = 1; }
            } else {
                if (j == 0) { *start = pos + s0 - 1; *step = -1; }
                else { *start = amp ? sl - 1 : (inner ? pos + s0 + d + s1 - 1 : far_outer); *step = -1; }
            }
        } else {
            if (pd.strand0 == 0) {
                if (j == 0) { *start = pos; *step = 1; }
                else { *start = amp ? sl - 1 : (inner ? pos + s0 + d + s1 - 1 : far_outer); *step = -1; }
            } else {
                if (j == 0) { *start = amp ? sl - 1 : (inner ? pos + s1 + d + s0 - 1 : far_outer); *step = -1; }
                else { *start = pos; *step = 1; }
            }
        }
    } else {
        if (pd.strand0 == 0) { *start = pos; *step = 1; }
        else if (amp) { *start = sl - 1; *step = -1; }
        else { *start = pos + s0 - 1; *step = -1; }
    }
}

#if DW_HAS(0)
// Haplotype summary for k_place: one word per SUMM_CELLS cells -- how many of them are INSERT / DELETE cells (bit 4 of the cell),
// and whether any holds a base code >= 4 (N, '-').
__global__ void __launch_bounds__(256) k_summarize(const uint8_t *cells, int64_t l, uint16_t *summ)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x, first = b * SUMM_CELLS;
    if (first >= l) return;
    uint32_t indel = 0, non_acgt = 0;
#pragma unroll
    for (int q = 0; q < SUMM_CELLS / 16; ++q) {
        const int64_t at = first + 16 * q;
        if (at >= l) break;
        const uint4 v = *reinterpret_cast<const uint4 *>(cells + at);          // cells are readable (padded) up to a multiple of 16 past l
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t rem = l - (at + 4 * k);                              // cells of this word that belong to the contig
            const uint32_t live = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : ((1u << (8 * (int)rem)) - 1u);
            indel += (uint32_t)__popc(w[k] & live & 0x10101010u);
            non_acgt |= w[k] & live & 0x0C0C0C0Cu;
        }
    }
    summ[b] = (uint16_t)(indel | (non_acgt ? 0x8000u : 0u));
}

// A sufficient condition for an attempt to be accepted (dwgsim.c:824-843) without walking the read: take the 2s+3 cells from
// `start` in travel direction.  If they all lie inside the contig, none holds a base code >= 4 and at most s of them are
// INSERT / DELETE cells, then the walk of __gen_read (one base per NOCHANGE / SUBSTITUTE cell, inserted bases are never N)
// collects its s bases within those cells: ext_coor >= 0, k == s, num_n == 0 <= max_n.  Evaluated on the block summaries that
// cover the window (a superset, so still sufficient); anything else falls back to the exact walk.
DW_DEV bool attempt_surely_accepted(const uint16_t *summ, int64_t l, int64_t start, int step, int s)
{
    const int64_t far = start + (int64_t)step * (2 * (int64_t)s + 2);
    const int64_t lo = step > 0 ? start : far, hi = step > 0 ? far : start;
    if (lo < 0 || hi >= l) return false;
    uint32_t indel = 0, flags = 0;
    for (int64_t b = lo / SUMM_CELLS; b <= hi / SUMM_CELLS; ++b) { const uint32_t v = summ[b]; indel += v & 0xffu; flags |= v; }
    return !(flags & 0x8000u) && indel <= (uint32_t)s;
}

// K5: per pair, the attempt that is accepted (dwgsim.c:833-843 retry rule) and the random-read flag.
// One lane per read end (LPP = 2: lanes 2q / 2q+1 test the two ends of pair q and exchange the verdict).
template <int LPP>
__global__ void __launch_bounds__(PAIRS_PER_BLOCK *LPP) k_place(SimArgs a)
{
    __shared__ uint32_t sm[17];
    const int tid = (int)threadIdx.x, j = (LPP == 2) ? (tid & 1) : 0;
    const uint64_t pair = (uint64_t)blockIdx.x * PAIRS_PER_BLOCK + (uint64_t)(tid / LPP);
    const bool valid = pair < a.n_pairs;
    const uint64_t ii = a.first_ii + pair;
    const RngKey key{a.p.seed, a.c.contig_index};
    const int sj = sel_len(a, j);
    uint32_t att = 0; bool is_rand = false, failed = false, done = !valid;
    while (__ballot(!done)) {                      // wave-uniform loop: the two lanes of a pair always agree on `done`
        bool ok = true;
        if (!done) {
            const PairDraw pd = draw_pair(a, key, ii, att);
            if (pd.is_rand) { is_rand = true; done = true; }
            else if (sj > 0) {
                int64_t start; int step;
                read_geom(a, pd, j, &start, &step);
                if (!attempt_surely_accepted(pd.hap ? a.summ[1] : a.summ[0], a.c.l, start, step, sj)) {     // rare: N, dense indels, contig ends
                    const ReadRes r = gen_read<false>(sel_hap(a, pd.hap), a.c.l, start, step, sj, j ? pd.strand1 : pd.strand0, nullptr, 0);
                    ok = r.ext_coor >= 0 && r.num_n <= a.p.max_n;
                }
            }
        }
        if (LPP == 2) { const int other = __shfl_xor((int)ok, 1); ok = ok && (other != 0); }   // every lane shuffles (no short-circuit)
        if (!done) {
            if (ok) done = true;
            else if (++att > (uint32_t)MAX_ATTEMPTS) { failed = true; done = true; }
        }
    }
    if (valid && j == 0) a.meta[pair] = att | (is_rand ? 0x80000000u : 0u);
    uint32_t total;
    (void)block_excl_scan((is_rand && j == 0) ? 1u : 0u, sm, &total);
    if (threadIdx.x == 0) a.block_rand[blockIdx.x] = total;
    const uint32_t retries = wave_sum_u32((valid && j == 0) ? att : 0u);
    if (lane_id() == 0 && retries) atomicAdd((unsigned long long *)&a.counters[1], (unsigned long long)retries);
    if (failed) atomicOr((unsigned long long *)&a.counters[2], 1ull);
}

#endif // DW_HAS(0): k_place

// ---- Ion Torrent flow-space errors: dwgsim.c:246-417 generate_errors_flows ----
// The reference edits the read in place; both passes only ever insert/delete at the position being
// examined, so they are replayed as transducers over packed 4-bit arrays in LDS (word w of a lane
// at base[w * stride]).  Draws: narrow uniforms of domain D_FLOW0 + read end, one sequential slot
// counter per read end.  The flow mask is per read (the reference's persistent mask is fully
// rewritten by every read's pass 1).
// word-cached access to a lane's packed array (BITS = 4: codes 0-5, BITS = 2: bases 0-3): the flow model reads and
// appends sequentially, so one LDS access serves 8 / 16 bases
template <int BITS>
struct PackReader {
    static constexpr int PER = 32 / BITS, SH = BITS == 4 ? 3 : 4; static constexpr uint32_t M = (1u << BITS) - 1;
    const uint32_t *base; int stride, cw; uint32_t word;
    DW_DEV void init(const uint32_t *b, int st) { base = b; stride = st; cw = -1; word = 0; }
    DW_DEV uint32_t get(int i) { const int w = i >> SH; if (w != cw) { cw = w; word = base[w * stride]; } return (word >> ((i & (PER - 1)) * BITS)) & M; }
};
template <int BITS>
struct PackAppender {
    static constexpr int PER = 32 / BITS, SH = BITS == 4 ? 3 : 4;
    uint32_t *base; int stride, n; uint32_t acc;
    DW_DEV void init(uint32_t *b, int st) { base = b; stride = st; n = 0; acc = 0; }
    DW_DEV void push(uint32_t v) { acc |= v << ((n & (PER - 1)) * BITS); if ((++n & (PER - 1)) == 0) { base[((n >> SH) - 1) * stride] = acc; acc = 0; } }
    DW_DEV void flush() { if (n & (PER - 1)) base[(n >> SH) * stride] = acc; }
};
struct FlowRng {             // scalar members + value selects only: keeps the generator state in registers
    uint32_t seed, contig, dom, att, slot, w0, w1, w2, w3; uint64_t ii;
    DW_DEV uint32_t next()
    {
        if ((slot & 3) == 0) { const U4 b = rng_block(RngKey{seed, contig}, dom, ii, att, 0, slot >> 2); w0 = b.x; w1 = b.y; w2 = b.z; w3 = b.w; }
        const uint32_t k = slot & 3; ++slot;
        const uint32_t lo = (k & 1) ? w1 : w0, hi = (k & 1) ? w3 : w2;
        return (k & 2) ? hi : lo;
    }
    DW_DEV int geometric(uint64_t thr) { int n = 0; while ((uint64_t)next() < thr) ++n; return n; }   // while (drand48() < e) n_err++
};
// Returns the new length, -1 if a buffer / the pass-2 stack overflowed or the read degenerated.  The final read is left in
// bufA (4-bit) in the orientation of the flow model; a reverse-strand read is turned back by the caller when it is read
// (dwgsim.c:408-414).  bufB: pass-1 output at 2 bits per base; stk: 8 (base, count) runs, two per word.
DW_DEV int flow_errors(FlowRng &rg, const uint8_t *flow, int F, uint64_t thr, uint32_t *bufA, uint32_t *bufB, uint32_t *stk, int stride,
                       int len, int strand, int cap, int32_t *n_err_out)
{
    // input = bufA (len bases, read back-to-front when strand == 1, N -> A: dwgsim.c:253-265), pass 1 -> bufB, pass 2 -> bufA
    PackReader<4> rd, la; rd.init(bufA, stride); la.init(bufA, stride);
    auto in = [&](PackReader<4> &r, int t) -> uint32_t { const uint32_t v = r.get(strand ? len - 1 - t : t); return v >= 4 ? 0u : v; };
    uint64_t mask = 0; int flow_i = 0, total = 0;
    { const uint32_t c0 = in(rd, 0); while (flow_i < F && c0 != flow[flow_i]) ++flow_i; if (flow_i == F) return -1; }
    // ---- pass 1 (dwgsim.c:281-364): one error event per homopolymer start ----
    PackAppender<2> o1; o1.init(bufB, stride);
    int t = 0; uint32_t prev_c = 4, pend_c = 0; int pend_n = 0;
    for (;;) {
        uint32_t c; bool from_pend = false;
        if (pend_n > 0) { c = pend_c; from_pend = true; } else if (t < len) c = in(rd, t); else break;
        if (o1.n >= cap) return -1;
        while (c != flow[flow_i]) { mask &= ~(1ull << flow_i); flow_i = flow_i + 1 == F ? 0 : flow_i + 1; }
        if (prev_c != c) {
            mask &= ~(1ull << flow_i);
            int n_err = rg.geometric(thr);
            if (n_err > 0) {
                if (rg.next() < 0x80000000u) {                  // insert n_err copies in front of the homopolymer
                    o1.push(c); pend_c = c; pend_n = n_err - 1;
                    total += n_err; prev_c = c;
                    continue;
                }
                int hp_l = 0; uint32_t next_c = c;              // delete: bounded by the homopolymer length
                while (t + hp_l < len) { next_c = in(la, t + hp_l); if (next_c != c) break; ++hp_l; }
                if (n_err > hp_l) n_err = hp_l;
                t += n_err; mask |= 1ull << flow_i; total += n_err;
                if (n_err == hp_l && (o1.n == 0 || prev_c == next_c)) {   // dot-fill (dwgsim.c:342-358)
                    if (next_c == c) return -1;                // the whole read was one deleted homopolymer (the reference asserts)
                    int jj = 0; while (next_c != flow[(flow_i + jj) % F]) ++jj;
                    const int kk = (int)(((uint64_t)rg.next() * (uint64_t)jj) >> 32);   // (int)(drand48() * j)
                    o1.push(flow[(flow_i + kk) % F]);
                } else if (t < len) { o1.push(in(rd, t)); ++t; }   // the base now at this position is not examined
                prev_c = c;
                continue;
            }
            prev_c = c;
        }
        o1.push(c);
        if (from_pend) --pend_n; else ++t;
    }
    o1.flush();
    const int n1 = o1.n;
    // ---- pass 2 (dwgsim.c:367-406): insertions in empty flows; inserted bases are examined again later, the examined base
    // itself stays behind them: a stack of (base, count) runs on top of the pass-1 output reproduces the in-place order. ----
    PackReader<2> r2; r2.init(bufB, stride);
    PackAppender<4> o2; o2.init(bufA, stride);
    auto stk_get = [&](int k) -> uint32_t { return (stk[(k >> 1) * stride] >> ((k & 1) * 16)) & 0xffffu; };
    auto stk_set = [&](int k, uint32_t v) { const uint32_t sh = (uint32_t)(k & 1) * 16; uint32_t w = stk[(k >> 1) * stride]; stk[(k >> 1) * stride] = (w & ~(0xffffu << sh)) | (v << sh); };
    int t2 = 0, sp = 0;
    for (;;) {
        uint32_t x;
        if (sp > 0) x = stk_get(sp - 1) >> 14; else if (t2 < n1) x = r2.get(t2); else break;
        if (o2.n >= cap) return -1;
        while (x != flow[flow_i]) {                 // empty flows in front of the examined base: each may insert
            const int n_err = rg.geometric(thr);
            if (!((mask >> flow_i) & 1) && n_err > 0) {
                if (sp >= 8 || n_err >= (1 << 14)) return -1;
                stk_set(sp, ((uint32_t)flow[flow_i] << 14) | (uint32_t)n_err); ++sp;
                total += n_err;
            }
            flow_i = flow_i + 1 == F ? 0 : flow_i + 1;
        }
        if (sp == 0) { o2.push(x); ++t2; }          // nothing in front of it: the base itself becomes final
        else {                                      // the first base of the top run (the examined base stays behind it)
            const uint32_t top = stk_get(sp - 1);
            o2.push(top >> 14);
            if ((top & 0x3fffu) <= 1) --sp; else stk_set(sp - 1, top - 1);
        }
    }
    o2.flush();
    *n_err_out += total;
    return o2.n;
}

// ---- FASTQ text assembly ----
struct Writer {               // sequential byte stream -> bursts of 64-byte aligned chunks
    // A lane's record is cut at 64-byte boundaries of the output buffer; a chunk is assembled in registers
    // (three finished 16-byte sub-blocks in s0..s5, the one being filled in lo/hi) and leaves as four
    // back-to-back dwordx4 stores, so L2 sees whole 64-byte request units instead of 16-byte crumbs
    // (partially written lines were being evicted and written back 2.6x, profiles/r01_p3).
    // Only the first / last chunk of a record is ragged: bytes [skip, upto) go out as dwords / bytes.
    uint8_t *blk; uint64_t lo, hi, s0, s1, s2, s3, s4, s5; uint32_t n, sub, skip;
    DW_DEV void init(uint8_t *p)
    {
        const uint32_t o = (uint32_t)((uintptr_t)p & 63);
        blk = p - o; sub = o >> 4; n = o & 15; skip = o;
        lo = hi = s0 = s1 = s2 = s3 = s4 = s5 = 0;
    }
    static DW_DEV void store16(uint8_t *dst, uint64_t a, uint64_t b, uint32_t from, uint32_t upto)   // bytes [from, upto) of a 16-byte block
    {
        if (from == 0 && upto == 16) { *reinterpret_cast<uint4 *>(dst) = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)); return; }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t w = (uint32_t)((q < 2 ? a : b) >> (32 * (q & 1)));
            const uint32_t b0 = 4 * q, b1 = b0 + 4;
            if (from <= b0 && upto >= b1) *reinterpret_cast<uint32_t *>(dst + b0) = w;
            else for (uint32_t k = b0; k < b1; ++k) if (k >= from && k < upto) dst[k] = (uint8_t)(w >> (8 * (k - b0)));
        }
    }
    DW_DEV void store_chunk(uint32_t upto)       // bytes [skip, upto) of the current chunk
    {
        if (skip == 0 && upto == 64) {           // the common case: one 64-byte burst
            uint4 *d = reinterpret_cast<uint4 *>(blk);
            d[0] = make_uint4((uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)s1, (uint32_t)(s1 >> 32));
            d[1] = make_uint4((uint32_t)s2, (uint32_t)(s2 >> 32), (uint32_t)s3, (uint32_t)(s3 >> 32));
            d[2] = make_uint4((uint32_t)s4, (uint32_t)(s4 >> 32), (uint32_t)s5, (uint32_t)(s5 >> 32));
            d[3] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
            return;
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t b0 = 16 * q;
            if (upto <= b0 || skip >= b0 + 16) continue;
            const uint64_t a = (q == sub) ? lo : (q == 0 ? s0 : q == 1 ? s2 : s4);
            const uint64_t b = (q == sub) ? hi : (q == 0 ? s1 : q == 1 ? s3 : s5);
            const uint32_t from = skip > b0 ? skip - b0 : 0, to = upto < b0 + 16 ? upto - b0 : 16;
            store16(blk + b0, a, b, from, to);
        }
    }
    DW_DEV void advance()                        // the 16-byte sub-block in lo/hi is complete
    {
        if (sub == 3) { store_chunk(64); blk += 64; sub = 0; skip = 0; }
        else {      // value selects, not conditional stores: keeps s0..s5 in registers
            const bool z0 = sub == 0, z1 = sub == 1, z2 = sub == 2;
            s0 = z0 ? lo : s0; s1 = z0 ? hi : s1; s2 = z1 ? lo : s2; s3 = z1 ? hi : s3; s4 = z2 ? lo : s4; s5 = z2 ? hi : s5;
            ++sub;
        }
        lo = hi = 0; n = 0;
    }
    DW_DEV void put(uint32_t b)
    {
        const uint64_t v = (uint64_t)b << (8 * (n & 7));
        if (n < 8) lo |= v; else hi |= v;
        if (++n == 16) advance();
    }
    DW_DEV void putn(uint64_t v, uint32_t cnt)   // cnt (1..8) bytes, little-endian in v, upper bytes zero
    {
        const uint32_t sh = 8 * (n & 7);
        if (n < 8) { lo |= v << sh; if (sh) hi |= v >> (64 - sh); }
        else hi |= v << sh;
        const uint32_t total = n + cnt;
        if (total >= 16) {
            const uint32_t over = total - 16;       // bytes that belong to the next sub-block (0..7)
            const uint64_t carry = over ? v >> (8 * (cnt - over)) : 0;
            advance();
            lo = carry; n = over;
        } else n = total;
    }
    DW_DEV void put4(uint32_t w) { putn((uint64_t)w, 4); }
    DW_DEV void flush() { const uint32_t upto = 16 * sub + n; if (upto > skip) store_chunk(upto); }
};
template <int OUT>            // OUT bit 0: the bwa stream of this read end, bit 1: the interleaved bfast stream
struct Out2 {
    Writer a, b;
    DW_DEV void put(uint32_t c) { if (OUT & 1) a.put(c); if (OUT & 2) b.put(c); }
    DW_DEV void put4(uint32_t w) { if (OUT & 1) a.put4(w); if (OUT & 2) b.put4(w); }
    DW_DEV void putn(uint64_t v, uint32_t cnt) { if (OUT & 1) a.putn(v, cnt); if (OUT & 2) b.putn(v, cnt); }
    DW_DEV void flush() { if (OUT & 1) a.flush(); if (OUT & 2) b.flush(); }
};
DW_DEV uint32_t ndigits10(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}
DW_DEV uint32_t ndigits16(uint64_t v) { return v ? (uint32_t)(67 - __clzll((long long)v)) >> 2 : 1u; }
// decimal digits of v as packed ASCII, most significant digit in the lowest byte (stream order);
// lead = one separator byte to emit in front (0 = none).  Numbers above 10^7 take the two-part path.
template <int OUT>
DW_DEV void put_dec(Out2<OUT> &o, uint32_t v, uint32_t lead)
{
    uint32_t low7 = 0; bool big = false;
    if (v >= 10000000u) { const uint32_t hi = v / 10000000u; low7 = v - hi * 10000000u; v = hi; big = true; }   // 8..10 digits
    uint64_t w = 0; uint32_t nd = 0;
    do { const uint32_t q = v / 10u; w = (w << 8) | ('0' + (v - q * 10u)); v = q; ++nd; } while (v);
    if (lead) { w = (w << 8) | lead; ++nd; }
    o.putn(w, nd);
    if (big) {                                 // the low seven digits, zero padded
        w = 0;
        for (int d = 0; d < 7; ++d) { const uint32_t q = low7 / 10u; w = (w << 8) | ('0' + (low7 - q * 10u)); low7 = q; }
        o.putn(w, 7);
    }
}
template <int OUT>
DW_DEV void put_hex(Out2<OUT> &o, uint64_t v)
{
    const uint32_t nd = ndigits16(v);
    for (uint32_t part = 0; part < 2; ++part) {      // up to 16 digits: the high (nd-8) first, then the low 8
        const uint32_t cnt = part == 0 ? (nd > 8 ? nd - 8 : 0) : (nd > 8 ? 8 : nd);
        if (!cnt) continue;
        const uint64_t x = part == 0 ? v >> 32 : (nd > 8 ? (v & 0xFFFFFFFFull) : v);
        uint64_t w = 0;
        for (uint32_t d = 0; d < cnt; ++d) { const uint32_t hx = (uint32_t)(x >> (4 * d)) & 15u; w = (w << 8) | (hx < 10 ? '0' + hx : 'a' + (hx - 10)); }
        o.putn(w, cnt);
    }
}
DW_DEV uint32_t base_char(uint32_t v) { return (uint32_t)((0x4E4E4E4E54474341ull >> (8 * (v & 7))) & 0xff); }  // "ACGTNNNN"
DW_DEV uint32_t base_chars4(uint32_t nibbles) { return lut8(0x4E4E4E4Eu, 0x54474341u, spread4(nibbles)); }        // four codes (<= 7) -> "ACGTNNNN"[code]
DW_DEV uint32_t colour_digits4(uint32_t nibbles) { return lut8(0x34343434u, 0x33323130u, spread4(nibbles)); }     // four colours -> "01234444"[colour]

// K6: one lane per read end (LPP = 2: lanes 2q / 2q+1 are the two ends of pair q; LPP = 1: single end).
// Opt-in phase timing (tools/phase_profile.sh builds a separate library with -DDW_PHASE_TIMING; the
// product build compiles these macros to nothing): per wave, shader-clock ticks spent in each phase
// are added to counters[8 + phase].
#ifdef DW_PHASE_TIMING
#define PH_INIT() uint64_t ph_t = __builtin_amdgcn_s_memtime()
#define PH_MARK(k) do { const uint64_t ph_n = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&a.counters[8 + (k)], (unsigned long long)(ph_n - ph_t)); ph_t = ph_n; } while (0)
#else
#define PH_INIT() do { } while (0)
#define PH_MARK(k) do { } while (0)
#endif

// ---- pieces of a FASTQ record shared by the Illumina / Ion Torrent and the SOLiD write paths ----
// '@' + "[prefix_]contig" (or "[prefix_]rand"): whole words from LDS (first 256 bytes), any rest from HBM
template <int OUT>
DW_DEV void put_name_fixed(Out2<OUT> &o, const uint32_t *fw, const uint8_t *fx, uint32_t fixed_len)
{
    const uint32_t flen = fixed_len + 1, inl = flen < 256u ? flen : 256u;
    uint32_t q = 0;
    for (; q + 4 <= inl; q += 4) o.put4(fw[q >> 2]);
    if (q < inl) o.putn((uint64_t)fw[q >> 2] & ((1ull << (8 * (inl - q))) - 1), inl - q);
    for (q = inl; q < flen; ++q) o.put(fx[q]);
}
// "_0_0_0_0_1_1_0:0:0_0:0:0_<hex>" of a random read (dwgsim.c:1044-1048)
template <int OUT>
DW_DEV void put_rand_tail(Out2<OUT> &o, uint64_t rand_ii)
{
    o.putn(0x305F305F305F305Full, 8);          // "_0_0_0_0"
    o.putn(0x3A305F315F315F00ull >> 8, 7);     // "_1_1_0:"
    o.putn(0x303A305F303A30ull, 7);            // "0:0_0:0"
    o.putn(0x5F303Aull, 3);                    // ":0_"
    put_hex(o, rand_ii);
}
struct NameCounts { int32_t e0, u0, i0, e1, u1, i1; };     // n_err : n_sub : n_indel of read end 1 and 2
// "_pos1_pos2_strand1_strand2_0_0_e:s:i_e:s:i_<hex>" (dwgsim.c:923-929)
template <int OUT>
DW_DEV void put_pair_tail(Out2<OUT> &o, int32_t x0, int32_t x1, uint32_t strand0, uint32_t strand1, const NameCounts &n, uint64_t ii)
{
    put_dec(o, (uint32_t)(x0 + 1), '_'); put_dec(o, (uint32_t)(x1 + 1), '_');
    o.putn((uint64_t)'_' | ((uint64_t)('0' + strand0) << 8) | ((uint64_t)'_' << 16) | ((uint64_t)('0' + strand1) << 24)
               | ((uint64_t)'_' << 32) | ((uint64_t)'0' << 40) | ((uint64_t)'_' << 48) | ((uint64_t)'0' << 56), 8);     // "_S_S_0_0"
    put_dec(o, (uint32_t)n.e0, '_'); put_dec(o, (uint32_t)n.u0, ':'); put_dec(o, (uint32_t)n.i0, ':');
    put_dec(o, (uint32_t)n.e1, '_'); put_dec(o, (uint32_t)n.u1, ':'); put_dec(o, (uint32_t)n.i1, ':');
    o.put('_');
    put_hex(o, ii);
}
DW_DEV uint32_t pair_tail_len(int32_t x0, int32_t x1, const NameCounts &n, uint64_t ii)
{
    return 1 + ndigits10((uint32_t)(x0 + 1)) + 1 + ndigits10((uint32_t)(x1 + 1)) + 9     // _P0_P1 _S_S_0_0_
         + ndigits10((uint32_t)n.e0) + 1 + ndigits10((uint32_t)n.u0) + 1 + ndigits10((uint32_t)n.i0) + 1
         + ndigits10((uint32_t)n.e1) + 1 + ndigits10((uint32_t)n.u1) + 1 + ndigits10((uint32_t)n.i1) + 1 + ndigits16(ii);
}
// Quality characters of one read end, in order (dwgsim.c:899-918): emit(i, q) for i = 0 .. n - 1.  qb = base quality per position
// (positions >= nq reuse the last entry: Ion Torrent reads can outgrow the table, their error rate is uniform, dwgsim_opt.c:338-343).
template <class F>
DW_DEV void for_each_quality(const SimParams &p, RngKey key, uint32_t dom, uint64_t ii, uint32_t att, const int8_t *qb, int nq, int n, F &&emit)
{
    if (p.fixed_quality >= 0) { for (int i = 0; i < n; ++i) emit(i, (uint32_t)p.fixed_quality); return; }
    if (!(0 < p.quality_std)) {
        for (int i = 0; i < n; ++i) { int32_t q = qb[i < nq ? i : nq - 1]; if (q < 33) q = 33; if (q > 73) q = 73; emit(i, (uint32_t)q); }
        return;
    }
    uint32_t m = 0; int pr = 0; const int np = (n + 1) >> 1;
    while (pr < np) {
        // (the two base qualities are fetched before the arithmetic that hides their latency)
        const int i0 = 2 * pr, i1 = 2 * pr + 1;
        const int32_t qb0 = qb[i0 < nq ? i0 : nq - 1], qb1 = qb[i1 < nq ? i1 : nq - 1];
        const U4 blk = rng_block(key, dom, ii, att, m, (uint32_t)pr);
        // two polar tries per block (narrow uniforms): v = 2 * (w * 2^-32) - 1 = w * 2^-31 - 1, exact
        const double a1 = (double)blk.x * 0x1p-31 - 1.0, a2 = (double)blk.y * 0x1p-31 - 1.0;
        const double b1 = (double)blk.z * 0x1p-31 - 1.0, b2 = (double)blk.w * 0x1p-31 - 1.0;
        const double ra = a1 * a1 + a2 * a2, rb = b1 * b1 + b2 * b2;
        const bool oka = !(ra >= 1.0 || ra == 0.0), okb = !(rb >= 1.0 || rb == 0.0);
        if (!oka && !okb) { ++m; continue; }
        const double v1 = oka ? a1 : b1, v2 = oka ? a2 : b2, rsq = oka ? ra : rb;
        // rsq is a multiple of 2^-62 in (0, 1): -2 log(rsq) in [2^-52, 86], the quotient in [2^-52, 2^69] -- the range-restricted forms apply
        const double fac = sqrt_mid(div_mid(-2.0 * det_log<true>(rsq), rsq));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * pr + h;
            const double nrm = (h ? v1 : v2) * fac;       // first normal of the pair is v2*fac (dwgsim.c:170), the cached one v1*fac
            if (i < n) {
                int32_t q = (int8_t)((h ? qb1 : qb0) + (int32_t)((nrm * p.quality_std) + 0.5));
                if (q < 33) q = 33;
                if (q > 73) q = 73;
                emit(i, (uint32_t)q);
            }
        }
        ++pr; m = 0;
    }
}

// DT = 0: Illumina base-space errors; DT = 2: Ion Torrent flow-space errors (variable read length).
template <int LPP, int OUT, int DT>
__global__ void __launch_bounds__(SIM_THREADS, (DT == 2 ? DW_ION_WAVES : DT == 1 ? 1 : OUT != 3 ? DW_SIM_WAVES : DW_SIM_WAVES_BOTH)) k_simulate(SimArgs a)
{
    DW_DYN_SHARED(uint32_t, dyn_lds);                                    // [lds_words][blockDim] packed bases
    __shared__ uint32_t sm_rand[1][16], sm_bytes[3][16];     // one scratch area per scan: each is written once
    __shared__ uint32_t s_ticket;
    __shared__ uint64_t s_rbase, s_base[3];
    __shared__ uint32_t s_fixed[2][64];          // "@[prefix_]contig" and "@[prefix_]rand", first 256 bytes
    __shared__ uint8_t s_flow[64];               // Ion Torrent flow order
    constexpr int nthr = SIM_THREADS, PPB = SIM_THREADS / LPP, nwaves = SIM_THREADS / 64;      // PPB pairs per block
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    PH_INIT();
    if (tid == 0) s_ticket = (uint32_t)atomicAdd((unsigned long long *)&a.counters[0], 1ull);
    for (int q = tid; q < 128; q += nthr)                                                          // buffers are padded to 256 + 16 bytes
        (&s_fixed[0][0])[q] = q < 64 ? reinterpret_cast<const uint32_t *>(a.name_fixed)[q] : reinterpret_cast<const uint32_t *>(a.rand_fixed)[q - 64];
    if (DT == 2 && tid < 64) s_flow[tid] = a.flow[tid];
    __syncthreads();
    const uint32_t t = s_ticket;                                  // logical block: predecessors have started
    const int j = (LPP == 2) ? (tid & 1) : 0;
    const uint64_t pair = (uint64_t)t * PPB + (uint64_t)(tid / LPP);
    const bool valid = pair < a.n_pairs;
    const uint64_t ii = a.first_ii + pair;
    const RngKey key{a.p.seed, a.c.contig_index};
    const int s = sel_len(a, j);
    // this lane's packed bases: word w at lds[w * nthr].  Illumina: LDS.  Ion Torrent: the (much larger, sequentially accessed)
    // read buffers live in a global scratch so that LDS does not cap residency; only the 4-word run stack stays in LDS
    uint32_t *lds = (DT == 2) ? a.flow_scratch + (size_t)t * ((size_t)(a.lds_words + ((a.cap + 15) >> 4)) * nthr) + tid : dyn_lds + tid;

    PH_MARK(0);     // ticket, fixed strings
    // ---- attempts until the pair is accepted (dwgsim.c:649-843): placement, haplotype, strands, base extraction of this
    // read end, N filter; the two lanes of a pair exchange their verdicts and retry together with attempt + 1 ----
    PairDraw pd; pd.is_rand = true; pd.pos = pd.d = 0; pd.hap = 0; pd.strand0 = pd.strand1 = 0;
    ReadRes rr{0, 0, 0, 0, 0};
    uint32_t att = 0; bool is_rand = false, done = !valid;
    while (__ballot(!done)) {
        bool ok = true;
        if (!done) {
            pd = draw_pair(a, key, ii, att);
            if (pd.is_rand) { is_rand = true; done = true; rr = ReadRes{0, 0, 0, 0, 0}; }
            else if (s > 0) {
                int64_t start; int step;
                read_geom(a, pd, j, &start, &step);
                PH_MARK(7);     // placement draws (phase 1 below is then the base extraction alone)
                rr = gen_read<true>(sel_hap(a, pd.hap), a.c.l, start, step, s, j ? pd.strand1 : pd.strand0, lds, nthr);
                ok = rr.ext_coor >= 0 && rr.num_n <= a.p.max_n;
            }
        }
        if (LPP == 2) { const int other = __shfl_xor((int)ok, 1); ok = ok && (other != 0); }   // every lane shuffles
        if (!done) {
            if (ok) done = true;
            else if (++att > (uint32_t)MAX_ATTEMPTS) { atomicOr((unsigned long long *)&a.counters[2], 1ull); done = true; }
        }
    }
    { const uint32_t retries = wave_sum_u32((valid && j == 0) ? att : 0u); if (lane == 0 && retries) atomicAdd((unsigned long long *)&a.counters[1], (unsigned long long)retries); }
    // running random-read index (dwgsim.c:1042,1096): look-back over the blocks' random counts + rank inside the block
    uint32_t rrank, rtot;
    { const uint32_t v[1] = {(is_rand && j == 0) ? 1u : 0u}; uint32_t ex[1], tot[1]; block_excl_scan_n<1>(v, sm_rand, ex, tot); rrank = ex[0]; rtot = tot[0]; }
    if (wave == 0) {
        const uint64_t g = lookback_excl(a.status[2], t, rtot, 0);
        if (lane == 0) { s_rbase = g; if ((uint64_t)t + 1 == (a.n_pairs + PPB - 1) / PPB) a.counters[3] = g + rtot; }
    }
    // (the barrier that publishes s_rbase comes after the error phase, which does not need the index: the look-back's latency
    // overlaps with that work instead of idling three waves)
    PH_MARK(1);     // placement + base extraction
    // ---- sequencing errors (dwgsim.c:233-244) or random bases (dwgsim.c:999-1001) ----
    // narrow draws: one Philox block tests four bases; an error marks bit 3 of the base's nibble and its
    // substituted base is drawn afterwards, only for the (few) marked bases
    int32_t n_err = 0;
    int s_out = s;                              // read length after errors (changes only for Ion Torrent)
    bool flow_reversed = false;
    const int nw = (s + 7) >> 3;
    if (DT == 2 && valid && !is_rand && s > 0) {  // dwgsim.c:861-864
        FlowRng rg; rg.seed = key.seed; rg.contig = key.contig; rg.dom = D_FLOW0 + (uint32_t)j; rg.att = att; rg.slot = 0; rg.ii = ii; rg.w0 = rg.w1 = rg.w2 = rg.w3 = 0;
        s_out = flow_errors(rg, s_flow, a.flow_len, (j ? a.e_thr[1] : a.e_thr[0])[0], lds, lds + (size_t)a.lds_words * nthr, dyn_lds + tid,
                            nthr, s, j ? pd.strand1 : pd.strand0, a.cap, &n_err);
        if (s_out < 0) { atomicOr((unsigned long long *)&a.counters[2], 2ull); s_out = 0; }
        flow_reversed = (j ? pd.strand1 : pd.strand0) != 0;     // the read is turned back while it is written (dwgsim.c:408-414)
    }
    int32_t err_first = 0;                      // SOLiD: an error on the first colour (n_err_first, dwgsim.c:240)
    if (valid && (DT != 2 || is_rand)) {
        // eight bases (one staged word) at a time: nibble-parallel N clamp / colour conversion, eight 32-bit threshold compares
        const uint32_t *thr = j ? a.e_thr32[1] : a.e_thr32[0];
        uint32_t prev_base = 0;                 // SOLiD: previous base in base space; the adaptor counts as 'A' (dwgsim.c:849)
        for (int w = 0; w < nw; ++w) {
            uint4 ta = make_uint4(0, 0, 0, 0), tb = ta;
            if (!is_rand) { ta = *reinterpret_cast<const uint4 *>(thr + 8 * w); tb = *reinterpret_cast<const uint4 *>(thr + 8 * w + 4); }
            const U4 q0 = rng_block(key, D_BASE0 + (uint32_t)j, ii, att, 0, (uint32_t)(2 * w));
            const U4 q1 = rng_block(key, D_BASE0 + (uint32_t)j, ii, att, 0, (uint32_t)(2 * w + 1));
            const uint32_t rw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const int rem = s - 8 * w;
            const uint32_t live = rem >= 8 ? 0xFFFFFFFFu : ((1u << (4 * rem)) - 1u);      // nibbles of bases i < s
            uint32_t word;
            if (is_rand) {                                                      // random read: base = (int)(u * 4.0) & 3 (dwgsim.c:999-1001)
                word = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) word |= (rw[b] >> 30) << (4 * b);
            } else word = lds[w * nthr];
            if (DT == 1) {                                                      // colour = __gf_add(previous base, base): dwgsim.h:6, dwgsim.c:845-858 / :1022-1032
                const uint32_t prevw = (word << 4) | prev_base;
                prev_base = word >> 28;
                const uint32_t n = (word | prevw) & 0x44444444u;                // either base is not ACGT -> colour 4
                word = ((word ^ prevw) & 0x33333333u & ~((n >> 1) | (n >> 2))) | n;
            } else {
                const uint32_t n4 = word & 0x44444444u;                         // if (c >= 4) c = 4 (dwgsim.c:235)
                word &= ~((n4 >> 1) | (n4 >> 2));
            }
            if (!is_rand) {                                                     // drand48() < e[i]  <=>  w < thr[i]; an error marks bit 3 of the nibble
                const uint32_t t[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
                uint32_t hits = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) hits |= (rw[b] < t[b]) ? (8u << (4 * b)) : 0u;
                if (a.e_full) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) hits |= (t[b] == 0xFFFFFFFFu) ? (8u << (4 * b)) : 0u;
                }
                hits &= ~((word & 0x44444444u) << 1) & live;                    // N bases / colours take no error
                n_err += __popc(hits);
                if (DT == 1 && w == 0) err_first = (int32_t)((hits >> 3) & 1u);
                word |= hits;
            }
            lds[w * nthr] = word & live;
        }
        if (!is_rand) {
            int w = 0; uint32_t pend = 0;
            for (;;) {
                while (pend == 0 && w < nw) { pend = lds[w * nthr] & 0x88888888u; if (!pend) ++w; }
                if (!pend) break;
                const int b = (__ffs((int)pend) - 1) >> 2;
                pend &= pend - 1;
                const int i = w * 8 + b;
                const U4 q = rng_block(key, D_SUB0 + (uint32_t)j, ii, att, 0, (uint32_t)(i >> 2));
                const uint32_t rwd = (i & 2) ? ((i & 1) ? q.w : q.z) : ((i & 1) ? q.y : q.x);
                const uint32_t add = 1u + (uint32_t)(((uint64_t)rwd * 3u) >> 32);        // (int)(u * 3.0 + 1), exact
                uint32_t word = lds[w * nthr];
                const uint32_t c = (((word >> (4 * b)) & 7u) + add) & 3u;
                word = (word & ~(0xFu << (4 * b))) | (c << (4 * b));
                lds[w * nthr] = word;
                if (!pend) ++w;
            }
        }
    }
    __syncthreads();
    const uint64_t rand_ii = a.rand_base + s_rbase + rrank - ((LPP == 2 && j == 1 && is_rand) ? 1u : 0u);   // odd lane: its even partner was counted
    PH_MARK(2);     // error tests + substitutions
    // ---- name fields of the pair (dwgsim.c:923-929): both ends print both ends' numbers ----
    int32_t e0 = n_err, u0 = rr.n_sub, i0 = rr.n_indel, x0 = rr.ext_coor;     // read end 1
    int32_t e1c = 0, u1 = 0, i1 = 0, x1 = 0;                                   // read end 2 (single-end: zeros, dwgsim.c:643)
    if (LPP == 2) {
        const int32_t o0 = __shfl_xor(n_err, 1), o1 = __shfl_xor(rr.n_sub, 1), o2 = __shfl_xor(rr.n_indel, 1), o3 = __shfl_xor(rr.ext_coor, 1);
        if (j == 0) { e1c = o0; u1 = o1; i1 = o2; x1 = o3; }
        else { e1c = n_err; u1 = rr.n_sub; i1 = rr.n_indel; x1 = rr.ext_coor; e0 = o0; u0 = o1; i0 = o2; x0 = o3; }
    }
    // SOLiD: the BWA files print the counts "minus the first colour" (dwgsim.c:945-946); only n_err and n_indel can differ
    int32_t e0w = e0, i0w = i0, e1w = e1c, i1w = i1;
    if (DT == 1) {
        int32_t f0 = err_first, g0 = rr.n_ins, f1 = 0, g1 = 0;
        if (LPP == 2) {
            const int32_t of = __shfl_xor(err_first, 1), og = __shfl_xor(rr.n_ins, 1);
            if (j == 0) { f1 = of; g1 = og; } else { f1 = err_first; g1 = rr.n_ins; f0 = of; g0 = og; }
        }
        e0w = e0 - f0; i0w = i0 - g0; e1w = e1c - f1; i1w = i1 - g1;
    }
    const NameCounts nc{e0, u0, i0, e1c, u1, i1}, ncw{e0w, u0, i0w, e1w, u1, i1w};
    uint32_t tail_len, tail_len_w, fixed_len;
    if (is_rand) { fixed_len = (uint32_t)a.rand_fixed_len; tail_len = tail_len_w = 25u + ndigits16(rand_ii); }   // "_0_0_0_0_1_1_0:0:0_0:0:0_" (25 chars) + hex
    else {
        fixed_len = (uint32_t)a.name_fixed_len;
        tail_len = pair_tail_len(x0, x1, nc, ii);
        tail_len_w = (DT == 1) ? pair_tail_len(x0, x1, ncw, ii) : tail_len;
    }
    const bool emits = valid && s_out > 0;
    // record lengths.  SOLiD: BWA drops the first colour and its quality (dwgsim.c:950-955); BFAST prepends the adaptor 'A' (:968-975)
    const uint32_t Lbwa = !emits ? 0u : (DT == 1) ? (1u + fixed_len + tail_len_w + 2u + 1u + 2u * (uint32_t)(s_out - 1) + 3u + 1u)
                                                  : (1u + fixed_len + tail_len + 2u + 1u + (uint32_t)s_out + 3u + (uint32_t)s_out + 1u);

    // ---- record offsets: block scan + decoupled look-back over logical blocks ----
    uint32_t e1, e2, eb = 0, T1, T2, Tb = 0;
    if (DT == 1) {
        const uint32_t v[3] = {j == 0 ? Lbwa : 0u, j == 1 ? Lbwa : 0u, emits ? (1u + fixed_len + tail_len + 1u + 1u + 2u * (uint32_t)s_out + 3u + 1u) : 0u};
        uint32_t ex[3], tot[3]; block_excl_scan_n<3>(v, sm_bytes, ex, tot);
        e1 = ex[0]; e2 = ex[1]; eb = ex[2]; T1 = tot[0]; T2 = tot[1]; Tb = tot[2];
    } else {
        const uint32_t v[2] = {j == 0 ? Lbwa : 0u, j == 1 ? Lbwa : 0u};
        uint32_t ex[2], tot[2]; block_excl_scan_n<2>(v, sm_bytes, ex, tot);
        e1 = ex[0]; e2 = ex[1]; T1 = tot[0]; T2 = tot[1];
    }
    if (wave == 0) {
        const uint64_t g = lookback_excl(a.status[0], t, T1, 0); if (lane == 0) s_base[0] = g;
        if (DT == 1) { const uint64_t gb = lookback_excl(a.status[3], t, Tb, 0); if (lane == 0) s_base[2] = gb; }
    }
    if (wave == (nwaves > 1 ? 1 : 0)) { const uint64_t g = lookback_excl(a.status[1], t, T2, 0); if (lane == 0) s_base[1] = g; }
    __syncthreads();
    const uint64_t G1 = s_base[0], G2 = s_base[1];
    const uint64_t reads_before_block = (uint64_t)t * PPB * (uint64_t)LPP;
    const uint64_t nvalid_before = (uint64_t)tid;                  // valid lanes form a prefix of the block
    const uint64_t off_bwa = (j == 0) ? G1 + e1 : G2 + e2;
    // Illumina / Ion Torrent: a BFAST record is its BWA record minus the 2-byte "/1" suffix, so its offset follows from the two BWA scans
    const uint64_t off_bf = (DT == 1) ? s_base[2] + eb : G1 + G2 - 2 * reads_before_block + e1 + e2 - 2 * nvalid_before;
    if (tid == nthr - 1) {
        const uint64_t nblocks = (a.n_pairs + PPB - 1) / PPB;
        if ((uint64_t)t + 1 == nblocks) {
            const uint64_t nreads = a.n_pairs * (uint64_t)((a.p.len[1] > 0) ? 2 : 1);
            a.counters[4] = a.p.has_bwa ? G1 + T1 : 0;
            a.counters[5] = a.p.has_bwa ? G2 + T2 : 0;
            a.counters[6] = !a.p.has_bfast ? 0 : (DT == 1) ? s_base[2] + Tb : G1 + T1 + G2 + T2 - 2 * nreads;
        }
    }

    PH_MARK(3);     // name lengths, block scan, look-back
    // ---- SOLiD records (dwgsim.c:934-976, :1056-1094): the two outputs differ in name counts, suffix, alphabet and length ----
    if (DT == 1) {
        if (emits) {
            for (int which = 0; which < 2; ++which) {            // 0: BWA stream of this end, 1: BFAST
                if (!(OUT & (1 << which))) continue;
                Out2<1> o;
                o.a.init(which ? a.out[2] + off_bf : (j ? a.out[1] : a.out[0]) + off_bwa);
                put_name_fixed(o, is_rand ? s_fixed[1] : s_fixed[0], is_rand ? a.rand_fixed : a.name_fixed, fixed_len);
                if (is_rand) put_rand_tail(o, rand_ii);
                else put_pair_tail(o, x0, x1, pd.strand0, pd.strand1,           // (value selects: a struct select would go through memory)
                                   NameCounts{which ? e0 : e0w, u0, which ? i0 : i0w, which ? e1c : e1w, u1, which ? i1 : i1w}, ii);
                if (which == 0) o.putn((uint64_t)'/' | ((uint64_t)('2' - j) << 8) | ((uint64_t)'\n' << 16), 3);      // F3 is annotated "/2", R3 "/1" (dwgsim.c:938-939)
                else { o.put('\n'); o.put('A'); }
                const int first = which ? 0 : 1;                // BWA skips the first colour and its quality
                for (int w = 0; w * 8 < s_out; ++w) {
                    const uint32_t word = lds[w * nthr];
                    const uint32_t c0 = which ? colour_digits4(word) : base_chars4(word), c1 = which ? colour_digits4(word >> 16) : base_chars4(word >> 16);
                    const int lo = w == 0 ? first : 0, hi = s_out - w * 8 < 8 ? s_out - w * 8 : 8;
                    if (lo == 0 && hi == 8) { o.put4(c0); o.put4(c1); }
                    else for (int b = lo; b < hi; ++b) o.put(((b < 4 ? c0 : c1) >> (8 * (b & 3))) & 0xff);
                }
                o.put('\n'); o.put('+'); o.put('\n');
                for_each_quality(a.p, key, D_QUAL0 + (uint32_t)j, ii, att, j ? a.qbase[1] : a.qbase[0], s, s_out,
                                 [&](int i, uint32_t q) { if (i >= first) o.put(q); });       // same draws for both outputs
                o.put('\n');
                o.flush();
            }
        }
    } else
    // ---- write the record(s) ----
    if (valid && s_out > 0) {
        Out2<OUT> o;
        if (OUT & 1) o.a.init((j ? a.out[1] : a.out[0]) + off_bwa);
        if (OUT & 2) o.b.init(a.out[2] + off_bf);
        put_name_fixed(o, is_rand ? s_fixed[1] : s_fixed[0], is_rand ? a.rand_fixed : a.name_fixed, fixed_len);
        if (is_rand) put_rand_tail(o, rand_ii);
        else put_pair_tail(o, x0, x1, pd.strand0, pd.strand1, nc, ii);
        if (OUT & 1) o.a.putn((uint64_t)'/' | ((uint64_t)('1' + j) << 8) | ((uint64_t)'\n' << 16), 3);
        if (OUT & 2) o.b.put('\n');
        PH_MARK(4); // header line
        // bases
        PackReader<4> rev; rev.init(lds, nthr);
        for (int w = 0; w * 8 < s_out; ++w) {
            uint32_t word;
            if (DT == 2 && flow_reversed) {         // base i of the record = base s_out-1-i of the flow-model orientation
                word = 0;
                for (int b = 0; b < 8; ++b) { const int i = w * 8 + b; if (i < s_out) word |= rev.get(s_out - 1 - i) << (4 * b); }
            } else word = lds[w * nthr];
            const int rem = s_out - w * 8;
            if (rem >= 8) {
                o.put4(base_chars4(word)); o.put4(base_chars4(word >> 16));
            } else {
                const uint32_t c0 = base_chars4(word), c1 = base_chars4(word >> 16);
                for (int b = 0; b < rem; ++b) o.put(((b < 4 ? c0 : c1) >> (8 * (b & 3))) & 0xff);
            }
        }
        o.put('\n'); o.put('+'); o.put('\n');
        PH_MARK(5); // sequence line
        // qualities (dwgsim.c:899-918), four characters per store
        {
            uint32_t qacc = 0, nq = 0;
            for_each_quality(a.p, key, D_QUAL0 + (uint32_t)j, ii, att, j ? a.qbase[1] : a.qbase[0], s, s_out, [&](int, uint32_t q) {
                qacc |= q << (8 * nq);
                if (++nq == 4) { o.put4(qacc); qacc = 0; nq = 0; }
            });
            for (uint32_t q = 0; q < nq; ++q) o.put((qacc >> (8 * q)) & 0xff);
        }
        o.put('\n');
        o.flush();
    }
    PH_MARK(6);     // quality line
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (declared in dw_launch.hpp)
// ------------------------------------------------------------------------------------------------
[[maybe_unused]] static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

#if DW_HAS(0)

void launch_pack(hipStream_t st, const uint8_t *ascii, uint8_t *ref, uint8_t *h0, uint8_t *h1, int64_t l)
{
    const uint64_t nchunk = (uint64_t)(l + 15) >> 4;
    uint32_t nb = cdiv(nchunk, 256); if (nb > (1u << 16)) nb = 1u << 16; if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_pack, dim3(nb), dim3(256), 0, st, ascii, ref, h0, h1, l);
}
void launch_site_scan(hipStream_t st, const uint8_t *ref, int64_t l, WalkParams wp, uint32_t contig_index, uint16_t *mask, uint32_t *block_count)
{
    hipLaunchKernelGGL(k_site_scan, dim3(cdiv((uint64_t)l, SCAN_POS_PER_BLOCK)), dim3(SCAN_THREADS), 0, st, ref, l, wp, contig_index, mask, block_count);
}
void launch_scan_excl(hipStream_t st, uint32_t *data, uint32_t n, uint64_t *total_out)
{
    hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, st, data, n, total_out);
}
void launch_compact(hipStream_t st, const uint16_t *mask, const uint32_t *block_base, int32_t *out, int64_t l)
{
    hipLaunchKernelGGL(k_compact, dim3(cdiv((uint64_t)l, SCAN_POS_PER_BLOCK)), dim3(SCAN_THREADS), 0, st, mask, block_base, out);
}
void launch_events(hipStream_t st, const int32_t *cand, uint32_t n, const uint8_t *ref, int64_t l, WalkParams wp, uint32_t contig_index, Event *ev, uint32_t *max_del)
{
    if (n) hipLaunchKernelGGL(k_events, dim3(cdiv(n, 256)), dim3(256), 0, st, cand, n, ref, l, wp, contig_index, ev, max_del);
}
void launch_resolve(hipStream_t st, Event *ev, uint32_t n, const uint32_t *max_del, uint4 *flags, uint32_t *tot4)
{
    if (n) hipLaunchKernelGGL(k_resolve, dim3(cdiv(n, 256)), dim3(256), 0, st, ev, n, max_del, flags);
    hipLaunchKernelGGL(k_scan4, dim3(1), dim3(1024), 0, st, flags, n, tot4);
}
void launch_apply(hipStream_t st, Event *ev, uint32_t n, const uint4 *flags, ContigDev c, WalkParams wp)
{
    if (n) hipLaunchKernelGGL(k_apply, dim3(cdiv(n, 256)), dim3(256), 0, st, ev, n, flags, c, wp);
}
void launch_justify_seq(hipStream_t st, const Event *ev, uint32_t n, ContigDev c)
{
    if (n) hipLaunchKernelGGL(k_justify_seq, dim3(1), dim3(64), 0, st, ev, n, c);
}
void launch_justify(hipStream_t st, const Event *ev, uint32_t n, ContigDev c, int32_t *lo, int32_t *sufmin, uint8_t *bound)
{
    if (!n) return;
    hipLaunchKernelGGL(k_jreach, dim3(cdiv(n, 256)), dim3(256), 0, st, ev, n, c, lo);
    hipLaunchKernelGGL(k_sufmin, dim3(1), dim3(1024), 0, st, lo, n, sufmin);
    hipLaunchKernelGGL(k_jbound, dim3(cdiv(n, 256)), dim3(256), 0, st, ev, n, c, sufmin, bound);
    hipLaunchKernelGGL(k_jrun, dim3(cdiv(n, 64)), dim3(64), 0, st, ev, n, c, bound);
}
void launch_apply_patches(hipStream_t st, const int32_t *pos, const uint16_t *cells, uint32_t n, uint8_t *h0, uint8_t *h1)
{
    if (n) hipLaunchKernelGGL(k_apply_patches, dim3(cdiv(n, 256)), dim3(256), 0, st, pos, cells, n, h0, h1);
}
void launch_collect_mask(hipStream_t st, const uint8_t *h0, const uint8_t *h1, int64_t l, uint16_t *mask, uint32_t *block_count)
{
    hipLaunchKernelGGL(k_collect_mask, dim3(cdiv((uint64_t)l, SCAN_POS_PER_BLOCK)), dim3(SCAN_THREADS), 0, st, h0, h1, l, mask, block_count);
}
void launch_gather(hipStream_t st, const int32_t *pos, uint32_t n, const uint8_t *h0, const uint8_t *h1, uint16_t *cells)
{
    if (n) hipLaunchKernelGGL(k_gather, dim3(cdiv(n, 256)), dim3(256), 0, st, pos, n, h0, h1, cells);
}
// Self-test of the range-restricted fp64 forms (dw_common.hpp) against the compiler's own `/`, sqrt() and the general det_log:
// operands drawn exactly as the quality path draws them, plus mantissa x exponent pairs over [2^-70, 2^70].  mism[0..2] count
// bitwise differences of div_mid, sqrt_mid, det_log<true>; mism[3] counts the comparisons made.
__global__ void __launch_bounds__(256) k_selftest_fp64(uint32_t seed, uint64_t n, uint64_t *mism)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t bad_div = 0, bad_sqrt = 0, bad_log = 0, done = 0;
    if (i < n) {
        const RngKey key{seed, 0u};
        const U4 b = rng_block(key, 31, i, 0, 0, 0), c = rng_block(key, 31, i, 0, 0, 1);
        const double a1 = (double)b.x * 0x1p-31 - 1.0, a2 = (double)b.y * 0x1p-31 - 1.0, r = a1 * a1 + a2 * a2;
        if (r < 1.0 && r != 0.0) {
            const double l1 = det_log(r), l2 = det_log<true>(r);
            bad_log += dbl_bits(l1) != dbl_bits(l2);
            const double x = -2.0 * l1, q1 = x / r, q2 = div_mid(x, r);
            bad_div += dbl_bits(q1) != dbl_bits(q2);
            bad_sqrt += dbl_bits(sqrt(q1)) != dbl_bits(sqrt_mid(q1));
            const double f = bits_dbl((dbl_bits(r) & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull) - 1.0;       // a det_log-style f / (2 + f)
            bad_div += dbl_bits(f / (2.0 + f)) != dbl_bits(div_mid(f, 2.0 + f));
            done += 4;
        }
        const double x = ldexp(1.0 + u53(c.x, c.y), (int)(b.z % 141u) - 70), y = ldexp(1.0 + u53(c.z, c.w), (int)(b.w % 141u) - 70);
        bad_div += dbl_bits(x / y) != dbl_bits(div_mid(x, y));
        bad_sqrt += dbl_bits(sqrt(x)) != dbl_bits(sqrt_mid(x));
        done += 2;
    }
    const uint32_t s0 = wave_sum_u32(bad_div), s1 = wave_sum_u32(bad_sqrt), s2 = wave_sum_u32(bad_log), s3 = wave_sum_u32(done);
    if ((threadIdx.x & 63) == 0) {
        if (s0) atomicAdd((unsigned long long *)&mism[0], (unsigned long long)s0);
        if (s1) atomicAdd((unsigned long long *)&mism[1], (unsigned long long)s1);
        if (s2) atomicAdd((unsigned long long *)&mism[2], (unsigned long long)s2);
        atomicAdd((unsigned long long *)&mism[3], (unsigned long long)s3);
    }
}
void launch_selftest_fp64(hipStream_t st, uint32_t seed, uint64_t n, uint64_t *mism)
{
    hipLaunchKernelGGL(k_selftest_fp64, dim3(cdiv(n, 256)), dim3(256), 0, st, seed, n, mism);
}
void launch_summarize(hipStream_t st, const uint8_t *cells, int64_t l, uint16_t *summ)
{
    const uint64_t nb = (uint64_t)(l + SUMM_CELLS - 1) / SUMM_CELLS;
    if (nb) hipLaunchKernelGGL(k_summarize, dim3(cdiv(nb, 256)), dim3(256), 0, st, cells, l, summ);
}
void launch_place(hipStream_t st, const SimArgs &a)
{
    if (a.p.len[1] > 0) hipLaunchKernelGGL(k_place<2>, dim3(cdiv(a.n_pairs, PAIRS_PER_BLOCK)), dim3(PAIRS_PER_BLOCK * 2), 0, st, a);
    else hipLaunchKernelGGL(k_place<1>, dim3(cdiv(a.n_pairs, PAIRS_PER_BLOCK)), dim3(PAIRS_PER_BLOCK), 0, st, a);
}
// one launcher per (LPP, DT) family, each defined in its own part
void launch_sim_2_0(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_1_0(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_2_2(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_1_2(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_2_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_1_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_simulate(hipStream_t st, const SimArgs &a)
{
    const bool pe = a.p.len[1] > 0, ion = a.p.data_type == 2;
    const uint32_t nb = cdiv(a.n_pairs, SIM_THREADS / (pe ? 2 : 1));
    const int out = (a.p.has_bwa ? 1 : 0) | (a.p.has_bfast ? 2 : 0);
    const uint32_t nthr = SIM_THREADS;
    const size_t lds = (size_t)(ion ? 4 : a.lds_words) * nthr * 4;   // Ion Torrent: only the pass-2 run stack (8 runs); its read buffers are in a.flow_scratch
    const bool solid = a.p.data_type == 1;
    if (pe) { if (ion) launch_sim_2_2(st, a, nb, lds, out); else if (solid) launch_sim_2_1(st, a, nb, lds, out); else launch_sim_2_0(st, a, nb, lds, out); }
    else { if (ion) launch_sim_1_2(st, a, nb, lds, out); else if (solid) launch_sim_1_1(st, a, nb, lds, out); else launch_sim_1_0(st, a, nb, lds, out); }
}
#endif // DW_HAS(0): launchers

#define DW_SIM_FAMILY(LPP, DT)                                                                                   \
    void launch_sim_##LPP##_##DT(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out)             \
    {                                                                                                            \
        const uint32_t nthr = SIM_THREADS;                                                                       \
        if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, DT>), dim3(nb), dim3(nthr), lds, st, a);            \
        else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, DT>), dim3(nb), dim3(nthr), lds, st, a);       \
        else hipLaunchKernelGGL((k_simulate<LPP, 3, DT>), dim3(nb), dim3(nthr), lds, st, a);                     \
    }
#if DW_HAS(1)
DW_SIM_FAMILY(2, 0)
#endif
#if DW_HAS(2)
DW_SIM_FAMILY(1, 0)
#endif
#if DW_HAS(3)
DW_SIM_FAMILY(2, 2)
#endif
#if DW_HAS(4)
DW_SIM_FAMILY(1, 2)

// -B (dwgsim_opt.c:415-457): lane = one random read of read end a.end pushed through the flow model on the forward strand; the block
// adds its error and length sums to counters[8], [9].  Draws: bases = narrow words of (D_CALIB + end, read, attempt 0), flow model =
// the sequential narrow stream of (D_CALIB + end, read, attempt 1).
__global__ void __launch_bounds__(PAIRS_PER_BLOCK) k_calibrate(CalibArgs a)
{
    DW_DYN_SHARED(uint32_t, dyn_lds);
    __shared__ uint8_t s_flow[64];
    const int tid = (int)threadIdx.x, nthr = PAIRS_PER_BLOCK;
    if (tid < 64) s_flow[tid] = a.flow[tid];
    __syncthreads();
    const uint64_t jj = (uint64_t)blockIdx.x * PAIRS_PER_BLOCK + (uint64_t)tid;
    uint32_t *buf = a.scratch + (size_t)blockIdx.x * ((size_t)(a.lds_words + ((a.cap + 15) >> 4)) * nthr) + tid;
    int32_t n_err = 0; int s_out = 0;
    if (jj < a.n_reads) {
        const RngKey key{a.seed, 0u};
        const uint32_t dom = D_CALIB + (uint32_t)a.end;
        for (int w = 0; w * 8 < a.len; ++w) {
            const U4 q0 = rng_block(key, dom, jj, 0, 0, (uint32_t)(2 * w)), q1 = rng_block(key, dom, jj, 0, 0, (uint32_t)(2 * w + 1));
            const uint32_t rw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) if (w * 8 + b < a.len) word |= (rw[b] >> 30) << (4 * b);      // (int)(u * 4.0) & 3
            buf[w * nthr] = word;
        }
        FlowRng rg; rg.seed = a.seed; rg.contig = 0; rg.dom = dom; rg.att = 1; rg.slot = 0; rg.ii = jj; rg.w0 = rg.w1 = rg.w2 = rg.w3 = 0;
        s_out = flow_errors(rg, s_flow, a.flow_len, a.thr, buf, buf + (size_t)a.lds_words * nthr, dyn_lds + tid, nthr, a.len, 0, a.cap, &n_err);
        if (s_out < 0) { atomicOr((unsigned long long *)&a.counters[2], 2ull); s_out = 0; n_err = 0; }
    }
    const uint32_t es = wave_sum_u32((uint32_t)n_err), ls = wave_sum_u32((uint32_t)s_out);
    if ((tid & 63) == 0) { atomicAdd((unsigned long long *)&a.counters[8], (unsigned long long)es); atomicAdd((unsigned long long *)&a.counters[9], (unsigned long long)ls); }
}
void launch_calibrate(hipStream_t st, const CalibArgs &a)
{
    hipLaunchKernelGGL(k_calibrate, dim3(cdiv(a.n_reads, PAIRS_PER_BLOCK)), dim3(PAIRS_PER_BLOCK), (size_t)4 * PAIRS_PER_BLOCK * 4, st, a);
}
#endif
#if DW_HAS(5)
DW_SIM_FAMILY(2, 1)
#endif
#if DW_HAS(6)
DW_SIM_FAMILY(1, 1)
#endif

} // namespace dw
