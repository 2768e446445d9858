// dw_walk.hip -- hand-written HIP kernels (gfx950 / MI355X) of the mutation walk
// (replaces src/mut.c:591-643 mut_diref + :481-589 mut_left_justify + the cell list behind :781-893 mut_print).
//
//     k_pack          ASCII -> base codes, initialises both haplotypes            HBM: 1 B in, 3 B out / base
//     k_site_scan_slots / k_slot_scan / k_slot_gather (or k_site_scan_list)   candidate sites: a gap chain per window of 256 positions -> ordered list     HBM: a byte per candidate
//     k_scan_excl     single-block exclusive scan of per-block counts
//     k_compact       ordered compaction of a bitmask into a position list
//     k_events        one thread per candidate: speculative event (type, ploidy, lengths)
//     k_resolve       liveness of candidates (deletion runs swallow later candidates)
//     k_apply         writes live events into the haplotype cells + insertion tables
//     k_jreach / k_sufmin / k_jbound / k_jrun   left-justification of indels: exact sequential
//                     semantics inside independent clusters, clusters in parallel
//     k_apply_patches file-driven mutations (-m / -b / -v) resolved on the host, scattered here
//     k_collect_mask / k_gather   list of mutated cells for the host's txt/vcf writer
//     k_mut_debug     the reference's consistency asserts (mut.c:379-425)
//     k_make_view     4-bit read view of a finished haplotype (what base extraction reads) + its two levels of summaries (what the random-read count reads)
//
// Byte/integer work only: no MFMA.  Host-callable launchers (dw_launch.hpp) are at the end.
#include "dw_device.hpp"
#include "dw_launch.hpp"
#include <stdlib.h>

namespace dw {

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
// K1: candidate sites.  mut.c:618 `c < 4 && drand48() < opt->mut_rate`, once per ACGT position.  Rounds 2-5 drew a 16-bit uniform per position (a Philox
// block per eight): 386 M blocks per 3.09 Gb genome, ALU-bound at 1 G positions per ms -- 2.1 of the 5.5 ms of a whole-genome walk, on every device that
// walks it -- to learn "no" 999 times in 1000.  The candidates of a contig are a Bernoulli(r') process over its positions; restricted to a WINDOW of 256
// positions it is independent of every other window, so every window has a gap chain of its own (dw_common.hpp D_WALK_SITE, geom_gap): S_0 = G_0, S_(m+1) =
// S_m + 1 + G_(m+1), candidates at 256 q + S_m while S_m < 256 -- one Philox block per four candidates.  A LANE takes a window: it walks its chain once to
// count the candidates that fall on A, C, G or T inside the contig (the one byte of the pristine 4-bit view each needs is all the sequence that is read),
// and once more to write them.
// ------------------------------------------------------------------------------------------------
constexpr int SITE_WINDOW = 256, SITE_THREADS = 256, SITE_BLOCK_POS = SITE_WINDOW * SITE_THREADS;      // positions per window (a lane) and per block (65 536)
static_assert(GROUP_ALIGN % SITE_WINDOW == 0, "a window lies inside one contig: contigs start at multiples of GROUP_ALIGN");
struct SiteChain {
    uint32_t m, w0, w1, w2, w3;
    DW_DEV uint32_t gap(RngKey key, uint32_t q, const WalkParams &wp, const uint32_t *lg)
    {
        if ((m & 3u) == 0u) { const U4 b = rng_block<false>(key, D_WALK_SITE, (uint64_t)q, 0, 0, m >> 2); w0 = b.x; w1 = b.y; w2 = b.z; w3 = b.w; }
        const uint32_t k = m & 3u; ++m;
        const uint32_t w = (k & 2u) ? ((k & 1u) ? w3 : w2) : ((k & 1u) ? w1 : w0);
        return wp.mut_thr >= 0x100000000ull ? 0u : geom_gap(w, lg, wp.gap_r, wp.gap_s);
    }
};
// the candidates of the window that starts at group coordinate g0 (a multiple of SITE_WINDOW): f(group coordinate) for each, in rising order; returns how many
template <class F>
DW_DEV uint32_t site_window(const uint8_t *__restrict__ refview, int64_t l_total, const SegTab &seg, const WalkParams &wp, const uint32_t *lg, int64_t g0, F &&f)
{
    if (g0 >= l_total || wp.mut_thr == 0) return 0;
    const uint32_t sk = seg_of(seg, g0);
    const int64_t p0 = g0 - seg.start[sk], l = seg.len[sk];      // position inside the contig: what the draws are indexed by
    if (p0 >= l) return 0;                                       // (the padding behind a contig)
    const RngKey key{wp.seed, seg.cindex[sk]};
    const uint32_t q = (uint32_t)(p0 >> 8);
    const uint32_t lim = l - p0 < SITE_WINDOW ? (uint32_t)(l - p0) : (uint32_t)SITE_WINDOW;
    SiteChain ch; ch.m = 0; ch.w0 = ch.w1 = ch.w2 = ch.w3 = 0;
    uint32_t n = 0;
    for (uint32_t S = ch.gap(key, q, wp, lg); S < lim; S += 1u + ch.gap(key, q, wp, lg)) {
        const int64_t g = g0 + S;
        const uint32_t nib = (refview[g >> 1] >> (4 * (g & 1))) & 15u;      // the pristine view: a nibble below 4 is A, C, G or T
        if (nib < 4u) { f(g, n); ++n; }
    }
    return n;
}

// one kernel with a decoupled look-back (re-runs with exact capacities, mutation rates at which slots would be as large as the list): the block's place in
// the ordered list from the blocks in front (logical block ids from a ticket, so that every predecessor has started).  status: one word per block, ticket:
// one word, both zeroed by the host; total -> *n_out.
__global__ void __launch_bounds__(SITE_THREADS) k_site_scan_list(const uint8_t *__restrict__ refview, int64_t l_total, SegTab seg, WalkParams wp, uint64_t *status, uint64_t *ticket,
                                                              int32_t *__restrict__ out, uint32_t cap, uint64_t *n_out, uint32_t n_blocks)
{
    __shared__ uint32_t sm[1][16], s_lg[FLOW_LG_ENTRIES];
    __shared__ uint32_t s_t; __shared__ uint64_t s_base;
    for (int q = (int)threadIdx.x; q < FLOW_LG_ENTRIES; q += SITE_THREADS) s_lg[q] = wp.lg[q];
    if (threadIdx.x == 0) s_t = (uint32_t)atomicAdd((unsigned long long *)ticket, 1ull);
    __syncthreads();
    const uint32_t t = uniform_u32(s_t);
    const int64_t g0 = ((int64_t)t * SITE_THREADS + threadIdx.x) * SITE_WINDOW;
    const uint32_t cnt[1] = {site_window(refview, l_total, seg, wp, s_lg, g0, [](int64_t, uint32_t) {})};
    uint32_t off[1], tot[1];
    block_excl_scan_n<1>(cnt, sm, off, tot);
    if (threadIdx.x < 64) {
        const uint64_t g = lookback_excl(status, t, tot[0], 0);
        if (threadIdx.x == 0) { s_base = g; if (t + 1 == n_blocks) *n_out = g + tot[0]; }
    }
    __syncthreads();
    const uint64_t at0 = s_base + off[0];
    if (cnt[0]) (void)site_window(refview, l_total, seg, wp, s_lg, g0, [&](int64_t g, uint32_t k) { if (at0 + k < cap) out[at0 + k] = (int32_t)g; });      // (past the capacity: the host re-runs)
}

// K1 without ANY dependency between blocks (round 6): a block writes its candidates into a SLOT of its own (slot_cap entries: mean + 8 sigma + 32 of the
// Binomial(65 536, r)) with its count beside; one small block then scans the counts (k_slot_scan) and k_slot_gather moves the slots' entries to their places
// in the ordered list -- 12 bytes per candidate, a thousandth of the positions.  A block that outgrows its slot (never, at mean + 8 sigma; forced in the
// tests) raises a flag and the host runs the walk again through the look-back form.
__global__ void __launch_bounds__(SITE_THREADS) k_site_scan_slots(const uint8_t *__restrict__ refview, int64_t l_total, SegTab seg, WalkParams wp, int32_t *__restrict__ slots, uint32_t slot_cap,
                                                               uint32_t *__restrict__ slot_cnt)
{
    __shared__ uint32_t sm[1][16], s_lg[FLOW_LG_ENTRIES];
    for (int q = (int)threadIdx.x; q < FLOW_LG_ENTRIES; q += SITE_THREADS) s_lg[q] = wp.lg[q];
    __syncthreads();
    const uint32_t t = blockIdx.x;
    const int64_t g0 = ((int64_t)t * SITE_THREADS + threadIdx.x) * SITE_WINDOW;
    const uint32_t cnt[1] = {site_window(refview, l_total, seg, wp, s_lg, g0, [](int64_t, uint32_t) {})};
    uint32_t off[1], tot[1];
    block_excl_scan_n<1>(cnt, sm, off, tot);
    if (threadIdx.x == 0) slot_cnt[t] = tot[0];
    int32_t *const slot = slots + (size_t)t * slot_cap;
    const uint32_t at0 = off[0];
    if (cnt[0]) (void)site_window(refview, l_total, seg, wp, s_lg, g0, [&](int64_t g, uint32_t k) { if (at0 + k < slot_cap) slot[at0 + k] = (int32_t)g; });      // (past the slot: the host re-runs)
}
// one block: exclusive scan of the nb slot counts -> slot_base[0 .. nb), the candidates in all -> *n_out; over[0] = 1 and over[1] = that total if a slot
// was outgrown (*n_out = 0 then: the kernels behind find nothing to do)
__global__ void __launch_bounds__(1024) k_slot_scan(const uint32_t *__restrict__ slot_cnt, uint32_t nb, uint32_t slot_cap, uint32_t *__restrict__ slot_base, uint64_t *n_out, uint32_t *over)
{
    __shared__ uint32_t sm[17];
    __shared__ uint32_t s_over;
    if (threadIdx.x == 0) s_over = 0;
    const uint32_t per = (nb + 1023u) / 1024u, a = threadIdx.x * per, b = a + per < nb ? a + per : nb;
    uint32_t sum = 0; bool big = false;
    for (uint32_t q = a; q < b; ++q) { const uint32_t c = slot_cnt[q]; sum += c; big |= c > slot_cap; }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, sm, &total);      // (its first barrier also publishes s_over = 0)
    if (big) atomicOr(&s_over, 1u);
    for (uint32_t q = a; q < b; ++q) { slot_base[q] = run; run += slot_cnt[q]; }
    __syncthreads();
    if (threadIdx.x == 0) { const bool o = s_over != 0; *n_out = o ? 0ull : (uint64_t)total; if (o) { over[0] = 1u; over[1] = total; } }
}
// a block per slot: its entries to their places in the ordered list
__global__ void __launch_bounds__(256) k_slot_gather(const int32_t *__restrict__ slots, uint32_t slot_cap, const uint32_t *__restrict__ slot_cnt, const uint32_t *__restrict__ slot_base,
                                                     int32_t *__restrict__ out, uint32_t cap)
{
    const uint32_t t = blockIdx.x, base = slot_base[t];
    uint32_t n = slot_cnt[t]; if (n > slot_cap) n = slot_cap;
    for (uint32_t k = threadIdx.x; k < n; k += 256u) if (base + k < cap) out[base + k] = slots[(size_t)t * slot_cap + k];
}

// ---- the walk as SPARSE work (round 5).  A walk touches about one cell in a thousand; rounds 1-4 nevertheless rewrote dense arrays for every walk (two
// resets of 1 byte per base, k_make_view over 3 bytes per base: 2.5 of the ~8 ms of kernels per genome).  Now the read views and the haplotype
// summaries of the UNMUTATED group are made once, at upload (refview / refsumm / refsumm2 stay as pristine copies), and a walk records the
// 64-cell chunks it may have written in a bitmap (one bit per chunk): the footprint of every live event, from the lower end of its
// left-justification scan (k_jreach's `lo`, a proven lower bound of every cell the event's scan reads or writes) to its last cell.  Two passes
// over the bitmap do the rest: AFTER the walk the views and summaries of the dirty chunks are recomputed from the cells; BEFORE the next walk of
// the group the dirty chunks' cells, views and summaries are set back from the pristine copies. ----
__global__ void k_mark_dirty(const Event *__restrict__ ev, Count nc, const int32_t *__restrict__ lo, uint32_t *__restrict__ dirty)
{
    const uint32_t n = count_of(nc), k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Event e = ev[k];
    if (!e.live) return;
    const int64_t right = (int64_t)e.pos + (e.type == 2 ? (int64_t)e.len - 1 : 0);
    const int64_t a = (int64_t)(lo[k] < e.pos ? lo[k] : e.pos) >> 6, b = right >> 6;
    for (int64_t q = a; q <= b; ++q) atomicOr(&dirty[q >> 5], 1u << (q & 31));
}
// one LANE per chunk; a word of the bitmap = 32 chunks = 2048 cells = two words of the coarse summaries is half a wave, and a coarse word's sixteen fine
// words are summed across its sixteen lanes.  (One thread per bitmap word -- 31 k threads for a 64 Mb contig, each walking its word's chunks in turn --
// took 97 us of a 0.3 ms walk.)  RESTORE = false: views + summaries from the cells; true: cells, views and summaries from the pristine copies.
template <bool RESTORE, bool PER_CHUNK>
__global__ void __launch_bounds__(256) k_dirty_chunks(const uint32_t *__restrict__ dirty, uint32_t n_words, int64_t l_live, const uint8_t *__restrict__ ref, const uint8_t *__restrict__ refview,
                                                      const uint16_t *__restrict__ refsumm, const uint16_t *__restrict__ refsumm2,
                                                      uint8_t *__restrict__ cells0, uint8_t *__restrict__ cells1, uint8_t *__restrict__ view0, uint8_t *__restrict__ view1,
                                                      uint16_t *__restrict__ summ0, uint16_t *__restrict__ summ1, uint16_t *__restrict__ summ2_0, uint16_t *__restrict__ summ2_1)
{
    // PER_CHUNK: a lane per chunk (small groups: the kernel is latency, not throughput -- 97 -> 59 us for a 64 Mb contig); otherwise a thread per word of
    // the bitmap that takes the word's dirty chunks in turn (large groups: a wave's lanes all have work; a lane per chunk there runs the chunk code
    // for two lanes of a wave at a time: 0.43 -> 1.18 ms per 1.5 Gb group)
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t w = PER_CHUNK ? (uint32_t)(gid >> 5) : (uint32_t)gid;
    const bool in = w < n_words;
    const uint32_t touched = in ? dirty[w] : 0u;
    if (PER_CHUNK) { if (!__ballot(touched != 0u)) return; }    // (wave-uniform: neither of the wave's two words has a dirty chunk)
    else if (!touched) return;
    uint32_t bits = PER_CHUNK ? ((touched >> (gid & 31u)) & 1u) << (gid & 31u) : touched;      // the chunks of the word this thread takes
    uint32_t fine[2] = {0u, 0u};                               // (PER_CHUNK) this lane's chunk's fine summaries as they stand after this kernel
    const bool mine = bits != 0u;
    while (bits) {
        const int q = __ffs((int)bits) - 1; bits &= bits - 1;
        const int64_t ch = (int64_t)w * 32 + q, first = ch * SUMM_CELLS;
        if (RESTORE) {
#pragma unroll
            for (int k = 0; k < SUMM_CELLS / 16; ++k) { const uint4 v = *reinterpret_cast<const uint4 *>(ref + first + 16 * k); *reinterpret_cast<uint4 *>(cells0 + first + 16 * k) = v; *reinterpret_cast<uint4 *>(cells1 + first + 16 * k) = v; }
#pragma unroll
            for (int k = 0; k < SUMM_CELLS / 32; ++k) { const uint4 v = *reinterpret_cast<const uint4 *>(refview + (first >> 1) + 16 * k); *reinterpret_cast<uint4 *>(view0 + (first >> 1) + 16 * k) = v; *reinterpret_cast<uint4 *>(view1 + (first >> 1) + 16 * k) = v; }
            const uint16_t sv = refsumm[ch]; summ0[ch] = sv; summ1[ch] = sv;
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint8_t *cells = h ? cells1 : cells0; uint8_t *view = h ? view1 : view0;
                uint32_t indel = 0, non_acgt = 0;
#pragma unroll
                for (int half = 0; half < SUMM_CELLS / 32; ++half) {
                    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int64_t at = first + 32 * half + 16 * qq;
                        const uint4 v = *reinterpret_cast<const uint4 *>(cells + at);
                        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int b = 0; b < 16; ++b) {
                            const uint32_t c = (wd[b >> 2] >> (8 * (b & 3))) & 0xffu, ty = c & TMASK, base = c & 0xfu;
                            const uint32_t nib = ty == T_NONE ? (base < 4 ? base : base == 4 ? 8u : 9u) : (ty == T_SUB && base < 4) ? 4u + base : 15u;
                            const int cell = 16 * qq + b;
                            out[cell >> 3] |= nib << (4 * (cell & 7));
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int64_t rem = l_live - (at + 4 * k);               // cells of this word that belong to the group (the rest is padding)
                            const uint32_t lm = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : ((1u << (8 * (int)rem)) - 1u);
                            indel += (uint32_t)__popc(wd[k] & lm & 0x10101010u);
                            non_acgt |= wd[k] & lm & 0x0C0C0C0Cu;
                        }
                    }
                    *reinterpret_cast<uint4 *>(view + ((first + 32 * half) >> 1)) = make_uint4(out[0], out[1], out[2], out[3]);
                }
                fine[h] = indel | (non_acgt ? 0x8000u : 0u);
                (h ? summ1 : summ0)[ch] = (uint16_t)fine[h];
            }
        }
    }
    // the coarse summaries (SUMM2_CELLS cells = 16 chunks) that hold a dirty chunk: from their sixteen fine words
    static_assert(SUMM2_CELLS / SUMM_CELLS == 16, "a coarse summary = the sixteen lanes of half a bitmap word");
    if (!PER_CHUNK) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (!((touched >> (16 * half)) & 0xFFFFu)) continue;
            const int64_t s2 = (int64_t)w * 2 + half;
            if (s2 * SUMM2_CELLS >= l_live) continue;
            if (RESTORE) { const uint16_t sv = refsumm2[s2]; summ2_0[s2] = sv; summ2_1[s2] = sv; }
            else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint16_t *sm1 = h ? summ1 : summ0;
                    uint32_t cnt = 0, fl = 0;
                    for (int k = 0; k < SUMM2_CELLS / SUMM_CELLS; ++k) { const int64_t c2 = s2 * (SUMM2_CELLS / SUMM_CELLS) + k; if (c2 * SUMM_CELLS < l_live) { const uint32_t v = sm1[c2]; cnt += v & 0x7fffu; fl |= v & 0x8000u; } }
                    (h ? summ2_1 : summ2_0)[s2] = (uint16_t)(cnt | fl);
                }
            }
        }
        return;
    }
    // ... one fine word per lane of the half-word, summed across the sixteen lanes
    const int q = (int)(gid & 31u);
    const int64_t ch = (int64_t)w * 32 + q;
    const int64_t s2 = (int64_t)w * 2 + (q >> 4);
    const bool live2 = in && ((touched >> (16 * (q >> 4))) & 0xFFFFu) != 0u && s2 * SUMM2_CELLS < l_live;
    if (RESTORE) { if (live2 && (q & 15) == 0) { const uint16_t sv = refsumm2[s2]; summ2_0[s2] = sv; summ2_1[s2] = sv; } }
    else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t v = 0;
            if (live2 && ch * SUMM_CELLS < l_live) v = mine ? fine[h] : (uint32_t)(h ? summ1 : summ0)[ch];
            uint32_t cnt = v & 0x7fffu, fl = v & 0x8000u;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { cnt += (uint32_t)__shfl_xor((int)cnt, m); fl |= (uint32_t)__shfl_xor((int)fl, m); }      // (every lane of the wave shuffles)
            if (live2 && (q & 15) == 0) (h ? summ2_1 : summ2_0)[s2] = (uint16_t)(cnt | fl);
        }
    }
}

// single-block exclusive scan (in place) of n uint32; optional 64-bit total.  One barrier per chunk: the LDS scratch alternates
// between two areas and every thread carries the running total itself.
__global__ void k_scan_excl(uint32_t *data, uint32_t n, uint64_t *total_out)
{
    __shared__ uint32_t sm[2][1][16];
    uint64_t grand = 0; int buf = 0;
    for (uint32_t base = 0; base < n; base += blockDim.x, buf ^= 1) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v[1] = {i < n ? data[i] : 0u};
        uint32_t ex[1], total[1];
        block_excl_scan_n<1>(v, sm[buf], ex, total);
        if (i < n) data[i] = ex[0] + (uint32_t)grand;
        grand += total[0];
    }
    if (total_out && threadIdx.x == 0) *total_out = grand;
}

// ordered compaction: positions of set bits of `mask` (one uint16 per thread of the producing kernel)
__global__ void k_compact(const uint16_t *__restrict__ mask, const uint32_t *__restrict__ block_base, int32_t *__restrict__ out, uint32_t cap)
{
    __shared__ uint32_t sm[17];
    uint32_t bits = mask[(int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x];
    uint32_t total;
    uint32_t off = block_base[blockIdx.x] + block_excl_scan((uint32_t)__popc(bits), sm, &total);
    const int64_t p0 = ((int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_POS_PER_THREAD;
    while (bits) { const int b = __ffs((int)bits) - 1; bits &= bits - 1; if (off < cap) out[off] = (int32_t)(p0 + b); ++off; }   // (past the capacity: the host re-runs)
}

// ------------------------------------------------------------------------------------------------
// K2a: one thread per candidate site: the event it would be if it is live.  mut.c:619-640, 287-308.
// ------------------------------------------------------------------------------------------------
__global__ void k_events(const int32_t *__restrict__ cand, Count nc, const uint8_t *__restrict__ ref, SegTab seg,
                         WalkParams wp, Event *__restrict__ ev, uint32_t *__restrict__ max_del)
{
    const uint32_t n_cand = count_of(nc);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_cand) {
        const int64_t g = cand[k];
        const uint32_t sk = seg_of(seg, g);
        const RngKey key{wp.seed, seg.cindex[sk]};
        const int64_t p = g - seg.start[sk], l = seg.len[sk];          // position inside its contig
        const uint32_t c = ref[g];
        const U4 b1 = rng_block<false>(key, D_WALK, (uint64_t)p, 0, 0, 1);   // slots 2,3
        const U4 b2 = rng_block<false>(key, D_WALK, (uint64_t)p, 0, 0, 2);   // slots 4,5
        Event e; e.pos = (int32_t)g; e.live = 1; e.len = 1; e.base = (uint8_t)c; e.seg = sk;
        if (u_lo(b1) >= wp.indel_frac) {                 // substitution (mut.c:619-626)
            e.type = 1;
            e.base = (uint8_t)((c + (uint32_t)(uint64_t)(u_hi(b1) * 3.0 + 1)) & 3);
            e.hap = (wp.is_hap || u_lo(b2) < 0.333333) ? 3 : (u_hi(b2) < 0.5 ? 1 : 2);
        } else if (u_hi(b1) < 0.5) {                     // deletion (mut.c:628-636) + its run (mut.c:610-617)
            e.type = 2;
            e.hap = (wp.is_hap || u_lo(b2) < 0.3333333) ? 3 : (u_hi(b2) < 0.5 ? 1 : 2);
            uint32_t len = 1;
            for (int64_t q = p + 1; q < l; ++q) {
                if ((int64_t)len < wp.indel_min || rng_slot<false>(key, D_WALK, (uint64_t)q, 0, 0) < wp.indel_extend) ++len; else break;
            }
            e.len = len;
            atomicMax(max_del, len);
        } else {                                         // insertion (mut.c:637-639 -> :287-308)
            e.type = 3;
            uint64_t num = 0; uint32_t kk = 0;
            do { ++num; } while (num < 0xFFFFFFFFull && ((int64_t)num < wp.indel_min || rng_slot<false>(key, D_WALK_INSLEN, (uint64_t)p, 0, kk++) < wp.indel_extend));
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
__global__ void k_resolve(Event *ev, Count nc, const uint32_t *max_del_p, uint4 *flags)
{
    const uint32_t n_cand = count_of(nc);
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

// exclusive scan of the four insertion-allocation columns; totals -> tot[0..3].  Eight consecutive rows per thread (all loads in flight at
// once, a sequential scan in registers), then one block scan of the thread totals: one barrier per 8192 rows.  One block does it all for
// short candidate lists (seglen = 0); long ones (a chr20-sized contig has 64 k candidates, chr1 250 k) are cut into WALK_SEGS segments
// scanned by one block each, k_scan4_fix then adds the totals of the segments in front.
constexpr uint32_t WALK_SEGS = 32;
static uint32_t g_walk_seg_min_rows = 16384;            // capacity from which the segmented form is used (tests lower it: dwgsim_hip_debug_option("walk_seg_min"))
void walk_debug_seg_min(uint32_t rows) { g_walk_seg_min_rows = rows ? rows : 16384; }
__global__ void k_scan4(uint4 *flags, Count nc, uint32_t *tot, uint32_t seglen, uint4 *part)
{
    const uint32_t n = count_of(nc);
    const uint32_t lo = seglen ? blockIdx.x * seglen : 0u, hi = seglen ? (lo + seglen < n ? lo + seglen : n) : n;
    constexpr uint32_t ITEMS = 8;
    __shared__ uint32_t sm[2][4][16];
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0; int buf = 0;
    for (uint32_t base = lo; base < hi; base += blockDim.x * ITEMS, buf ^= 1) {
        const uint32_t i0 = base + threadIdx.x * ITEMS;
        uint4 v[ITEMS];
#pragma unroll
        for (uint32_t q = 0; q < ITEMS; ++q) v[q] = i0 + q < hi ? flags[i0 + q] : make_uint4(0, 0, 0, 0);
        uint32_t t[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t q = 0; q < ITEMS; ++q) {         // row -> its exclusive prefix inside the thread (the dead flag stays in bit 31 of x)
            const uint32_t dead = v[q].x & 0x80000000u, x = v[q].x & 0x7fffffffu, y = v[q].y, z = v[q].z, w = v[q].w;
            v[q] = make_uint4(t[0] | dead, t[1], t[2], t[3]);
            t[0] += x; t[1] += y; t[2] += z; t[3] += w;
        }
        uint32_t ex[4], total[4];
        block_excl_scan_n<4>(t, sm[buf], ex, total);
#pragma unroll
        for (uint32_t q = 0; q < ITEMS; ++q)
            if (i0 + q < hi) flags[i0 + q] = make_uint4(((v[q].x & 0x7fffffffu) + ex[0] + c0) | (v[q].x & 0x80000000u), v[q].y + ex[1] + c1, v[q].z + ex[2] + c2, v[q].w + ex[3] + c3);
        c0 += total[0]; c1 += total[1]; c2 += total[2]; c3 += total[3];
    }
    if (threadIdx.x == 0) { if (seglen) part[blockIdx.x] = make_uint4(c0, c1, c2, c3); else { tot[0] = c0; tot[1] = c1; tot[2] = c2; tot[3] = c3; } }
}
__global__ void k_scan4_fix(uint4 *flags, Count nc, uint32_t *tot, uint32_t seglen, const uint4 *part)
{
    const uint32_t n = count_of(nc), i = blockIdx.x * blockDim.x + threadIdx.x, s = i / seglen;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (uint32_t q = 0; q < WALK_SEGS; ++q) { const uint4 p = part[q]; if (q < s) { a0 += p.x; a1 += p.y; a2 += p.z; a3 += p.w; } t0 += p.x; t1 += p.y; t2 += p.z; t3 += p.w; }
    if (i == 0) { tot[0] = t0; tot[1] = t1; tot[2] = t2; tot[3] = t3; }
    if (i < n && s) { const uint4 f = flags[i]; flags[i] = make_uint4(((f.x & 0x7fffffffu) + a0) | (f.x & 0x80000000u), f.y + a1, f.z + a2, f.w + a3); }
}

// K3: write live events into the cells and the insertion tables.
__global__ void k_apply(Event *ev, Count nc, const uint4 *flags, ContigDev c, WalkParams wp)
{
    const uint32_t n_cand = count_of(nc);
    adopt_device_sizes(c);
    if (c.tot4 && (c.tot4[1] > c.cap_bases[0] || c.tot4[3] > c.cap_bases[1])) return;      // the inserted-base pools are too small: the host re-runs with larger ones
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
        const RngKey key{wp.seed, c.seg.cindex[e.seg]};
        const uint64_t pl = (uint64_t)(p - c.seg.start[e.seg]);                  // position inside the contig: index of the draws
        const uint32_t idx[2] = {f.x & 0x7fffffffu, f.z}, off[2] = {f.y, f.w};
        for (int h = 0; h < 2; ++h) if (e.hap & (1 << h)) {
            c.hap[h].cells[p] = T_INS | e.base;
            c.hap[h].ins_pos[idx[h]] = (int32_t)p;
            c.hap[h].ins_len[idx[h]] = e.len;
            c.hap[h].ins_off[idx[h]] = off[h];
        }
        for (uint32_t j = 0; j < e.len; ++j) {      // draw j lands at printed index len-1-j (mut.c:313-315, :347-365 read by :249-279)
            const uint8_t b = (uint8_t)(uint64_t)(rng_slot<false>(key, D_WALK_INSBASE, pl, 0, j) * 4.0);
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
// [lo, hi): the group coordinates of the contig the position belongs to (the reference's 0 and seq->l)
DW_DEV void justify_ins(HapDev &h, int64_t i, int64_t lo)          // mut.c:427-478
{
    const uint32_t idx = ins_find(h, i);
    const uint32_t n = h.ins_len[idx];
    uint8_t *P = h.ins_bases + h.ins_off[idx];
    int64_t j = i;
    while (j > lo && (h.cells[j - 1] & TMASK) == T_NONE && P[n - 1] == (h.cells[j - 1] & 3)) {
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
DW_DEV void justify_visit(ContigDev &c, int64_t i, int *prev_del, int64_t lo, int64_t hi)
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
            const int64_t dl = del_run(h0, i, hi);
            if (hi <= i + dl) return;
            if (i > lo) for (int64_t j = i - 1;; --j) {
                const uint8_t a = h0.cells[j], b = h1.cells[j];
                if ((a & TMASK) != T_INS && (b & TMASK) != T_INS && (a & TMASK) != T_DEL && (b & TMASK) != T_DEL
                    && (a & 3) == (h0.cells[j + dl] & 3) && (b & 3) == (h1.cells[j + dl] & 3)) { del_swap(h0, j, dl); del_swap(h1, j, dl); }
                else break;
                if (j == lo) break;
            }
        } else { prev_del[0] = prev_del[1] = 0; justify_ins(h0, i, lo); justify_ins(h1, i, lo); }
    } else {
        if ((c1 & TMASK) == T_SUB || (c2 & TMASK) == T_SUB) { prev_del[0] = prev_del[1] = 0; }
        else if ((c1 & TMASK) == T_DEL || (c2 & TMASK) == T_DEL) {
            const int x = ((c1 & TMASK) == T_DEL) ? 0 : 1;
            if (prev_del[x] == 1) return;
            prev_del[x] = 1;
            HapDev &h = c.hap[x];
            const int64_t dl = del_run(h, i, hi);
            if (hi <= i + dl) return;
            if (i > lo) for (int64_t j = i - 1;; --j) {
                const uint8_t a = h.cells[j];
                if ((a & TMASK) == T_NONE && (a & 3) == (h.cells[j + dl] & 3)) del_swap(h, j, dl); else break;
                if (j == lo) break;
            }
        } else if ((c1 & TMASK) == T_INS) { prev_del[0] = prev_del[1] = 0; justify_ins(h0, i, lo); }
        else { prev_del[0] = prev_del[1] = 0; justify_ins(h1, i, lo); }
    }
}
// sequential cross-check (DWGSIM_HIP_JUSTIFY=seq): one thread walks every live event of the contig
__global__ void k_justify_seq(const Event *ev, Count nc, ContigDev c)
{
    const uint32_t n_cand = count_of(nc);
    adopt_device_sizes(c);
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int prev_del[2] = {0, 0};
    int64_t last = -1; uint32_t cur = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < n_cand; ++k) {
        const Event e = ev[k];
        if (!e.live) continue;
        const int64_t lo = c.seg.start[e.seg], hi = lo + c.seg.len[e.seg];
        if (e.seg != cur) { cur = e.seg; prev_del[0] = prev_del[1] = 0; last = lo - 1; }      // every contig is justified on its own (mut.c:481-489)
        const int64_t p = e.pos, right = p + (e.type == 2 ? (int64_t)e.len - 1 : 0);
        if (prev_del[0] | prev_del[1])
            for (int64_t q = last + 1; q < p; ++q) if (c.ref[q] < 4) { prev_del[0] = prev_del[1] = 0; break; }
        for (int64_t i = p; i <= right; ++i) justify_visit(c, i, prev_del, lo, hi);
        last = right;
    }
}

// ---- parallel left-justification --------------------------------------------------------------
// The reference's pass is sequential, but an indel only interacts with what its leftward scan can
// touch.  (1) k_jreach: per live event, a conservative lower bound `lo` of every cell its scan can
// read or write: continue while the cell is mutated on either haplotype (pre-justify state) or the
// bases are periodic there (deletion: cell[j]&3 == cell[j+L]&3, insertion: the rotated copy keeps
// matching).  Shifts preserve (cell & 3) at every position, so the true scan never goes further.
// (2) k_sufmin: suffix minimum of lo.  (3) k_jbound: event b starts a new cluster iff no event >= b
// can reach the previous live event's footprint and an unmutated non-N position separates them
// (prev_del is then 0, mut.c:585-587).  (4) k_jrun: one thread per cluster replays the exact
// sequential semantics (justify_visit) over its events; clusters touch disjoint cells.
DW_DEV int64_t reach_del(const ContigDev &c, int h, int64_t p, int64_t lo, int64_t hi)
{
    // period = the run of DELETE cells the sequential pass would measure at p (adjacent runs merge,
    // mut.c:503 / :535 / :557) on haplotype h
    const int64_t L = del_run(c.hap[h], p, hi);
    const uint8_t *cl = c.hap[h].cells;
    int64_t j = p - 1;
    for (; j >= lo; --j) {
        const bool mutated = ((c.hap[0].cells[j] | c.hap[1].cells[j]) & TMASK) != 0;
        // the scan compares the bases the CELLS hold (a substituted cell L further right holds its new base, and the homozygous scan walks
        // through substituted cells: mut.c:515-516), not the reference's: with the reference here, a deletion that crosses a substitution
        // was given too short a reach, its cluster ran beside the one it reaches into, and the result depended on which thread came first
        if (!(mutated || (p + L < hi && (cl[j] & 3) == (cl[j + L] & 3)))) break;   // run at the contig end never moves (mut.c:506)
    }
    return j < lo ? lo : j;                            // last cell read
}
DW_DEV int64_t reach_ins(const ContigDev &c, int h, int64_t p, int64_t lo)
{
    const uint32_t idx = ins_find(c.hap[h], p);
    const uint32_t n = c.hap[h].ins_len[idx];
    const uint8_t *P = c.hap[h].ins_bases + c.hap[h].ins_off[idx];
    // rotating left by one makes the cell's base the new first base: after r rotations the last
    // inserted base is P[n-1-r] while r < n, then ref[p-1-(r-n)] & 3 (bases rotated in earlier)
    int64_t j = p - 1; int64_t r = 0;
    for (; j >= lo; --j, ++r) {
        const bool mutated = ((c.hap[0].cells[j] | c.hap[1].cells[j]) & TMASK) != 0;
        const uint32_t last = r < (int64_t)n ? (uint32_t)P[n - 1 - r] : (uint32_t)(c.ref[p - 1 - (r - n)] & 3);
        if (!(mutated || last == (uint32_t)(c.ref[j] & 3))) break;
    }
    return j < lo ? lo : j;
}
__global__ void k_jreach(const Event *__restrict__ ev, Count nc, ContigDev c, int32_t *__restrict__ lo)
{
    const uint32_t n_cand = count_of(nc);
    adopt_device_sizes(c);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    const Event e = ev[k];
    int64_t reach = 0x7fffffff;
    if (e.live) {
        const int64_t p = e.pos, lo = c.seg.start[e.seg], hi = lo + c.seg.len[e.seg];
        reach = p;                                      // substitution: no scan; it only matters as a neighbour (prev_del)
        if (e.type == 2) reach = reach_del(c, (e.hap & 1) ? 0 : 1, p, lo, hi);         // haplotype 1 for hom and hap-1 events
        else if (e.type == 3) reach = reach_ins(c, (e.hap & 1) ? 0 : 1, p, lo);        // both copies carry the same bases before justification
        else if (e.type == 4) {                         // cell patched from a mutation-input file: whatever the two cells hold
            for (int h = 0; h < 2; ++h) {
                const uint8_t t = c.hap[h].cells[p] & TMASK;
                int64_t r = p;
                if (t == T_DEL) r = reach_del(c, h, p, lo, hi); else if (t == T_INS) r = reach_ins(c, h, p, lo);
                if (r < reach) reach = r;
            }
        }
    }
    lo[k] = (int32_t)reach;
}
// sufmin[k] = min(lo[k..n)): one block walks the array (or, seglen > 0, its segment of it) from the end; k_sufmin_fix folds in the minima of
// the segments behind
__global__ void k_sufmin(const int32_t *__restrict__ lo, Count nc, int32_t *__restrict__ sufmin, uint32_t seglen, int32_t *part)
{
    const uint32_t n = count_of(nc);
    const int64_t a = seglen ? (int64_t)blockIdx.x * seglen : 0, b = seglen ? (a + seglen < (int64_t)n ? a + seglen : (int64_t)n) : (int64_t)n;
    __shared__ int32_t sm[2][16];
    int32_t carry = 0x7fffffff; int buf = 0;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    const uint32_t nchunk = b > a ? (uint32_t)((b - a + blockDim.x - 1) / blockDim.x) : 0u;
    for (uint32_t ch = 0; ch < nchunk; ++ch, buf ^= 1) {
        // walk the range from its end: thread t handles element (b-1) - (ch*blockDim + t)
        const int64_t i = b - 1 - ((int64_t)ch * blockDim.x + threadIdx.x);
        int32_t v = i >= a ? lo[i] : 0x7fffffff;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(v, d); if (lane >= d && o < v) v = o; }   // inclusive min-scan
        if (lane == 63) sm[buf][wave] = v;
        __syncthreads();
        int32_t pre = carry, all = carry;                    // minimum of everything before this wave / of the whole chunk so far
        for (int w = 0; w < nw; ++w) { const int32_t t = sm[buf][w]; if (t < all) all = t; if (w < wave && t < pre) pre = t; }
        if (pre < v) v = pre;
        if (i >= a) sufmin[i] = v;
        carry = all;
    }
    if (seglen && threadIdx.x == 0) part[blockIdx.x] = carry;
}
__global__ void k_sufmin_fix(int32_t *__restrict__ sufmin, Count nc, uint32_t seglen, const int32_t *__restrict__ part)
{
    const uint32_t n = count_of(nc), i = blockIdx.x * blockDim.x + threadIdx.x, s = i / seglen;
    int32_t m = 0x7fffffff;
    for (uint32_t q = 0; q < WALK_SEGS; ++q) { const int32_t p = part[q]; if (q > s && p < m) m = p; }
    if (i < n && m < sufmin[i]) sufmin[i] = m;
}
__global__ void k_jbound(const Event *__restrict__ ev, Count nc, ContigDev c, const int32_t *__restrict__ sufmin, uint8_t *__restrict__ bound)
{
    const uint32_t n_cand = count_of(nc);
    adopt_device_sizes(c);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    uint8_t b = 0;
    if (ev[k].live) {
        int64_t a = (int64_t)k - 1;
        while (a >= 0 && !ev[a].live) --a;
        if (a < 0 || ev[a].seg != ev[k].seg) b = 1;       // the first live event of a contig: every contig is justified on its own
        else {
            const int64_t right_a = (int64_t)ev[a].pos + (ev[a].type == 2 ? (int64_t)ev[a].len - 1 : 0);
            if ((int64_t)sufmin[k] > right_a) {
                for (int64_t q = right_a + 1; q < (int64_t)ev[k].pos; ++q) if (c.ref[q] < 4) { b = 1; break; }
            }
        }
    }
    bound[k] = b;
}
__global__ void k_jrun(const Event *__restrict__ ev, Count nc, ContigDev c, const uint8_t *__restrict__ bound)
{
    const uint32_t n_cand = count_of(nc);
    adopt_device_sizes(c);
    const uint32_t k0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k0 >= n_cand || !bound[k0]) return;
    int prev_del[2] = {0, 0};
    int64_t last = -1;
    for (uint32_t k = k0; k < n_cand; ++k) {
        const Event e = ev[k];
        if (!e.live) continue;
        if (k > k0 && bound[k]) break;
        const int64_t p = e.pos, right = p + (e.type == 2 ? (int64_t)e.len - 1 : 0);
        const int64_t lo = c.seg.start[e.seg], hi = lo + c.seg.len[e.seg];      // (a cluster never spans two contigs: k_jbound)
        if (k > k0 && (prev_del[0] | prev_del[1]))      // an unmutated non-N position in the gap resets prev_del (mut.c:585-587)
            for (int64_t q = last + 1; q < p; ++q) if (c.ref[q] < 4) { prev_del[0] = prev_del[1] = 0; break; }
        for (int64_t i = p; i <= right; ++i) justify_visit(c, i, prev_del, lo, hi);
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
// per listed position: haplotype-1 cell | haplotype-2 cell << 8 | reference code << 16 | reference code of the position in front << 24 (the text writer
// needs no host copy of the sequence; a position that opens its contig finds the N of the padding in front of it and never prints it)
__global__ void k_gather(const int32_t *__restrict__ pos, uint32_t n, const uint8_t *__restrict__ ref, const uint8_t *__restrict__ h0, const uint8_t *__restrict__ h1, uint32_t *__restrict__ cells)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { const int32_t p = pos[k]; cells[k] = (uint32_t)h0[p] | ((uint32_t)h1[p] << 8) | ((uint32_t)ref[p] << 16) | ((uint32_t)(p > 0 ? ref[p - 1] : 4) << 24); }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (declared in dw_launch.hpp)
// ------------------------------------------------------------------------------------------------
void launch_pack(hipStream_t st, const uint8_t *ascii, uint8_t *ref, uint8_t *h0, uint8_t *h1, int64_t l)
{
    const uint64_t nchunk = (uint64_t)(l + 15) >> 4;
    uint32_t nb = cdiv(nchunk, 256); if (nb > (1u << 16)) nb = 1u << 16; if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_pack, dim3(nb), dim3(256), 0, st, ascii, ref, h0, h1, l);
}
void launch_site_scan_list(hipStream_t st, const uint8_t *refview, int64_t l, SegTab seg, WalkParams wp, uint64_t *status, uint64_t *ticket, int32_t *out, uint32_t cap, uint64_t *n_out)
{
    const uint32_t nb = (uint32_t)cdiv((uint64_t)l, (uint64_t)SITE_BLOCK_POS);
    hipLaunchKernelGGL(k_site_scan_list, dim3(nb), dim3(SITE_THREADS), 0, st, refview, l, seg, wp, status, ticket, out, cap, n_out, nb);
}
uint32_t site_scan_blocks(int64_t l) { return (uint32_t)cdiv((uint64_t)l, (uint64_t)SITE_BLOCK_POS); }
uint32_t site_scan_block_positions() { return (uint32_t)SITE_BLOCK_POS; }
// slots: nb x slot_cap entries; aux: 2 nb words (counts, bases); over: two words, zeroed by the caller (k_slot_scan)
void launch_site_scan_slots(hipStream_t st, const uint8_t *refview, int64_t l, SegTab seg, WalkParams wp, int32_t *slots, uint32_t slot_cap, uint32_t *aux, int32_t *out, uint32_t cap, uint64_t *n_out, uint32_t *over)
{
    const uint32_t nb = site_scan_blocks(l);
    if (!nb) return;
    hipLaunchKernelGGL(k_site_scan_slots, dim3(nb), dim3(SITE_THREADS), 0, st, refview, l, seg, wp, slots, slot_cap, aux);
    hipLaunchKernelGGL(k_slot_scan, dim3(1), dim3(1024), 0, st, aux, nb, slot_cap, aux + nb, n_out, over);
    hipLaunchKernelGGL(k_slot_gather, dim3(nb), dim3(256), 0, st, slots, slot_cap, aux, aux + nb, out, cap);
}
void launch_mark_dirty(hipStream_t st, const Event *ev, Count n, const int32_t *lo, uint32_t *dirty)
{
    if (n.host) hipLaunchKernelGGL(k_mark_dirty, dim3(cdiv(n.host, 256)), dim3(256), 0, st, ev, n, lo, dirty);
}
void launch_dirty_chunks(hipStream_t st, bool restore, const uint32_t *dirty, uint32_t n_words, int64_t l_live, const uint8_t *ref, const uint8_t *refview, const uint16_t *refsumm, const uint16_t *refsumm2,
                         uint8_t *cells0, uint8_t *cells1, uint8_t *view0, uint8_t *view1, uint16_t *summ0, uint16_t *summ1, uint16_t *summ2_0, uint16_t *summ2_1)
{
    if (!n_words) return;
    // a lane per chunk while that is at most 2^21 lanes (groups of up to 128 Mb: where the two forms meet, 0.29 us per Mb + 80 us against 0.79 us per Mb);
    // a thread per bitmap word beyond (k_dirty_chunks).  DWGSIM_HIP_DIRTY_MAP=word|chunk forces either (tests)
    bool per_chunk = (uint64_t)n_words * 32 <= (1ull << 21);
    if (const char *e = getenv("DWGSIM_HIP_DIRTY_MAP")) per_chunk = e[0] == 'c';
    const dim3 grid(cdiv((uint64_t)n_words * (per_chunk ? 32 : 1), 256));
    if (restore) { if (per_chunk) hipLaunchKernelGGL((k_dirty_chunks<true, true>), grid, dim3(256), 0, st, dirty, n_words, l_live, ref, refview, refsumm, refsumm2, cells0, cells1, view0, view1, summ0, summ1, summ2_0, summ2_1); else hipLaunchKernelGGL((k_dirty_chunks<true, false>), grid, dim3(256), 0, st, dirty, n_words, l_live, ref, refview, refsumm, refsumm2, cells0, cells1, view0, view1, summ0, summ1, summ2_0, summ2_1); }
    else { if (per_chunk) hipLaunchKernelGGL((k_dirty_chunks<false, true>), grid, dim3(256), 0, st, dirty, n_words, l_live, ref, refview, refsumm, refsumm2, cells0, cells1, view0, view1, summ0, summ1, summ2_0, summ2_1); else hipLaunchKernelGGL((k_dirty_chunks<false, false>), grid, dim3(256), 0, st, dirty, n_words, l_live, ref, refview, refsumm, refsumm2, cells0, cells1, view0, view1, summ0, summ1, summ2_0, summ2_1); }
}
void launch_scan_excl(hipStream_t st, uint32_t *data, uint32_t n, uint64_t *total_out)
{
    hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, st, data, n, total_out);
}
void launch_compact(hipStream_t st, const uint16_t *mask, const uint32_t *block_base, int32_t *out, int64_t l, uint32_t cap)
{
    hipLaunchKernelGGL(k_compact, dim3(cdiv((uint64_t)l, SCAN_POS_PER_BLOCK)), dim3(SCAN_THREADS), 0, st, mask, block_base, out, cap);
}
// The launchers below size their grids for n.host elements (an exact count or a capacity); the kernels work on min(*n.dev, n.host).
void launch_events(hipStream_t st, const int32_t *cand, Count n, const uint8_t *ref, SegTab seg, WalkParams wp, Event *ev, uint32_t *max_del)
{
    if (n.host) hipLaunchKernelGGL(k_events, dim3(cdiv(n.host, 256)), dim3(256), 0, st, cand, n, ref, seg, wp, ev, max_del);
}
void launch_resolve(hipStream_t st, Event *ev, Count n, const uint32_t *max_del, uint4 *flags, uint32_t *tot4)
{
    if (n.host) hipLaunchKernelGGL(k_resolve, dim3(cdiv(n.host, 256)), dim3(256), 0, st, ev, n, max_del, flags);
    if (n.host < g_walk_seg_min_rows) { hipLaunchKernelGGL(k_scan4, dim3(1), dim3(1024), 0, st, flags, n, tot4, 0u, (uint4 *)nullptr); return; }
    const uint32_t seglen = (n.host + WALK_SEGS - 1) / WALK_SEGS;                    // (flags has room for WALK_SEGS more rows behind its n.host: the segment totals)
    hipLaunchKernelGGL(k_scan4, dim3(WALK_SEGS), dim3(1024), 0, st, flags, n, tot4, seglen, flags + n.host);
    hipLaunchKernelGGL(k_scan4_fix, dim3(cdiv(n.host, 256)), dim3(256), 0, st, flags, n, tot4, seglen, (const uint4 *)(flags + n.host));
}
void launch_apply(hipStream_t st, Event *ev, Count n, const uint4 *flags, ContigDev c, WalkParams wp)
{
    if (n.host) hipLaunchKernelGGL(k_apply, dim3(cdiv(n.host, 256)), dim3(256), 0, st, ev, n, flags, c, wp);
}
void launch_justify_seq(hipStream_t st, const Event *ev, Count n, ContigDev c)
{
    if (n.host) hipLaunchKernelGGL(k_justify_seq, dim3(1), dim3(64), 0, st, ev, n, c);
}
void launch_justify(hipStream_t st, const Event *ev, Count n, ContigDev c, int32_t *lo, int32_t *sufmin, uint8_t *bound)
{
    if (!n.host) return;
    hipLaunchKernelGGL(k_jreach, dim3(cdiv(n.host, 256)), dim3(256), 0, st, ev, n, c, lo);
    if (n.host < g_walk_seg_min_rows) hipLaunchKernelGGL(k_sufmin, dim3(1), dim3(1024), 0, st, lo, n, sufmin, 0u, (int32_t *)nullptr);
    else {                                                                          // (sufmin has room for WALK_SEGS more entries behind its n.host: the segment minima)
        const uint32_t seglen = (n.host + WALK_SEGS - 1) / WALK_SEGS;
        hipLaunchKernelGGL(k_sufmin, dim3(WALK_SEGS), dim3(1024), 0, st, lo, n, sufmin, seglen, sufmin + n.host);
        hipLaunchKernelGGL(k_sufmin_fix, dim3(cdiv(n.host, 256)), dim3(256), 0, st, sufmin, n, seglen, (const int32_t *)(sufmin + n.host));
    }
    hipLaunchKernelGGL(k_jbound, dim3(cdiv(n.host, 256)), dim3(256), 0, st, ev, n, c, sufmin, bound);
    hipLaunchKernelGGL(k_jrun, dim3(cdiv(n.host, 64)), dim3(64), 0, st, ev, n, c, bound);
}
// mut.c:379-425 mut_debug(): the consistency asserts the reference runs over both haplotypes before and after the left-justification
// (mut.c:753, :757; live in its build -- a violating mutation input ends the reference with SIGABRT).  16 positions per thread; the smallest
// failing position and which assert it fails go to *verdict as (position << 8 | code), code 1..3:
//   1 hom substitution whose base equals the reference's   (c[0]&0x3) != (c[1]&0x3)
//   2 het substitution, both haplotypes carry the same base (c[1]&0x3) != (c[2]&0x3)
//   3 het substitution, neither haplotype keeps the reference base
// (the insertion-length asserts cannot fail here: every insertion is created with at least one base.)
__global__ void __launch_bounds__(SCAN_THREADS) k_mut_debug(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ h0, const uint8_t *__restrict__ h1, int64_t l, uint64_t *verdict)
{
    const int64_t first = ((int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_POS_PER_THREAD;
    if (first >= l) return;
    const uint4 a = *reinterpret_cast<const uint4 *>(h0 + first), b = *reinterpret_cast<const uint4 *>(h1 + first);
    if ((((a.x | a.y | a.z | a.w) | (b.x | b.y | b.z | b.w)) & 0x30303030u) == 0) return;       // no mutated cell among these 16 (cells are padded past l with unmutated N)
    const uint4 r = *reinterpret_cast<const uint4 *>(ref + first);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, rw[4] = {r.x, r.y, r.z, r.w};
    for (int q = 0; q < SCAN_POS_PER_THREAD && first + q < l; ++q) {
        const uint32_t c0 = (rw[q >> 2] >> (8 * (q & 3))) & 0xffu, c1 = (aw[q >> 2] >> (8 * (q & 3))) & 0xffu, c2 = (bw[q >> 2] >> (8 * (q & 3))) & 0xffu;
        if (c0 >= 4 || ((c1 & TMASK) == T_NONE && (c2 & TMASK) == T_NONE)) continue;
        uint32_t code = 0;
        if ((c1 & BTMASK) == (c2 & BTMASK)) { if ((c1 & TMASK) == T_SUB && (c0 & 3) == (c1 & 3)) code = 1; }
        else if ((c1 & TMASK) == T_SUB || (c2 & TMASK) == T_SUB) {
            if ((c1 & 3) == (c2 & 3)) code = 2;
            else if (!((c0 & 3) == (c1 & 3) || (c0 & 3) == (c2 & 3))) code = 3;
        }
        if (code) { atomicMin((unsigned long long *)verdict, ((unsigned long long)(first + q) << 8) | code); return; }
    }
}
void launch_mut_debug(hipStream_t st, const uint8_t *ref, const uint8_t *h0, const uint8_t *h1, int64_t l, uint64_t *verdict)
{
    if (l > 0) hipLaunchKernelGGL(k_mut_debug, dim3(cdiv((uint64_t)l, SCAN_POS_PER_BLOCK)), dim3(SCAN_THREADS), 0, st, ref, h0, h1, l, verdict);
}
// The read view of a finished haplotype (HapDev::view): one nibble per cell, 32 cells per thread.
__global__ void __launch_bounds__(256) k_make_view(const uint8_t *__restrict__ cells0, const uint8_t *__restrict__ cells1, int64_t n_cells, int64_t l_live, uint8_t *__restrict__ view0, uint8_t *__restrict__ view1,
                                                   uint16_t *__restrict__ summ0, uint16_t *__restrict__ summ1, uint16_t *__restrict__ summ2_0, uint16_t *__restrict__ summ2_1)
{
    const uint8_t *__restrict__ cells = blockIdx.y ? cells1 : cells0;       // grid.y = haplotype
    uint8_t *__restrict__ view = blockIdx.y ? view1 : view0;
    uint16_t *__restrict__ summ = blockIdx.y ? summ1 : summ0, *__restrict__ summ2 = blockIdx.y ? summ2_1 : summ2_0;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, first = t * 32;
    const bool live = first < n_cells;                             // n_cells (the padded length) is a multiple of 16
    uint32_t out[4] = {0, 0, 0, 0}, indel = 0, non_acgt = 0;
    if (live) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (first + 16 * q >= n_cells) break;
            const uint4 v = *reinterpret_cast<const uint4 *>(cells + first + 16 * q);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const uint32_t c = (w[b >> 2] >> (8 * (b & 3))) & 0xffu, ty = c & TMASK, base = c & 0xfu;
                const uint32_t nib = ty == T_NONE ? (base < 4 ? base : base == 4 ? 8u : 9u) : (ty == T_SUB && base < 4) ? 4u + base : 15u;
                const int cell = 16 * q + b;
                out[cell >> 3] |= nib << (4 * (cell & 7));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t rem = l_live - (first + 16 * q + 4 * k);               // cells of this word that belong to the group (the rest is padding)
                const uint32_t lm = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : ((1u << (8 * (int)rem)) - 1u);
                indel += (uint32_t)__popc(w[k] & lm & 0x10101010u);
                non_acgt |= w[k] & lm & 0x0C0C0C0Cu;
            }
        }
        *reinterpret_cast<uint4 *>(view + (first >> 1)) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    // the haplotype summaries dwgsim_hip_count_random reads (dw_simulate.hip k_place): per SUMM_CELLS and per SUMM2_CELLS cells, how many are
    // INSERT / DELETE cells (bit 4) and whether any holds a base code >= 4.  Every lane takes part in the shuffles.
    uint32_t i1 = indel, f1 = non_acgt ? 1u : 0u;
#pragma unroll
    for (int d = 1; d < SUMM_CELLS / 32; d <<= 1) { i1 += (uint32_t)__shfl_xor((int)i1, d); f1 |= (uint32_t)__shfl_xor((int)f1, d); }
    uint32_t i2 = i1, f2 = f1;
#pragma unroll
    for (int d = SUMM_CELLS / 32; d < SUMM2_CELLS / 32; d <<= 1) { i2 += (uint32_t)__shfl_xor((int)i2, d); f2 |= (uint32_t)__shfl_xor((int)f2, d); }
    if (first < l_live) {
        if ((t & (SUMM_CELLS / 32 - 1)) == 0) summ[t / (SUMM_CELLS / 32)] = (uint16_t)(i1 | (f1 ? 0x8000u : 0u));
        if ((t & (SUMM2_CELLS / 32 - 1)) == 0) summ2[t / (SUMM2_CELLS / 32)] = (uint16_t)(i2 | (f2 ? 0x8000u : 0u));
    }
}
void launch_make_view(hipStream_t st, const uint8_t *cells0, const uint8_t *cells1, int64_t n_cells, int64_t l_live, uint8_t *view0, uint8_t *view1, uint16_t *summ0, uint16_t *summ1, uint16_t *summ2_0, uint16_t *summ2_1)
{
    if (n_cells > 0) hipLaunchKernelGGL(k_make_view, dim3(cdiv((uint64_t)n_cells, 256 * 32), 2), dim3(256), 0, st, cells0, cells1, n_cells, l_live, view0, view1, summ0, summ1, summ2_0, summ2_1);
}
void launch_apply_patches(hipStream_t st, const int32_t *pos, const uint16_t *cells, uint32_t n, uint8_t *h0, uint8_t *h1)
{
    if (n) hipLaunchKernelGGL(k_apply_patches, dim3(cdiv(n, 256)), dim3(256), 0, st, pos, cells, n, h0, h1);
}
void launch_collect_mask(hipStream_t st, const uint8_t *h0, const uint8_t *h1, int64_t l, uint16_t *mask, uint32_t *block_count)
{
    hipLaunchKernelGGL(k_collect_mask, dim3(cdiv((uint64_t)l, SCAN_POS_PER_BLOCK)), dim3(SCAN_THREADS), 0, st, h0, h1, l, mask, block_count);
}
void launch_gather(hipStream_t st, const int32_t *pos, uint32_t n, const uint8_t *ref, const uint8_t *h0, const uint8_t *h1, uint32_t *cells)
{
    if (n) hipLaunchKernelGGL(k_gather, dim3(cdiv(n, 256)), dim3(256), 0, st, pos, n, ref, h0, h1, cells);
}

} // namespace dw
