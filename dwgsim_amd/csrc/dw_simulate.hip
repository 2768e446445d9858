// dw_simulate.hip -- hand-written HIP kernels (gfx950 / MI355X) of the per-read-pair loop
// (replaces the loop body src/dwgsim.c:636-1099 of the reference).
//
//     k_simulate      per read end: attempt loop (placement, haplotype, strands, base extraction through indels, N filter),
//                     errors (Illumina base space / SOLiD colour space / Ion Torrent flow space), qualities, FASTQ formatting,
//                     decoupled look-back for record offsets, 64-byte burst stores
//     k_place, k_place_rest  dwgsim_hip_count_random (sharding): random reads in a read-index range without producing them
//     k_calibrate     -B: per-base calibration of the Ion Torrent flow error (dwgsim_opt.c:415-457)
//     k_selftest_fp64 device self-test of the range-restricted fp64 forms (dw_common.hpp)
//
// Byte/integer work plus fp64 for the normals: no MFMA.  fp64 expressions mirror the reference's evaluation order; compile
// with -ffp-contract=off.
//
// The file is compiled in parts so the k_simulate variants build in parallel (csrc/Makefile):
//   DW_PART 0: k_place, k_place_rest, k_selftest_fp64, host launchers and the k_simulate dispatcher
//   DW_PART 1..6: k_simulate<LPP, *, DT> for (LPP, DT) = (2,0) (1,0) (2,2) (1,2) (2,1) (1,1); part 4 also holds k_calibrate
//   DW_PART 7, 8: the one-wave-per-block variants for long Illumina / SOLiD reads
//   DW_PART 9, 10: the two-kernel form (SPLIT) of the paired / single-end Illumina variants
//   DW_PART 11, 12, 13: the Ion Torrent variants whose read buffers live in LDS (256-lane blocks paired / single-end, the smaller blocks)
//   DW_PART 14, 15: ... their two-kernel form, paired / single-end
//   DW_PART -1 (default): everything in one translation unit
#include <algorithm>
#include "dw_read.hpp"
#include "dw_launch.hpp"

#ifndef DW_PART
#define DW_PART -1
#endif
#define DW_HAS(part) (DW_PART == -1 || DW_PART == (part))
#ifndef DW_SIM_WAVES
#define DW_SIM_WAVES 5       // minimum waves per SIMD requested for the Illumina variants (one less when both output families are written):
                             // the kernel sits 1-2 VGPRs above these occupancy steps without the hint; measured +4 % at 5 vs 4 waves, 6 spills (so do the SOLiD variants with any hint)
#endif
#ifndef DW_PRIO_DROP
#define DW_PRIO_DROP 1       // where a wave of the single kernel gives up its raised issue priority: 0 = once its block's look-backs are resolved, 1 = after
                             // the name line that follows them (profiles/r04_split.txt section 5: 5.84 -> 5.75 ms at 2 x 150 bp, Ion Torrent unchanged)
#endif
#ifndef DW_SOLID_WAVES
#define DW_SOLID_WAVES 4     // minimum waves per SIMD requested for the SOLiD variants: without a hint (1) the round-5 kernel takes 152 registers = three waves (126 in
                             // round 4); capped at 128 it spills 20-46 registers and is 11-33 % FASTER (2 x 50 -o 0 / -o 1 / 75 + 35 -o 2: profiles/r05_bench_lines_final.txt section 7)
#endif
#ifndef DW_QUAL_FIFO
#ifndef DW_ONE_LB
#define DW_ONE_LB 1          // 1: the single Illumina kernel places its records with ONE look-back (k_simulate, "ONE_LB"); 0: the three of rounds 2-5 (analysis builds)
#endif
#define DW_QUAL_FIFO 0       // 1: the quality line's pairs of characters placed by two-byte LDS stores (quality_line_fifo); 0: compacted in registers as in rounds
                             // 2-4.  Measured (profiles/r05_bench_lines_final.txt, one box each): 14.0 k -> 13.2-13.5 k VALU per wave, but four more LDS accesses per
                             // block (SQ_WAIT_INST_LDS x 4, bank conflicts x 4): 2 x 150 -o 1 1 % SLOWER, E. coli-sized 7 % slower, -o 0 and 2 x 250 equal: not adopted
#endif
#ifndef DW_SIM_WAVES_BOTH
#define DW_SIM_WAVES_BOTH 4  // ... when both output families are written (-o 0) through two register writers (WR = 0; through the FIFO one image serves both)
#endif
#ifndef DW_SIMB_WAVES
#define DW_SIMB_WAVES 6      // ... requested for the second half of the two-kernel form (text assembly)
#endif
#ifndef DW_IONA_WAVES
#define DW_IONA_WAVES 4      // ... for the first half of its two-kernel form (no text FIFOs in LDS: four blocks per CU)
#endif
#ifndef DW_IONL_WAVES
#define DW_IONL_WAVES 3      // ... for the Ion Torrent variant whose read buffers live in LDS: LDS, not registers, bounds its residency
#endif
#ifndef DW_ION_WAVES
#define DW_ION_WAVES 5       // minimum waves per SIMD requested for the (latency-bound) Ion Torrent variants: without the hint the window registers of the extraction push them to 104 VGPRs = 4 waves (measured 165 -> 187 M reads/s at 5; 6 brings nothing)
#endif

namespace dw {

#if DW_HAS(0)
// (the haplotype summaries k_place reads are written by k_make_view at the end of the walk, dw_walk.hip: one word per SUMM_CELLS (64) cells and one
// per SUMM2_CELLS (1024) -- how many of them are INSERT / DELETE cells, and whether any holds a base code >= 4)
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

// The same for a PAIR whose insert size is not known yet -- from two Philox blocks, the random-read test and the position uniform of the first
// placement try, instead of four plus the fp64 polar normal.  Without -x and -a the first try always stands (dwgsim.c:672-675: pos <= l - d and
// d >= s0 + s1), pos = (int)((l - d + 1) u) falls as d grows, and d = (int)(normal * std_dev + dist + 0.5) clamped to [s0 + s1, l] lies within
// dist +- place_k: the polar method delivers |normal| <= sqrt(-2 ln rsq) with rsq >= 2^-104 (v = 2u - 1 is a multiple of 2^-52), i.e. below
// 12.01; the host takes 12.1 std_dev + 2.  Whatever the strands, both reads start inside [pos, pos + d + s0 + s1) (read_geom) and walk at most
// 2s + 2 cells from there.  If the coarse summaries of that whole span show no base code >= 4 and at most min(s0, s1) indel cells, both reads
// satisfy attempt_surely_accepted for every insert size and strand the pair can still draw.  (hap: from the same block as the random-read test.)
DW_DEV bool pair_surely_accepted(const SimArgs &a, const SegCtx &sc, RngKey key, uint64_t ii, uint32_t att, const U4 &b0)
{
    const int32_t s0 = a.p.len[0], s1 = a.p.len[1], smax = s0 > s1 ? s0 : s1, smin = s1 > 0 && s1 < s0 ? s1 : s0;
    const int64_t l = sc.l;
    if (l < (int64_t)s0 + s1 + 1 || sc.l_place != sc.l) return false;      // (a placement length of its own -- dwgsim_hip_contig_set_placement_length -- draws other positions: draw_pair decides)
    int64_t dlo = 0, dhi = 0;
    if (s1 > 0) {
        dlo = (int64_t)a.p.dist - a.place_k; dhi = (int64_t)a.p.dist + a.place_k;
        if (dlo < s0 + s1) dlo = s0 + s1;
        if (dhi < s0 + s1) dhi = s0 + s1;
        if (dlo > l) dlo = l;
        if (dhi > l) dhi = l;
    }
    const double u = u_lo(rng_block(key, D_PLACE, ii, att, 0, 0));                                 // slot 0: the position uniform of try 0 (dwgsim.c:671)
    const int64_t pos_lo = (int64_t)(int32_t)((double)(l - dhi + 1) * u), pos_hi = (int64_t)(int32_t)((double)(l - dlo + 1) * u);
    const int64_t lo = pos_lo - (2 * (int64_t)smax + 2), hi = pos_hi + dhi + s0 + s1 + 2 * (int64_t)smax + 2;
    if (lo < 0 || hi >= l) return false;
    const int hap = u_hi(b0) < a.p.mut_freq ? 0 : 1;
    const uint16_t *summ2 = (hap ? a.summ2[1] : a.summ2[0]) + sc.start / SUMM2_CELLS;
    uint32_t indel = 0, flags = 0;
    for (int64_t b = lo / SUMM2_CELLS; b <= hi / SUMM2_CELLS; ++b) { const uint32_t v = summ2[b]; indel += v & 0x7fffu; flags |= v; }
    return !(flags & 0x8000u) && indel <= (uint32_t)smin;
}

// one pair, exactly: the attempt that is accepted (dwgsim.c:833-843 retry rule) and whether it ends as a random read
DW_DEV void place_pair_exact(const SimArgs &a, const SegCtx &sc, RngKey key, uint64_t ii, bool &is_rand, bool &failed, uint32_t &att)
{
    att = 0; is_rand = false; failed = false;
    for (;;) {
        const U4 b0 = rng_block(key, D_PAIR, ii, att, 0, 0);
        if (!(a.p.rand_read < u_lo(b0))) { is_rand = true; return; }
        if (a.place_fast && pair_surely_accepted(a, sc, key, ii, att, b0)) return;
        const PairDraw pd = draw_pair(a, sc, key, ii, att);
        if (pd.is_rand) { is_rand = true; return; }      // (a placement that gave up on the target regions ends as a discarded random read: draw_pair)
        bool ok = true;
        for (int j = 0; j < 2 && ok; ++j) {
            const int sj = sel_len(a, j);
            if (sj <= 0) continue;
            int64_t start; int step;
            read_geom(a, sc, pd, j, &start, &step);
            if (!attempt_surely_accepted((pd.hap ? a.summ[1] : a.summ[0]) + sc.start / SUMM_CELLS, sc.l, start, step, sj)) {     // rare: N, dense indels, contig ends
                const ReadRes r = gen_read(sel_hap(a, sc, pd.hap), sc.l, start, step, sj, j ? pd.strand1 : pd.strand0, NoSink{});
                ok = r.ext_coor >= 0 && r.num_n <= a.p.max_n;
            }
        }
        if (ok) return;
        if (++att > (uint32_t)MAX_ATTEMPTS) { failed = true; return; }
    }
}

// K5 (dwgsim_hip_count_random: sharding): how many pairs of every range end as random reads.  One lane per PAIR.  k_place settles what two
// Philox blocks can settle -- a random read at the first attempt, or a genomic pair whose whole neighbourhood is clean -- and hands the rest
// (windows near N runs, contig ends or dense indels: a per cent or so) to k_place_rest, packed, so that no wave drags 63 settled lanes through
// the long path.
__global__ void __launch_bounds__(PLACE_PAIRS) k_place(SimArgs a)
{
    __shared__ uint32_t s_cnt[PLACE_PAIRS / 64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SegPtr sg = as_constant(a.segs) + seg_of_block(as_constant(a.segs), a.n_seg, blockIdx.x);          // the read-index range this block works on
    const SegCtx sc = seg_ctx(a, sg);
    const uint64_t pair = (uint64_t)(blockIdx.x - sg->first_block) * PLACE_PAIRS + (uint64_t)tid;      // inside the range
    const bool valid = pair < sg->n_pairs;
    const uint64_t ii = sg->first_ii + pair;
    const RngKey key{a.p.seed, uniform_u32(sg->contig_index)};
    bool is_rand = false, open = false;
    if (valid) {
        const U4 b0 = rng_block(key, D_PAIR, ii, 0, 0, 0);
        is_rand = !(a.p.rand_read < u_lo(b0));                                                      // dwgsim.c:649
        open = !is_rand && !(a.place_fast && pair_surely_accepted(a, sc, key, ii, 0, b0));
    }
    const uint64_t rm = __ballot(is_rand), om = __ballot(open);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(rm);
    if (om) {      // the wave's open pairs go to one of the lists, in one piece
        const uint32_t li = (blockIdx.x * (PLACE_PAIRS / 64) + (uint32_t)wave) % PLACE_LISTS;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.place_list_n[16 * li], (uint32_t)__popcll(om));
        base = (uint32_t)__shfl((int)base, 0);
        const uint32_t at = base + (uint32_t)__popcll(om & ((1ull << lane) - 1ull));
        if (open) {
            if (at < a.place_list_cap) a.place_list[(size_t)li * a.place_list_cap + at] = blockIdx.x * (uint32_t)PLACE_PAIRS + (uint32_t)tid;
            else atomicOr((unsigned long long *)&a.counters[2], 16ull);                             // (a list overflowed: the host runs the count again with room for every pair)
        }
    }
    __syncthreads();
    if (tid == 0) { uint32_t t = 0; for (int w = 0; w < PLACE_PAIRS / 64; ++w) t += s_cnt[w]; a.block_rand[blockIdx.x] = t; }
}

// ... the pairs k_place left open, one lane each.  A wave may hold pairs of several ranges (contigs): they are worked off range by range, so that
// the range's fields and the RNG key stay wave-uniform.
__global__ void __launch_bounds__(128) k_place_rest(SimArgs a)
{
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t li = blockIdx.x % PLACE_LISTS, sub = blockIdx.x / PLACE_LISTS, nsub = gridDim.x / PLACE_LISTS;
    uint32_t n = a.place_list_n[16 * li];
    if (n > a.place_list_cap) n = a.place_list_cap;
    for (uint32_t base = sub * 128u; base < n; base += nsub * 128u) {                               // (base is block-uniform: every lane of a wave runs the same iterations)
        const uint32_t q = base + threadIdx.x;
        bool todo = q < n;
        uint32_t e = 0, myseg = 0;
        if (todo) { e = a.place_list[(size_t)li * a.place_list_cap + q]; myseg = seg_of_block(as_constant(a.segs), a.n_seg, e / PLACE_PAIRS); }
        uint32_t retries = 0;
        for (;;) {
            const uint64_t m = __ballot(todo);
            if (!m) break;
            const uint32_t s = uniform_u32((uint32_t)__shfl((int)myseg, __ffsll((unsigned long long)m) - 1));
            if (todo && myseg == s) {
                const SegPtr sg = as_constant(a.segs) + s;
                const SegCtx sc = seg_ctx(a, sg);
                const uint64_t pair = (uint64_t)(e / PLACE_PAIRS - sg->first_block) * PLACE_PAIRS + (uint64_t)(e % PLACE_PAIRS);
                const RngKey key{a.p.seed, uniform_u32(sg->contig_index)};
                bool is_rand, failed; uint32_t att;
                place_pair_exact(a, sc, key, sg->first_ii + pair, is_rand, failed, att);
                if (is_rand) atomicAdd((unsigned long long *)&a.range_rand[s], 1ull);
                if (failed) atomicOr((unsigned long long *)&a.counters[2], 1ull);
                retries = att;
                todo = false;
            }
        }
        const uint32_t rs = wave_sum_u32(retries);
        if (lane == 0 && rs) atomicAdd((unsigned long long *)&a.counters[1], (unsigned long long)rs);
    }
}

// random reads per range: what k_place counted per block (block_rand, scanned: exclusive prefix; total in counters[3]) + what k_place_rest added
__global__ void __launch_bounds__(256) k_range_counts(SimArgs a)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= (uint32_t)a.n_seg) return;
    const SegPtr sg = as_constant(a.segs);
    const uint64_t lo = a.block_rand[sg[q].first_block], hi = q + 1 < (uint32_t)a.n_seg ? (uint64_t)a.block_rand[sg[q + 1].first_block] : a.counters[3];
    a.range_rand[q] += hi - lo;
    if (q == 0) { uint64_t open = 0; for (int li = 0; li < PLACE_LISTS; ++li) open += a.place_list_n[16 * li]; a.counters[5] = open; }      // (how many pairs took the long path: analysis)
}
#endif // DW_HAS(0): k_place

// K6: one lane per read end (LPP = 2: lanes 2q / 2q+1 are the two ends of pair q; LPP = 1: single end).
// DW_PROBE_INIT / DW_PROBE_MARK (phase clocks) and probe::off (parts switched off to weigh them): dw_probe.hpp -- nothing in the product build.

// ---- pieces of a FASTQ record shared by the Illumina / Ion Torrent and the SOLiD write paths ----
// '@' + "[prefix_]contig" (or "[prefix_]rand"): whole words from LDS (first 128 bytes), any rest from HBM
template <class O>
DW_DEV void put_name_fixed(O &o, const uint32_t *fw, const uint8_t *fx, uint32_t fixed_len)
{
    const uint32_t flen = fixed_len + 1, inl = flen < 128u ? flen : 128u;
    uint32_t q = 0;
    for (; q + 4 <= inl; q += 4) o.put4(fw[q >> 2]);
    if (q < inl) o.putn((uint64_t)fw[q >> 2] & ((1ull << (8 * (inl - q))) - 1), inl - q);
    for (q = inl; q < flen; ++q) o.put(fx[q]);
}
// "_0_0_0_0_1_1_0:0:0_0:0:0_<hex>" of a random read (dwgsim.c:1044-1048)
template <class O>
DW_DEV void put_rand_tail(O &o, uint64_t rand_ii)
{
    o.putn(0x305F305F305F305Full, 8);          // "_0_0_0_0"
    o.putn(0x3A305F315F315F00ull >> 8, 7);     // "_1_1_0:"
    o.putn(0x303A305F303A30ull, 7);            // "0:0_0:0"
    o.putn(0x303Aull, 2);                      // ":0"
    put_hex(o, rand_ii, '_');
}
struct NameCounts { int32_t e0, u0, i0, e1, u1, i1; };     // n_err : n_sub : n_indel of read end 1 and 2
// "_pos1_pos2_strand1_strand2_0_0_e:s:i_e:s:i_<hex>" (dwgsim.c:923-929)
template <class O>
DW_DEV void put_pair_tail(O &o, int32_t x0, int32_t x1, uint32_t strand0, uint32_t strand1, const NameCounts &n, uint64_t ii)
{
    put_dec(o, (uint32_t)(x0 + 1), '_'); put_dec(o, (uint32_t)(x1 + 1), '_');
    o.putn((uint64_t)'_' | ((uint64_t)('0' + strand0) << 8) | ((uint64_t)'_' << 16) | ((uint64_t)('0' + strand1) << 24)
               | ((uint64_t)'_' << 32) | ((uint64_t)'0' << 40) | ((uint64_t)'_' << 48) | ((uint64_t)'0' << 56), 8);     // "_S_S_0_0"
    put_counts(o, (uint32_t)n.e0, (uint32_t)n.u0, (uint32_t)n.i0);
    put_counts(o, (uint32_t)n.e1, (uint32_t)n.u1, (uint32_t)n.i1);
    put_hex(o, ii, '_');
}
DW_DEV uint32_t pair_tail_len(int32_t x0, int32_t x1, const NameCounts &n, uint64_t ii)
{
    return 1 + ndigits10((uint32_t)(x0 + 1)) + 1 + ndigits10((uint32_t)(x1 + 1)) + 8     // _P0_P1 _S_S_0_0
         + counts_len((uint32_t)n.e0, (uint32_t)n.u0, (uint32_t)n.i0) + counts_len((uint32_t)n.e1, (uint32_t)n.u1, (uint32_t)n.i1) + 1 + ndigits16(ii);
}
// ---- quality normals (dwgsim.c:156-175 ran_normal, :912): the integer offsets (int)(nrm * sigma + 0.5) that one Philox block delivers.
// The polar tries of a read's quality string form one sequential stream of 16-BIT uniforms: try t = the two halves of word t & 3 of block
// t >> 2 (low half v1, high half v2; u = h * 2^-16, v = 2u - 1 = (h - 32768) * 2^-15) -- four tries per Philox block.  An accepted try delivers
// two normals, v2 * fac first and the cached v1 * fac second (dwgsim.c:170-174).  The only consumer of a normal is the truncation
// (int)(nrm * sigma + 0.5), which 2^32 distinct tries resolve far beyond what a quality character can show.
// EXACT form: fp64 arithmetic in the reference's evaluation order. ----
DW_DEV void quality_try_exact(uint32_t w, double sigma, bool &ok, int32_t &k0, int32_t &k1)
{
    const double v1 = (double)((int32_t)(w & 0xFFFFu) - 32768) * 0x1p-15, v2 = (double)((int32_t)(w >> 16) - 32768) * 0x1p-15;      // exact
    const double rsq = v1 * v1 + v2 * v2;
    ok = !(rsq >= 1.0 || rsq == 0.0);
    if (!ok) return;
    // rsq is a multiple of 2^-30 in (0, 1): -2 log(rsq) in [2^-30, 42], the quotient in [2^-30, 2^36] -- the range-restricted forms apply
    const double fac = sqrt_mid(div_mid(-2.0 * det_log<true>(rsq), rsq));
    k0 = (int32_t)(((v2 * fac) * sigma) + 0.5);
    k1 = (int32_t)(((v1 * fac) * sigma) + 0.5);
}
// (the same for the read kernels, an accepted try; DW_DEV_NOINLINE: dw_intrin.hpp)
DW_DEV_NOINLINE uint64_t quality_try_exact_called(uint32_t w, double sigma)
{
    bool ok; int32_t k0 = 0, k1 = 0;
    quality_try_exact(w, sigma, ok, k0, k1);
    return (uint64_t)(uint32_t)k0 | ((uint64_t)(uint32_t)k1 << 32);
}
// LAZY form -- same results, a fraction of the work.  An estimate y of x = nrm * sigma + 0.5 with a PROVEN bound |y - x| < eps decides the
// integer whenever y is further than eps from every integer; only the rest (about 2 * eps of all values) takes the exact path.  The estimate is
// fp32: correctly rounded IEEE operations plus v_log_f32 / v_rcp_f32 / v_sqrt_f32, whose errors on the operand ranges used here are
// established exhaustively on the device (k_selftest_lazy: every float of the range) -- and since a try is one 32-bit word, k_selftest_lazy
// also compares the decision of EVERY possible try with the exact form (2^32 words per quality_std tested).  Error budget (DESIGN.md "Lazy
// quality normals"); with s = h - 32768 (v = s * 2^-15):
//   R = s1^2 + s2^2 is an exact integer <= 2^31: accept / reject (0 < R < 2^30) is exact;
//   Rf = fl(R) has relative error <= 2^-24; rt = Rf * 2^-30 (exact scaling, in [2^-30, 1));
//   L' = -log2(rt) has absolute error a <= 1.44 * 2^-24 (from Rf) + 2^-23 (1 + L') (v_log_f32, measured on every float), <= 1.1 * 2^-22 for L' <= 1;
//   for L' >= Lmin that moves sqrt(L') by at most a / (2 sqrt Lmin), i.e. nrm = s sqrt(2 ln 2 L' / R) by <= 3.4e-7 / sqrt(Lmin) (|s| <= sqrt R);
//   everything else is relative: v_rcp_f32 and v_sqrt_f32 (each measured < 2^-23), three multiplies, the conversions, the constant, the
//   relative error of L' above 1: < 2^-20.3 in all, times |nrm| <= 6.5: 5.0e-6.  So |y - x| <= sigma (3.4e-7 / sqrt(Lmin) + 5.0e-6) + the rounding
//   of the last fma; eps = 1.5 x (sigma (3.4e-7 / sqrt(Lmin) + 7.2e-6)) + 2^-18 (host: lazy_quality_params -- the budget of the 32-bit layout, kept:
//   it is the larger one), and k_selftest_lazy measures the largest |y - x| / eps on the device;
//   L' < Lmin means |nrm| < sqrt(2 ln 2 Lmin): Lmin is chosen by the host so that |nrm * sigma| < 0.45 there -- the offset is 0 without
//   further work (near1_zero) -- or, for a large sigma, Lmin = 2^-10 and the exact path runs.
struct QualLazy { float k, eps, lmin; int32_t near1_zero; };      // k = sqrt(2 ln 2) * quality_std
// Branch-free.  Returns bit 0: the try is accepted, bit 1: its offsets must come from the exact form (k0 / k1 hold the estimate's otherwise).
DW_DEV uint32_t quality_try_lazy(uint32_t w, const QualLazy &ql, int32_t &k0, int32_t &k1)
{
    const uint32_t wx = w ^ 0x80008000u;                                              // h - 32768 as a signed 16-bit value
    const int32_t s1 = (int32_t)(int16_t)(wx & 0xFFFFu), s2 = (int32_t)wx >> 16;
    const uint32_t R = (uint32_t)(s1 * s1) + (uint32_t)(s2 * s2);                     // exact, <= 2^31
    const bool acc = R - 1u < 0x3FFFFFFFu;                                            // 0 < R < 2^30  <=>  0 < rsq < 1
    const float Rf = (float)R, rt = Rf * 0x1p-30f;
    const float Lp = -__builtin_amdgcn_logf(rt);                                      // v_log_f32 (log2)
    const float f = __builtin_amdgcn_sqrtf(Lp * __builtin_amdgcn_rcpf(Rf));
    const float y0 = __builtin_fmaf((float)s2 * f, ql.k, 0.5f), y1 = __builtin_fmaf((float)s1 * f, ql.k, 0.5f);
    const float d0 = __builtin_fabsf(y0 - __builtin_rintf(y0)), d1 = __builtin_fabsf(y1 - __builtin_rintf(y1));
    const bool near1 = Lp < ql.lmin;
    const bool sure = near1 ? ql.near1_zero != 0 : (d0 >= ql.eps && d1 >= ql.eps);     // (also false for an infinite eps: sigma out of the fp32 path's range)
    k0 = near1 ? 0 : (int32_t)y0; k1 = near1 ? 0 : (int32_t)y1;
    return (acc ? 1u : 0u) | ((acc && !sure) ? 2u : 0u);
}
// The quality characters one Philox block of a read end's try stream delivers -- at most eight, in order, little-endian in `blk`; returns how many.
// pos = position of the first of them.  qbw = the base quality characters per position, staged in LDS as packed bytes: nq entries followed by
// at least eight copies of the last one (positions >= nq reuse the last entry: Ion Torrent reads can outgrow the table, their error rate is
// uniform, dwgsim_opt.c:338-343).  (dwgsim.c:899-918: q = (char)(base quality + offset), clamped to '!' .. 'I'.)
DW_DEV uint32_t quality_block(const U4 &b, const QualLazy &ql, double sigma, const uint32_t *qbw, int nq, int pos, uint64_t &blk)
{
    const uint32_t w[4] = {b.x, b.y, b.z, b.w};
    int32_t k[8]; uint32_t acc = 0, need = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) { const uint32_t r = quality_try_lazy(w[t], ql, k[2 * t], k[2 * t + 1]); acc |= (r & 1u) << t; need |= (r >> 1) << t; }
    if (need) {                              // rare (about 2 eps of the values; the branch is skipped when no lane of the wave takes it): the reference's own arithmetic decides
#pragma unroll
        for (int t = 0; t < 4; ++t) if ((need >> t) & 1u) { const uint64_t kk = quality_try_exact_called(w[t], sigma); k[2 * t] = (int32_t)(uint32_t)kk; k[2 * t + 1] = (int32_t)(uint32_t)(kk >> 32); }
    }
    // the base qualities of the (up to eight) positions this block can fill
    const int pc = pos < nq ? pos : nq;
    const uint32_t q0 = qbw[pc >> 2], q1 = qbw[(pc >> 2) + 1], q2 = qbw[(pc >> 2) + 2];
    const uint64_t qb8 = (uint64_t)__builtin_amdgcn_alignbyte(q1, q0, (uint32_t)pc & 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(q2, q1, (uint32_t)pc & 3u) << 32);
    blk = 0; uint32_t nb = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t qb2 = (uint32_t)(qb8 >> (8 * nb));                         // the base qualities of this try's two positions
        int32_t qa = (int8_t)((int32_t)(int8_t)(qb2 & 0xffu) + k[2 * t]), qc = (int8_t)((int32_t)(int8_t)((qb2 >> 8) & 0xffu) + k[2 * t + 1]);
        qa = qa < 33 ? 33 : qa > 73 ? 73 : qa; qc = qc < 33 ? 33 : qc > 73 ? 73 : qc;
        const bool on = (acc >> t) & 1u;
        blk |= (uint64_t)(on ? ((uint32_t)qa | ((uint32_t)qc << 8)) : 0u) << (8 * nb);
        nb += on ? 2u : 0u;
    }
    return nb;
}
// The quality line of one read end, block by block (dwgsim.c:899-918): put(blk, n) receives the next n (1 .. 8) characters, little-endian in blk,
// upper bytes zero.  n_chars characters in all; skip_first: the first one is drawn but not written (SOLiD BWA records drop the first colour's
// quality, dwgsim.c:950-955).
template <class F>
DW_DEV void for_each_quality_block(const SimParams &p, RngKey key, uint32_t dom, uint64_t ii, uint32_t att, const uint32_t *qbw, int nq, int n_chars, bool skip_first, F &&put)
{
    auto deliver = [&](uint64_t blk, uint32_t nb, int pos) {       // clip to the line's length, drop the first character if asked to
        if ((int)nb > n_chars - pos) { nb = (uint32_t)(n_chars - pos); blk &= (1ull << (8 * nb)) - 1ull; }
        if (skip_first && pos == 0 && nb) { blk >>= 8; --nb; }
        if (nb) put(blk, nb);
    };
    if (p.fixed_quality >= 0 || !(0 < p.quality_std)) {           // -q: one character; -Q 0: the base quality as it is
        for (int pos = 0; pos < n_chars; pos += 8) {
            uint64_t blk;
            if (p.fixed_quality >= 0) blk = 0x0101010101010101ull * (uint64_t)(uint32_t)p.fixed_quality;
            else {
                const int pc = pos < nq ? pos : nq;
                const uint32_t q0 = qbw[pc >> 2], q1 = qbw[(pc >> 2) + 1], q2 = qbw[(pc >> 2) + 2];
                const uint32_t lo = __builtin_amdgcn_alignbyte(q1, q0, (uint32_t)pc & 3u), hi = __builtin_amdgcn_alignbyte(q2, q1, (uint32_t)pc & 3u);
                blk = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) { int32_t q = (int8_t)(((b < 4 ? lo : hi) >> (8 * (b & 3))) & 0xffu); q = q < 33 ? 33 : q > 73 ? 73 : q; blk |= (uint64_t)(uint32_t)q << (8 * b); }
            }
            deliver(blk, 8, pos);
        }
        return;
    }
    const QualLazy ql{p.q_k, p.q_eps, p.q_lmin, p.q_near1};
    int pos = 0; uint32_t t = 0;
    while (pos < n_chars) {
        const U4 b = rng_block(key, dom, ii, att, 0, t++);
        uint64_t blk;
        const uint32_t nb = quality_block(b, ql, p.quality_std, qbw, nq, pos, blk);
        deliver(blk, nb, pos);
        pos += (int)nb;
    }
}

// The same line through the FIFO writer (dw_read.hpp FifoWriter): the pair of characters of each of a block's four tries goes straight to its place
// in the FIFO -- an unaligned two-byte LDS store `accepted tries so far` x 2 places past the write position, a rejected try's pair overwritten by the
// next one's --, its two base qualities are picked from the eight loaded in front of the draws by one v_perm_b32 at the same offset: no compaction of the
// accepted pairs in registers (rounds 2-4: seven 64-bit shifts, selects and masks per block -- the LDS does the byte placement here as it does for every
// append).  (Loading the base qualities pair by pair at their final offsets, after the draws, exposed four LDS round trips per block: the VALU count
// fell by 5 % and the time did not, profiles/r05_bench_lines_final.txt.)
template <class W>
DW_DEV void quality_line_fifo(W &w, const SimParams &p, RngKey key, uint32_t dom, uint64_t ii, uint32_t att, const uint32_t *qbw, int nq, int n_chars)
{
    if (p.fixed_quality >= 0 || !(0 < p.quality_std)) { for_each_quality_block(p, key, dom, ii, att, qbw, nq, n_chars, false, [&](uint64_t blk, uint32_t nb) { w.putn(blk, nb); }); return; }
    const QualLazy ql{p.q_k, p.q_eps, p.q_lmin, p.q_near1};
    int pos = 0; uint32_t t = 0;
    while (pos < n_chars) {
        // the base qualities of the (up to eight) positions this block can fill: loaded before the draws, whose arithmetic hides the LDS round trip
        // (positions >= nq reuse the last entry: the table holds eight copies of it behind the nq)
        const int pc = pos < nq ? pos : nq;
        const uint32_t q0 = qbw[pc >> 2], q1 = qbw[(pc >> 2) + 1], q2w = qbw[(pc >> 2) + 2];
        const U4 b = rng_block(key, dom, ii, att, 0, t++);
        const uint32_t wd[4] = {b.x, b.y, b.z, b.w};
        int32_t k[8]; uint32_t acc = 0, need = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint32_t r = quality_try_lazy(wd[q], ql, k[2 * q], k[2 * q + 1]); acc |= (r & 1u) << q; need |= (r >> 1) << q; }
        if (need) {                              // rare: the reference's own arithmetic decides (quality_block)
#pragma unroll
            for (int q = 0; q < 4; ++q) if ((need >> q) & 1u) { const uint64_t kk = quality_try_exact_called(wd[q], p.quality_std); k[2 * q] = (int32_t)(uint32_t)kk; k[2 * q + 1] = (int32_t)(uint32_t)(kk >> 32); }
        }
        const uint32_t qlo = __builtin_amdgcn_alignbyte(q1, q0, (uint32_t)pc & 3u), qhi = __builtin_amdgcn_alignbyte(q2w, q1, (uint32_t)pc & 3u);
        uint32_t off = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t q2 = lut8(qhi, qlo, off * 0x0101u + 0x0100u);       // bytes off, off + 1 of the eight (one v_perm_b32)
            int32_t qa = (int8_t)((int32_t)(int8_t)(q2 & 0xffu) + k[2 * q]), qc = (int8_t)((int32_t)(int8_t)((q2 >> 8) & 0xffu) + k[2 * q + 1]);
            qa = qa < 33 ? 33 : qa > 73 ? 73 : qa; qc = qc < 33 ? 33 : qc > 73 ? 73 : qc;
            w.poke2(off, (uint32_t)qa | ((uint32_t)qc << 8));
            off += ((acc >> q) & 1u) ? 2u : 0u;
        }
        const int left = n_chars - pos;
        w.advance((int)off < left ? off : (uint32_t)left);
        pos += (int)off;
    }
}

// sum over i < c of the hexadecimal digits of b + i: what the running index adds to the names of c consecutive random reads (dwgsim.c:1044-1048)
DW_DEV uint64_t hex_digits_sum(uint64_t b, uint64_t c)
{
    uint64_t sum = 0;
    for (int d = 1; d <= 16 && c; ++d) {                     // numbers with d digits: [16^(d-1), 16^d), from 0 for d = 1
        const uint64_t hi = d == 16 ? ~0ull : (1ull << (4 * d)) - 1ull;      // the largest of them
        if (b > hi) continue;
        const uint64_t n = hi - b + 1ull < c && d < 16 ? hi - b + 1ull : c;
        sum += n * (uint64_t)d; b += n; c -= n;
    }
    return sum;
}
// ---- scratch slots (Ion Torrent read buffers; the staging of reads too long for LDS) ----
// A block's read buffers are a global scratch (dw_read.hpp flow_errors).  Indexed by the block's ticket they were fresh memory for every block:
// written once, read once, every byte of them through the L2 to HBM and back (4.2 x the algorithmic traffic, profiles/r03_ion_*).  Now a
// block TAKES one of `per_xcd` slots of its XCD and RELEASES it when it ends, so the scratch in use is what the resident blocks need (a few
// hundred KB per CU) and stays in the L2 / the memory-side cache.  Slots never cross XCDs: the eight L2s are not coherent with each other, and
// a slot's next owner must find (and overwrite) the previous owner's lines in the SAME L2 -- dirty lines of one address in two L2s would be
// written back in either order.  free list of XCD x (8-byte words, relaxed agent-scope atomics like the look-back's): header words
// ff[32 x] = slots taken so far, ff[32 x + 16] = slots released so far; queue ff[256 + x * n_blocks + r] = (slot + 1) of the r-th release.
// The q-th taker owns slot q outright while q < per_xcd, and otherwise waits for the (q - per_xcd)-th release -- which has happened already
// when per_xcd >= the blocks an XCD can hold (dw_host.cpp sizes it so).  The slot is taken BEFORE the ticket: every block that owns a ticket
// owns a slot, so the block with the smallest ticket still running never waits for anything and the look-backs stay deadlock-free whatever
// per_xcd is ("flow_slots" = 1 in the tests).  Every word of a slot is written by its owner before the owner reads it (nothing of the previous
// owner's is ever looked at), and the owner's stores are in the L2 before it lets go (wait_stores + the block's last barrier).
DW_DEV uint32_t scratch_slot_take(uint64_t *ff, uint32_t per_xcd, uint32_t n_blocks)
{
    const uint32_t x = xcc_id();
    const uint64_t q = (uint64_t)atomicAdd((unsigned long long *)&ff[32 * x], 1ull);
    if (q < per_xcd) return x * per_xcd + (uint32_t)q;
    uint64_t *e = ff + 256 + (size_t)x * n_blocks + (size_t)(q - per_xcd);
    uint64_t v;
    do { v = status_load(e); if (v == 0) __builtin_amdgcn_s_sleep(2); } while (v == 0);
    return (uint32_t)(v - 1);
}
DW_DEV void scratch_slot_release(uint64_t *ff, uint32_t n_blocks, uint32_t slot)
{
    const uint32_t x = xcc_id();
    const uint64_t r = (uint64_t)atomicAdd((unsigned long long *)&ff[32 * x + 16], 1ull);
    status_store(ff + 256 + (size_t)x * n_blocks + (size_t)r, (uint64_t)slot + 1ull);
}
// DT = 0: Illumina base-space errors; DT = 1: SOLiD colour space; DT = 3: Ion Torrent flow-space errors (variable read length), the read's one
// in-place 2-bit buffer in LDS (dw_read.hpp flow_errors); DT = 2: the same with the buffer in a scratch slot of global memory -- for reads whose
// buffers LDS cannot hold (very long reads, per-flow error rates at which reads grow severalfold).
// NTHR: lanes per block.  SIM_THREADS_LONG (one wave) is the variant for reads too long to stage at SIM_THREADS lanes.
// WR = 1: records leave through the per-lane LDS FIFO; 0: straight from registers (dw_read.hpp FifoWriter / Writer)
// SPLIT: 0 = the whole path in one kernel: a block learns where its records go from a decoupled look-back over the blocks in front of it.
// 1 / 2 = the same code cut in two at the point where the record lengths are known (Illumina, 256-lane blocks): the FIRST HALF (1) runs up to
// there and hands its state over through HBM -- the staged, finished bases, 16 bytes of name fields per lane, three sums per block; k_split_scan
// turns the sums into every block's offsets; the SECOND HALF (2) writes the text.  Neither half waits for another block: in the single kernel a
// block stands still until every block in front of it has published its sizes, and the spread of their arrival times (a few per cent of a
// block's life, amplified by the maximum over the hundreds of blocks in flight) cost 0.8-0.9 of 5.96 ms (profiles/r04_knockouts.txt).
template <int LPP, int OUT, int DT, int NTHR = SIM_THREADS, int WR = 1, int SPLIT = 0>
__global__ void __launch_bounds__(NTHR, (DT == 3 ? (SPLIT == 1 ? DW_IONA_WAVES : SPLIT == 2 ? DW_SIMB_WAVES : DW_IONL_WAVES) : NTHR != SIM_THREADS ? (DT == 1 ? DW_SOLID_WAVES : 1) : DT == 2 ? DW_ION_WAVES : DT == 1 ? DW_SOLID_WAVES : SPLIT == 2 ? DW_SIMB_WAVES : (OUT != 3 || WR != 0) ? DW_SIM_WAVES : DW_SIM_WAVES_BOTH)) k_simulate(SimArgs a)
{
    static_assert(SPLIT == 0 || ((DT == 0 || DT == 3) && NTHR == SIM_THREADS), "the two-kernel form exists for the Illumina variants and for Ion Torrent with its buffers in LDS, 256-lane blocks");
    DW_DYN_SHARED(uint32_t, dyn_lds);                                    // [lds_words][blockDim] packed bases
    __shared__ uint32_t sm_all[4][16];     // one scratch row per scanned quantity: each is written once
    uint32_t (*const sm_bytes)[16] = sm_all, (*const sm_rand)[16] = sm_all + 3, (*const sm_one)[16] = sm_all;      // (the three-look-back forms: bytes in rows 0-2, random reads in row 3)
    __shared__ uint32_t s_ticket, s_slot;
    __shared__ uint64_t s_rbase, s_base[3];
    __shared__ uint32_t s_fixed[2][32];          // "@[prefix_]contig" and "@[prefix_]rand", first 128 bytes
    __shared__ FlowTables s_ft;                  // Ion Torrent: the flow order and its look-up tables (dw_read.hpp fill_flow_tables)
    __shared__ uint32_t s_lg[FLOW_LG_ENTRIES];   // Illumina / SOLiD: the log2 table the gaps between error sites interpolate in (dw_common.hpp geom_gap); Ion Torrent has it in s_ft
    constexpr int nthr = NTHR, PPB = NTHR / LPP, nwaves = NTHR / 64;      // PPB pairs per block
    // where a lane's packed bases are staged: LDS (Illumina / SOLiD reads up to ~1 180 bases, 256-lane blocks), or a SCRATCH SLOT in global memory --
    // the Ion Torrent read buffers, and every read too long for that (the one-wave blocks): staged in LDS a 2 000-base read left room for two waves
    // per CU (half the SIMDs idle, 17 % VALU-active: profiles/r04_variants_pmc.txt) and a 5 000-base read for none.  The slot is read and written
    // word by word in step by the lanes of a wave (word w of lane t at [w * nthr + t]) and lives in the L2 / memory-side cache
    constexpr bool ION = DT == 2 || DT == 3;
    constexpr bool GS = DT == 2 || (DT != 3 && NTHR != SIM_THREADS);
    // ONE LOOK-BACK (Illumina, one kernel).  Rounds 2-5 walked three chains: the random-read counts first (a random read's name ends in its running index: its
    // hexadecimal digits are part of the record lengths), then -- from every block only once it had its index -- the byte totals of the two streams.  Two fronts
    // one behind the other: a block stood still until the second one had passed it.  But the digits are arithmetic (the random reads in front of a block have
    // consecutive indices: hex_digits_sum), and the second stream's total follows from the first's (a pair's two records differ by twice the difference of the read
    // lengths).  So a block publishes ONE word -- its random reads (upper bits) and the bytes of its stream-1 records WITHOUT those digits (lower a.lb_shift bits) --, one chain
    // carries both sums, and everything else is added by the block itself.  One front, two barriers and two block scans fewer.
    // (SOLiD: the BFAST records' lengths are not a function of the BWA records' -- the name counts differ --: their bytes are a second word, on a chain of its own walked
    // by wave 1 at the same time: still one front)
    constexpr bool ONE_LB = DW_ONE_LB != 0 && (DT == 0 || DT == 1) && SPLIT == 0;
    const int LB_SHIFT = a.lb_shift;      // (bits of the byte sum; the host sizes it to the launch and refuses one whose two sums would not fit 62 bits: dw_host.cpp)
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    DW_PROBE_INIT();
    if (GS && tid == 0) { s_slot = scratch_slot_take(a.flow_free, (uint32_t)a.flow_slots, a.n_blocks); asm volatile("" ::: "memory"); }      // (before the ticket: see scratch_slot_take)
#ifndef DW_TICKET_LATE
    if (SPLIT == 0 && tid == 0) s_ticket = (uint32_t)atomicAdd((unsigned long long *)&a.counters[0], 1ull);
#endif
    if (SPLIT != 1) for (int q = tid; q < 32; q += nthr) s_fixed[1][q] = reinterpret_cast<const uint32_t *>(a.rand_fixed)[q];      // buffers are padded to 256 + 16 bytes
    if (ION && tid < 64) s_ft.flow[tid] = a.flow[tid];
    if (!ION && SPLIT != 2) for (int q = tid; q < FLOW_LG_ENTRIES; q += nthr) s_lg[q] = reinterpret_cast<const uint32_t *>(a.flow + 64)[q];
    // base-quality characters of both read ends (dwgsim.c:906-910), packed, behind the lanes' staging area
    const size_t stage_words = SPLIT == 2 ? 0 : DT == 3 ? (size_t)(a.lds_words + a.flow_stack_words) : DT == 2 ? (size_t)a.flow_stack_words : GS ? 0 : (size_t)a.lds_words;      // (the second half of the two-kernel form reads its bases from HBM)
    uint32_t *const s_qb = dyn_lds + stage_words * nthr;
    if (SPLIT != 1) for (int q = tid; q < 2 * a.qb_words; q += nthr) s_qb[q] = (q < a.qb_words ? a.qbase[0] : a.qbase[1])[q < a.qb_words ? q : q - a.qb_words];
    // this lane's text FIFO (record writer), behind the tables
    uint8_t *const s_fifo = reinterpret_cast<uint8_t *>(dyn_lds + (((stage_words * nthr + 2 * (size_t)a.qb_words) + 3) & ~(size_t)3)) + (size_t)tid * (WR == 2 ? SIM_FIFO_BYTES_WIDE : SIM_FIFO_BYTES);
#ifdef DW_TICKET_LATE      // (analysis: the ticket taken behind the staging of the tables instead of in front of it: what stands between a block's ticket and its look-back is on every later block's path)
    if (SPLIT == 0 && tid == 0) s_ticket = (uint32_t)atomicAdd((unsigned long long *)&a.counters[0], 1ull);
#endif
    if (SPLIT == 0) __syncthreads();
    constexpr int H = SPLIT;      // which half of the path this kernel is: 0 both (the single kernel), 1 first, 2 second
    // logical block.  One kernel: from an atomic ticket, so that a block's predecessors have started when it looks back at them.  Two kernels:
    // no block waits for another, the block index will do
    const uint32_t t = SPLIT == 0 ? uniform_u32(s_ticket) : (uint32_t)blockIdx.x;
    // the read-index range (contig, first index, count) this block belongs to: block-uniform, its fields live in scalar registers
    const SegPtr sg = as_constant(a.segs) + seg_of_block(as_constant(a.segs), a.n_seg, t);
    const SegCtx sc = seg_ctx(a, sg);
    const uint8_t *name_fixed = a.names + sg->name_off;
    if (H != 1) for (int q = tid; q < 32; q += nthr) s_fixed[0][q] = reinterpret_cast<const uint32_t *>(name_fixed)[q];      // (read after later barriers only)
    if (ION) { if (SPLIT != 0) __syncthreads(); fill_flow_tables(s_ft, a.flow_len, tid, nthr, a.flow); __syncthreads(); }      // (the flow order is in LDS: the single kernel's barrier above, or this one)
    const int j = (LPP == 2) ? (tid & 1) : 0;
    const uint64_t pair_in = (uint64_t)(t - sg->first_block) * PPB + (uint64_t)(tid / LPP);      // inside the range
    const bool valid = pair_in < sg->n_pairs;
    const uint64_t ii = sg->first_ii + pair_in;
    const uint64_t pair = sg->pair_off + pair_in;                 // inside the launch
    const RngKey key{a.p.seed, uniform_u32(sg->contig_index)};
    const int s = sel_len(a, j);
    // this lane's packed bases: word w at lds[w * nthr].  Illumina / SOLiD: 4 bits per base; Ion Torrent: the one in-place buffer of the flow model,
    // 2 bits per base, a.lds_words words for a.cap = 16 a.lds_words bases (dw_read.hpp flow_errors), in LDS (DT = 3) or in a scratch slot (DT = 2);
    // the run stack of its pass 2 (a.flow_stack_words words per lane) is in LDS either way
    // (second half of the two-kernel form: the tile's staged bases where the first half left them, read once, in batches of eight words)
    uint32_t *lds = GS ? a.flow_scratch + (size_t)uniform_u32(s_slot) * ((size_t)(DT == 2 ? flow_words_per_lane(a.lds_words) : a.lds_words) * nthr) + tid
                  : SPLIT == 2 ? a.split_state + (size_t)t * ((size_t)a.lds_words * nthr) + tid : dyn_lds + tid;
    uint32_t *const flow_stk = dyn_lds + (DT == 3 ? (size_t)a.lds_words * nthr : 0) + tid;
    const int capb = 16 * a.lds_words;              // (Ion Torrent) bases the buffer holds

    // A block's records can be placed only when every block in front of it has published its sizes: what a wave does BEFORE it publishes its own is on
    // the critical path of all the blocks behind it, what it does after (text assembly) is not.  So a wave runs at raised issue priority up to its
    // look-backs and at the normal one from there: 2 x 150 bp 5.98 -> 5.82 ms, 2 x 100 bp 7.14 -> 6.82 (profiles/r04_split.txt; equal priorities
    // throughout, or the other way round, change nothing)
    if (SPLIT == 0) wave_priority(1);
    DW_PROBE_MARK(a, 0);     // ticket, fixed strings
    // ---- attempts until the pair is accepted (dwgsim.c:649-843): placement, haplotype, strands, base extraction of this
    // read end, N filter; the two lanes of a pair exchange their verdicts and retry together with attempt + 1 ----
    PairDraw pd; pd.is_rand = true; pd.pos = pd.d = 0; pd.hap = 0; pd.strand0 = pd.strand1 = 0;
    ReadRes rr{0, 0, 0, 0, 0};
    uint32_t att = 0; bool is_rand = false, done = !valid;
    int32_t n_err = 0;
    int s_out = s;                              // read length after errors (changes only for Ion Torrent)
    bool flow_reversed = false;
    const int nw = (s + 7) >> 3;
    int32_t err_first = 0;                      // SOLiD: an error on the first colour (n_err_first, dwgsim.c:240)
    uint32_t rrank = 0, rtot = 0;
    uint32_t rb[4] = {0, 0, 0, 0};              // the words of the Philox block in use (error sites / random-read bases)
    if (H != 2) {
    while (__ballot(!done)) {
        bool ok = true;
        if (!done) {
            pd = draw_pair(a, sc, key, ii, att);
            if (pd.is_rand) { is_rand = true; done = true; rr = ReadRes{0, 0, 0, 0, 0}; }
            else if (s > 0) {
                int64_t start; int step;
                read_geom(a, sc, pd, j, &start, &step);
                DW_PROBE_MARK(a, 7);     // placement draws (phase 1 below is then the base extraction alone)
                if (ION) {        // straight into the flow model's buffer: 2 bits per base, N as A, a reverse-strand read turned round (dw_read.hpp FlowSink)
                    const bool rev = (j ? pd.strand1 : pd.strand0) != 0;
                    rr = gen_read(sel_hap(a, sc, pd.hap), sc.l, start, step, s, rev ? 1 : 0, FlowSink{lds, nthr, rev ? (capb >> 3) - 1 : ((capb - s) & ~7) >> 3, rev});
                }
                else if (probe::off(8)) { rr = ReadRes{(int32_t)start, 0, 0, 0, 0}; for (int w = 0; w * 8 < s; ++w) lds[w * nthr] = 0x32103210u; }
                else
                rr = gen_read(sel_hap(a, sc, pd.hap), sc.l, start, step, s, j ? pd.strand1 : pd.strand0, WordSink{lds, nthr});
                ok = rr.ext_coor >= 0 && rr.num_n <= a.p.max_n;
            }
        }
        if (LPP == 2) { const int other = __shfl_xor((int)ok, 1); ok = ok && (other != 0); }   // every lane shuffles
        if (!done) {
            if (ok) done = true;
            else if (++att > (uint32_t)MAX_ATTEMPTS) { atomicOr((unsigned long long *)&a.counters[2], 1ull); done = true; }
        }
    }
    DW_PROBE_MARKF(a, 5);    // (fine) the extraction of the last attempt, leaving the loop
    // failed attempts and outcome of the pair, and whether it opens its contig: input of the abort rule (k_failrule)
    if (valid && j == 0) a.meta[pair] = att | (is_rand ? 0x80000000u : 0u) | ((sg->contig_start && pair_in == 0) ? 0x40000000u : 0u);
    { const uint32_t retries = wave_sum_u32((valid && j == 0) ? att : 0u); if (lane == 0 && retries) atomicAdd((unsigned long long *)&a.counters[1], (unsigned long long)retries); }
    // running random-read index (dwgsim.c:1042,1096): look-back over the blocks' random counts + rank inside the block
    if (H == 0 && !ONE_LB) {
    { const uint32_t v[1] = {(is_rand && j == 0) ? 1u : 0u}; uint32_t ex[1], tot[1]; block_excl_scan_n<1>(v, sm_rand, ex, tot); rrank = ex[0]; rtot = tot[0]; }
    DW_PROBE_MARKF(a, 6);    // (fine) the block scan of the random reads
    if (wave == 0) {
        const uint64_t g = probe::off(128) ? (uint64_t)t * 6 : lookback_excl(a.status[2], t, rtot, 0);
        if (lane == 0) { s_rbase = g; if (t + 1 == a.n_blocks) a.counters[3] = g + rtot; }
    }
    }
    // (the barrier that publishes s_rbase comes after the error phase, which does not need the index: the look-back's latency
    // overlaps with that work instead of idling three waves)
    DW_PROBE_MARK(a, 1);     // placement + base extraction
    // ---- sequencing errors (dwgsim.c:233-244) or random bases (dwgsim.c:999-1001) ----
    // 16-bit draws: one Philox block tests eight bases (the low half of a uniform is drawn lazily, see below); an error marks bit 3 of
    // the base's nibble and its substituted base is drawn afterwards, only for the (few) marked bases
    if (ION) {                                  // dwgsim.c:861-864; every lane calls (the loops of the model are wave-uniform)
        const bool flows = valid && !is_rand && s > 0;
        FlowRng rg; rg.seed = key.seed; rg.contig = key.contig; rg.dom = D_FLOW0 + (uint32_t)j; rg.att = att; rg.evt = 0; rg.s = 0; rg.ii = ii; rg.w0 = rg.w1 = rg.w2 = rg.w3 = 0;
        const int so = flow_errors(flows, rg, s_ft, a.flow_len, (j ? a.e_thr[1] : a.e_thr[0])[0], j ? a.flow_gap_r[1] : a.flow_gap_r[0], j ? a.flow_gap_s[1] : a.flow_gap_s[0], lds, flow_stk, nthr, 2 * a.flow_stack_words, s, j ? pd.strand1 : pd.strand0, capb, &n_err);
        if (flows) {
            s_out = so;
            if (s_out < 0) { atomicOr((unsigned long long *)&a.counters[2], 2ull); s_out = 0; }
            flow_reversed = (j ? pd.strand1 : pd.strand0) != 0;     // the read is turned back while it is written (dwgsim.c:408-414)
        }
    }
    if (ION) {                                  // a random read (dwgsim.c:999-1001): base i = the 2-bit field i of the D_BASE0 stream (64 bases per Philox block), at positions 0 .. s - 1 of the buffer
        if (valid && is_rand) for (int w = 0; w * 16 < s; ++w) {
            if ((w & 3) == 0) { const U4 q0 = rng_block(key, D_BASE0 + (uint32_t)j, ii, att, 0, (uint32_t)(w >> 2)); rb[0] = q0.x; rb[1] = q0.y; rb[2] = q0.z; rb[3] = q0.w; }
            uint32_t pairs = (w & 2) ? ((w & 1) ? rb[3] : rb[2]) : ((w & 1) ? rb[1] : rb[0]);
            const int rem = s - 16 * w;
            if (rem < 16) pairs &= (1u << (2 * rem)) - 1u;
            lds[w * nthr] = pairs;
        }
    } else
    if (valid && !(probe::off(4))) {
        if (is_rand) {                           // random read: base i = the 2-bit field i of the D_BASE0 stream: sixteen bits = the eight bases of a staged word
            uint32_t prev_base = 0;
            for (int w = 0; w < nw; ++w) {
                if ((w & 7) == 0) { const U4 q0 = rng_block(key, D_BASE0 + (uint32_t)j, ii, att, 0, (uint32_t)(w >> 3)); rb[0] = q0.x; rb[1] = q0.y; rb[2] = q0.z; rb[3] = q0.w; }
                const uint32_t wd = ((w >> 1) & 2) ? (((w >> 1) & 1) ? rb[3] : rb[2]) : (((w >> 1) & 1) ? rb[1] : rb[0]);
                uint32_t word = pairs_to_nibbles((wd >> (16 * (w & 1))) & 0xFFFFu);
                const int rem = s - 8 * w;
                const uint32_t live = rem >= 8 ? 0xFFFFFFFFu : ((1u << (4 * rem)) - 1u);
                if (DT == 1) {                                                      // colour = __gf_add(previous base, base): dwgsim.h:6, dwgsim.c:1022-1032
                    const uint32_t prevw = (word << 4) | prev_base;
                    prev_base = word >> 28;
                    word = (word ^ prevw) & 0x33333333u;
                }
                lds[w * nthr] = word & live;
            }
        } else {
            if (DT == 1) {                       // SOLiD: bases -> colours (dwgsim.c:845-858): colour = previous base ^ base, 4 if either is not ACGT; the adaptor counts as 'A' (:849)
                uint32_t prev_base = 0;
                for (int w = 0; w < nw; ++w) {
                    uint32_t word = lds[w * nthr];
                    const uint32_t prevw = (word << 4) | prev_base;
                    prev_base = word >> 28;
                    const uint32_t n = (word | prevw) & 0x44444444u;
                    word = ((word ^ prevw) & 0x33333333u & ~((n >> 1) | (n >> 2))) | n;
                    const int rem = s - 8 * w;
                    lds[w * nthr] = rem >= 8 ? word : word & ((1u << (4 * rem)) - 1u);
                }
            }
            // error sites (dwgsim.c:237 `drand48() < e[i]`, once per base that is not N): the gap chain of the read end at the largest rate of its ramp, thinned
            // where the position's own rate is lower (dw_common.hpp D_BASE0 / D_BASE_REF0): a Philox block per FOUR errors -- rounds 2-5: per eight BASES.
            // The loop's trip count differs from lane to lane (a read end has ~3 errors); its gap index m does not, so the lanes draw their Philox blocks together
            const uint64_t tmax = j ? a.err_thr_max[1] : a.err_thr_max[0], gR = j ? a.err_gap_r[1] : a.err_gap_r[0];
            const int gS = j ? a.err_gap_s[1] : a.err_gap_s[0]; const bool ramp = (j ? a.err_ramp[1] : a.err_ramp[0]) != 0;
            const uint64_t *thr64 = j ? a.e_thr[1] : a.e_thr[0];
            if (tmax) {
                uint32_t m = 0, S = 0, tw[4] = {0, 0, 0, 0}, sw[4] = {0, 0, 0, 0};
                for (;;) {                                                                  // (a per-lane loop: no wave operation inside)
                    if ((m & 3u) == 0u) {
                        const U4 q0 = rng_block(key, D_BASE0 + (uint32_t)j, ii, att, 0, m >> 2); rb[0] = q0.x; rb[1] = q0.y; rb[2] = q0.z; rb[3] = q0.w;
                        if (ramp) { const U4 q1 = rng_block(key, D_BASE_REF0 + (uint32_t)j, ii, att, 0, m >> 2); tw[0] = q1.x; tw[1] = q1.y; tw[2] = q1.z; tw[3] = q1.w; }
                    }
                    const uint32_t k4 = m & 3u;
                    const uint32_t gw = (k4 & 2u) ? ((k4 & 1u) ? rb[3] : rb[2]) : ((k4 & 1u) ? rb[1] : rb[0]);
                    const uint32_t G = tmax >= 0x100000000ull ? 0u : geom_gap(gw, s_lg, gR, gS);
                    S = m ? S + 1u + G : G;                                                 // site m of the chain
                    if (S >= (uint32_t)s) break;
                    // (the substitution draws of sites 4 q .. 4 q + 3: one block, drawn once a site of the four is inside the read -- rounds 2-5 and the
                    // first form of round 6 drew a block per ERROR, in a loop of its own whose trip count was the wave's largest error count)
                    if (k4 == 0u) { const U4 q2 = rng_block(key, D_SUB0 + (uint32_t)j, ii, att, 0, m >> 2); sw[0] = q2.x; sw[1] = q2.y; sw[2] = q2.z; sw[3] = q2.w; }
                    const int w = (int)(S >> 3), sh = 4 * (int)(S & 7u);
                    const uint32_t word = lds[w * nthr];
                    bool hit = ((word >> sh) & 4u) == 0u;                                    // N bases / colour 4 take no error
                    if (ramp && hit) {                                                      // thinning: kept with probability thr[S] / thr_max (site m's own word)
                        const uint64_t ti = thr64[S];
                        const uint32_t w2 = (k4 & 2u) ? ((k4 & 1u) ? tw[3] : tw[2]) : ((k4 & 1u) ? tw[1] : tw[0]);
                        hit = ti >= tmax || (uint64_t)w2 * tmax < (ti << 32);
                    }
                    if (hit) {                                                              // dwgsim.c:238: c = (c + (int)(drand48() * 3.0 + 1)) & 3
                        const uint32_t rwd = (k4 & 2u) ? ((k4 & 1u) ? sw[3] : sw[2]) : ((k4 & 1u) ? sw[1] : sw[0]);
                        const uint32_t add = 1u + (uint32_t)(((uint64_t)rwd * 3u) >> 32);      // (int)(u * 3.0 + 1), exact
                        const uint32_t c = (((word >> sh) & 3u) + add) & 3u;
                        lds[w * nthr] = (word & ~(0xFu << sh)) | (c << sh);
                        ++n_err; if (DT == 1 && S == 0u) err_first = 1;
                    }
                    ++m;
                }
            }
        }
    }
    }      // H != 2: placement, extraction, errors
    // ---- the second half of the two-kernel form picks up what the first half left: 16 bytes of name fields per lane and the staged bases ----
    // the random reads and the bytes of stream 1 / 2 in front of this block: inside its chunk of 1024 blocks + in front of the chunk (k_split_scan1 / 2)
    // + the hexadecimal digits that the random reads in front of it add to their names (one record per random pair in each stream)
    uint64_t pre_rand = 0, pre_b1 = 0, pre_b2 = 0;
    if (SPLIT == 2) {
        const uint64_t *pi = a.split_pre + 4 * (size_t)t, *pc = a.split_chunk + 4 * (size_t)(t >> 10);
        pre_rand = pi[0] + pc[0];
        const uint64_t hexb = hex_digits_sum(a.chain[0], pre_rand);
        pre_b1 = pi[1] + pc[1] + hexb; pre_b2 = pi[2] + pc[2] + (LPP == 2 ? hexb : 0ull);
    }
    // (Ion Torrent: a read the flow model gave up on emits nothing, which the arithmetic offsets of the two-kernel form do not allow for -- and the batch
    // is run again or fails as a whole (dw_host.cpp dwgsim_hip_wait): nothing is written for it)
    if (ION && H == 2 && (a.counters[2] & 2ull)) return;
    if (H == 2) {
        const uint4 hm = reinterpret_cast<const uint4 *>(a.split_hand)[(size_t)t * nthr + tid];
        rr.ext_coor = (int32_t)hm.x; n_err = (int32_t)(hm.y & 0xffffu); rr.n_sub = (int32_t)(hm.y >> 16); rr.n_indel = (int32_t)(hm.z & 0xffffu); rr.n_ins = (int32_t)(hm.z >> 16);
        att = hm.w & 0x3fffu; is_rand = (hm.w >> 14) & 1u; pd.strand0 = (int)((hm.w >> 15) & 1u); pd.strand1 = (int)((hm.w >> 16) & 1u);
        if (ION) { s_out = (int)(hm.w >> 17); flow_reversed = !is_rand && (j ? pd.strand1 : pd.strand0) != 0; }
        __syncthreads();      // the tables of the prologue (names, base qualities)
        { const uint32_t v[1] = {(is_rand && j == 0) ? 1u : 0u}; uint32_t ex[1], tot[1]; block_excl_scan_n<1>(v, sm_rand, ex, tot); rrank = ex[0]; rtot = tot[0]; }
    }
    DW_PROBE_MARKF(a, 2);    // (fine) the error phase's own work
    if (H == 0 && !ONE_LB) __syncthreads();
    // (the first half does not know the running index yet: its lengths leave the hexadecimal digits of random reads' names out, k_split_scan adds them)
    uint64_t rand_ii = (H == 1 || ONE_LB) ? 0 : a.chain[0] + (SPLIT == 2 ? pre_rand : s_rbase) + rrank - ((LPP == 2 && j == 1 && is_rand) ? 1u : 0u);   // odd lane: its even partner was counted
    DW_PROBE_MARK(a, 2);     // error tests + substitutions
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
    if (is_rand) { fixed_len = (uint32_t)a.rand_fixed_len; tail_len = tail_len_w = 25u + ((H == 1 || ONE_LB) ? 0u : ndigits16(rand_ii)); }      // (ONE_LB: the digits are added below)   // "_0_0_0_0_1_1_0:0:0_0:0:0_" (25 chars) + hex
    else {
        fixed_len = (uint32_t)sg->name_fixed_len;
        tail_len = pair_tail_len(x0, x1, nc, ii);
        tail_len_w = (DT == 1) ? pair_tail_len(x0, x1, ncw, ii) : tail_len;
    }
    const bool emits = valid && s_out > 0;
    // record lengths.  SOLiD: BWA drops the first colour and its quality (dwgsim.c:950-955); BFAST prepends the adaptor 'A' (:968-975)
    const uint32_t Lbwa = !emits ? 0u : (DT == 1) ? (1u + fixed_len + tail_len_w + 2u + 1u + 2u * (uint32_t)(s_out - 1) + 3u + 1u)
                                                  : (1u + fixed_len + tail_len + 2u + 1u + (uint32_t)s_out + 3u + (uint32_t)s_out + 1u);

    // ---- record offsets: block scan + decoupled look-back over logical blocks ----
    // The BFAST offsets get their own scan + look-back when a BFAST record is not simply "its BWA record minus the 2-byte suffix" for
    // every lane: SOLiD (different lengths), Ion Torrent (a read the flow model gave up on emits nothing, and must not shift the others)
    // (the two-kernel form of Ion Torrent takes the arithmetic offsets: a batch with a read the model gave up on is run again as a whole, dw_host.cpp)
    constexpr bool BF_SCAN = DT != 0 && SPLIT == 0;
    uint32_t e1, e2, eb = 0, T1, T2, Tb = 0;
    uint64_t G1_one = 0, G2_one = 0, Gb_one = 0;
    if (ONE_LB) {
        // one scan (random reads, bytes of stream 1 / 2 without the digits), one word published, one look-back
        constexpr bool BF_ONE = DT == 1;      // (SOLiD: the BFAST stream's own sum)
        constexpr int NS = BF_ONE ? 4 : 3;
        const uint32_t Lbf1 = (BF_ONE && emits) ? (1u + fixed_len + tail_len + 1u + 1u + 2u * (uint32_t)s_out + 3u + 1u) : 0u;
        uint32_t v[NS], ex[NS], tot[NS];
        v[0] = (is_rand && j == 0) ? 1u : 0u; v[1] = j == 0 ? Lbwa : 0u; v[2] = j == 1 ? Lbwa : 0u; if (BF_ONE) v[NS - 1] = Lbf1;
        block_excl_scan_n<NS>(v, sm_one, ex, tot);
        rrank = ex[0]; e1 = ex[1]; e2 = ex[2]; rtot = tot[0]; if (BF_ONE) eb = ex[NS - 1];
        DW_PROBE_MARKF(a, 6);
        if (wave == 0) {
            const uint64_t g = lookback_excl(a.status[0], t, ((uint64_t)tot[0] << LB_SHIFT) | (uint64_t)tot[1], 0);
            if (lane == 0) s_base[0] = g;
        }
        if (BF_ONE && wave == (nwaves > 1 ? 1 : 0)) { const uint64_t gb = lookback_excl(a.status[3], t, (uint64_t)tot[NS - 1], 0); if (lane == 0) s_base[2] = gb; }
        __syncthreads();
        const uint64_t pk = s_base[0];
        const uint64_t base_r = pk >> LB_SHIFT, pre1 = pk & ((1ull << LB_SHIFT) - 1ull);
        const uint64_t rb0 = a.chain[0] + base_r;                                  // the index of the block's first random read
        const uint64_t hex_before = hex_digits_sum(a.chain[0], base_r);           // the digits of all the random reads in front of the block (block-uniform)
        const uint64_t pairs_before = sg->pair_off + (uint64_t)(t - sg->first_block) * PPB;      // every one of them has its record(s): Illumina reads keep their length
        G1_one = pre1 + hex_before;
        G2_one = LPP == 2 ? pre1 + (uint64_t)((int64_t)2 * (a.p.len[1] - a.p.len[0]) * (int64_t)pairs_before) + hex_before : 0ull;
        // this lane: its index, its name's length with the digits, and the digits of the block's random reads in front of its record in either stream
        const uint32_t cnt2 = rrank - ((LPP == 2 && j == 1 && is_rand) ? 1u : 0u);      // random pairs in front of this lane's pair
        rand_ii = rb0 + cnt2;
        if (is_rand) tail_len = tail_len_w = 25u + ndigits16(rand_ii);
        const uint32_t dlo = ndigits16(rb0);
        const bool one_width = dlo == ndigits16(rb0 + rtot);                       // (block-uniform; all but a handful of blocks of a launch)
        const uint32_t h1 = one_width ? dlo * rrank : (uint32_t)hex_digits_sum(rb0, rrank), h2 = one_width ? dlo * cnt2 : (uint32_t)hex_digits_sum(rb0, cnt2);
        const uint32_t htot = one_width ? dlo * rtot : (uint32_t)hex_digits_sum(rb0, rtot);
        e1 += h1; if (LPP == 2) e2 += h2;      // (single end: no second stream)
        T1 = tot[1] + htot; T2 = tot[2] + (LPP == 2 ? htot : 0u);
        if (BF_ONE) {      // both ends' records of a random pair carry the index
            eb += LPP == 2 ? h1 + h2 : h1; Tb = tot[NS - 1] + (LPP == 2 ? 2u : 1u) * htot;
            Gb_one = s_base[2] + (LPP == 2 ? 2ull : 1ull) * hex_before;
        }
        if (tid == nthr - 1 && t + 1 == a.n_blocks) a.counters[3] = base_r + rtot;
    } else
    if (BF_SCAN) {
        const uint32_t Lbf = !emits ? 0u : DT == 1 ? (1u + fixed_len + tail_len + 1u + 1u + 2u * (uint32_t)s_out + 3u + 1u) : Lbwa - 2u;
        const uint32_t v[3] = {j == 0 ? Lbwa : 0u, j == 1 ? Lbwa : 0u, Lbf};
        uint32_t ex[3], tot[3]; block_excl_scan_n<3>(v, sm_bytes, ex, tot);
        e1 = ex[0]; e2 = ex[1]; eb = ex[2]; T1 = tot[0]; T2 = tot[1]; Tb = tot[2];
    } else if (H != 1) {
        const uint32_t v[2] = {j == 0 ? Lbwa : 0u, j == 1 ? Lbwa : 0u};
        uint32_t ex[2], tot[2]; block_excl_scan_n<2>(v, sm_bytes, ex, tot);
        e1 = ex[0]; e2 = ex[1]; T1 = tot[0]; T2 = tot[1];
    } else { e1 = e2 = T1 = T2 = 0; }
    if (H == 1) {
        // ---- end of the first half: the block's sums (random pairs, bytes of stream 1 / 2 without the random reads' hexadecimal digits), this
        // lane's name fields and its staged bases go to HBM, each array laid out so that a wave's accesses are contiguous ----
        const uint32_t rsum = wave_sum_u32((is_rand && j == 0) ? 1u : 0u), b1 = wave_sum_u32(j == 0 ? Lbwa : 0u), b2 = wave_sum_u32(j == 1 ? Lbwa : 0u);
        if (lane == 0) { sm_rand[0][wave] = rsum; sm_bytes[0][wave] = b1; sm_bytes[1][wave] = b2; }
        const uint4 hm = make_uint4((uint32_t)rr.ext_coor, (uint32_t)n_err | ((uint32_t)rr.n_sub << 16), (uint32_t)rr.n_indel | ((uint32_t)rr.n_ins << 16),
                                    att | (is_rand ? 1u << 14 : 0u) | ((uint32_t)pd.strand0 << 15) | ((uint32_t)pd.strand1 << 16) | (ION ? (uint32_t)s_out << 17 : 0u));      // (Ion Torrent: the read's length after errors, < 2^15: the host checks the capacity)
        reinterpret_cast<uint4 *>(a.split_hand)[(size_t)t * nthr + tid] = hm;
        uint32_t *gs = a.split_state + (size_t)t * ((size_t)a.lds_words * nthr) + tid;
        if (valid) for (int w = 0, n_w = ION ? (s_out + 15) >> 4 : nw; w < n_w; ++w) gs[(size_t)w * nthr] = lds[w * nthr];
        __syncthreads();
        if (tid == 0) { uint32_t r = 0, t1 = 0, t2 = 0; for (int w = 0; w < nwaves; ++w) { r += sm_rand[0][w]; t1 += sm_bytes[0][w]; t2 += sm_bytes[1][w]; } reinterpret_cast<uint4 *>(a.split_agg)[t] = make_uint4(r, t1, t2, 0u); }
        return;
    }
    if (H == 0 && !ONE_LB) {
    if (wave == 0) {
        const uint64_t g = probe::off(128) ? (uint64_t)t * PPB * (uint64_t)(60 + 2 * s + a.rand_fixed_len + sg->name_fixed_len) : lookback_excl(a.status[0], t, T1, 0); if (lane == 0) s_base[0] = g;
        if (BF_SCAN) { const uint64_t gb = lookback_excl(a.status[3], t, Tb, 0); if (lane == 0) s_base[2] = gb; }
    }
    if (wave == (nwaves > 1 ? 1 : 0)) { const uint64_t g = probe::off(128) ? (uint64_t)t * PPB * (uint64_t)(60 + 2 * s + a.rand_fixed_len + sg->name_fixed_len) : lookback_excl(a.status[1], t, T2, 0); if (lane == 0) s_base[1] = g; }
    __syncthreads();
    }
    const uint64_t G1 = ONE_LB ? G1_one : SPLIT == 2 ? pre_b1 : s_base[0], G2 = ONE_LB ? G2_one : SPLIT == 2 ? pre_b2 : s_base[1];
    const uint64_t reads_before_block = (sg->pair_off + (uint64_t)(t - sg->first_block) * PPB) * (uint64_t)LPP;      // (every block in front of a range's last is full)
    const uint64_t nvalid_before = (uint64_t)tid;                  // valid lanes form a prefix of the block
    const uint64_t off_bwa = (j == 0) ? G1 + e1 : G2 + e2;
    // Illumina: a BFAST record is its BWA record minus the 2-byte "/1" suffix, so its offset follows from the two BWA scans
    const uint64_t off_bf = BF_SCAN ? (ONE_LB ? Gb_one : s_base[2]) + eb : G1 + G2 - 2 * reads_before_block + e1 + e2 - 2 * nvalid_before;
    if (H == 0 && tid == nthr - 1) {
        if (t + 1 == a.n_blocks) {
            const uint64_t nreads = a.n_pairs * (uint64_t)((a.p.len[1] > 0) ? 2 : 1);
            a.counters[4] = a.p.has_bwa ? G1 + T1 : 0;
            a.counters[5] = a.p.has_bwa ? G2 + T2 : 0;
            a.counters[6] = !a.p.has_bfast ? 0 : BF_SCAN ? (ONE_LB ? Gb_one : s_base[2]) + Tb : G1 + T1 + G2 + T2 - 2 * nreads;
        }
    }

    if (SPLIT == 0 && (DW_PRIO_DROP == 0 || DT == 1)) wave_priority(0);
    DW_PROBE_MARK(a, 3);     // name lengths, block scan, look-back
    // ---- SOLiD records (dwgsim.c:934-976, :1056-1094): the two outputs differ in name counts, suffix, alphabet and length ----
    if (DT == 1) {
        if (emits) {
            for (int which = 0; which < 2; ++which) {            // 0: BWA stream of this end, 1: BFAST
                if (!(OUT & (1 << which))) continue;
                Out2<1, WR> o;
                o.init(s_fifo, which ? a.out[2] + off_bf : (j ? a.out[1] : a.out[0]) + off_bwa, nullptr);
                put_name_fixed(o, is_rand ? s_fixed[1] : s_fixed[0], is_rand ? a.rand_fixed : name_fixed, fixed_len);
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
                for_each_quality_block(a.p, key, D_QUAL0 + (uint32_t)j, ii, att, s_qb + (j ? a.qb_words : 0), s, s_out, first != 0,
                                       [&](uint64_t blk, uint32_t nb) { o.putn(blk, nb); });       // same draws for both outputs
                o.put('\n');
                o.flush();
            }
        }
    } else
    // ---- write the record(s): every lane walks through (a lane without a record writes nothing: the quality blocks are drawn in a
    // wave-uniform loop) ----
    {
        const bool rec = valid && s_out > 0;
        Out2<OUT, WR> o;
        if (rec) {
        o.init(s_fifo, (OUT & 1) ? (j ? a.out[1] : a.out[0]) + off_bwa : nullptr, (OUT & 2) ? a.out[2] + off_bf : nullptr);
        if (!(probe::off(16))) {
        put_name_fixed(o, is_rand ? s_fixed[1] : s_fixed[0], is_rand ? a.rand_fixed : name_fixed, fixed_len);
        if (is_rand) put_rand_tail(o, rand_ii);
        else put_pair_tail(o, x0, x1, pd.strand0, pd.strand1, nc, ii);
        }
        o.put_suffix((uint64_t)'/' | ((uint64_t)('1' + j) << 8) | ((uint64_t)'\n' << 16), 3, (uint64_t)'\n', 1);
        if (SPLIT == 0 && DW_PRIO_DROP == 1) wave_priority(0);      // (the name line still at the raised priority: see DW_PRIO_DROP)
        DW_PROBE_MARK(a, 4); // header line
        // bases (the second writer of -o 0 starts a new section, so that sixteen bases are one store)
        o.rebase();
        // word w of the record (bases 8w .. 8w + 7) as nibbles.  Ion Torrent: from the 2-bit buffer, where the read stands at positions 0 .. s_out - 1 in the
        // orientation of the flow model; reverse strand: base i of the record = base s_out-1-i of it (dwgsim.c:408-414): the eight positions ending at
        // s_out-1-8w, fetched together and turned round
        auto rec_word = [&](int w) -> uint32_t {
            if (!ION) return lds[w * nthr];
            const Buf2 B{lds, nthr, a.lds_words};
            if (!flow_reversed) return pairs_to_nibbles(B.get8(8 * w));
            const int lo = s_out - 8 - 8 * w;                   // first position of the window (negative in the last, partial word)
            return pairs_to_nibbles(lo < 0 ? reverse_pairs16(B.get8(0)) >> (2 * (-lo)) : reverse_pairs16(B.get8(lo)));
        };
        auto put_last_word = [&](uint32_t word, int rem) __attribute__((always_inline)) {      // fewer than sixteen bases left: this word's share of them
            if (rem >= 8) { o.put4(base_chars4(word)); o.put4(base_chars4(word >> 16)); }
            else { const uint32_t c0 = base_chars4(word), c1 = base_chars4(word >> 16); for (int b = 0; b < rem; ++b) o.put(((b < 4 ? c0 : c1) >> (8 * (b & 3))) & 0xff); }
        };
        if (SPLIT == 2 && !ION) {      // the bases come from HBM: eight words (64 bases) fetched together, then written
            const int full16 = s_out >> 4;                       // whole groups of sixteen bases
            for (int g0 = 0; g0 < full16; g0 += 4) {
                uint32_t r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = (g0 + (k >> 1) < full16) ? lds[(size_t)(2 * g0 + k) * nthr] : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (g0 + q < full16) o.put16(base_chars4(r[2 * q]), base_chars4(r[2 * q] >> 16), base_chars4(r[2 * q + 1]), base_chars4(r[2 * q + 1] >> 16));
            }
            if (s_out & 15) {                                     // the last, shorter group
                const int w = 2 * full16, rem = s_out - 8 * w;
                const uint32_t wa = lds[(size_t)w * nthr], wb = rem > 8 ? lds[(size_t)(w + 1) * nthr] : 0u;
                put_last_word(wa, rem);
                if (rem > 8) put_last_word(wb, rem - 8);
            }
        } else if (ION) {
            // the 2-bit buffer (LDS, a scratch slot, or -- second half of the two-kernel form -- HBM): the words of SIXTY-FOUR bases fetched together, then
            // written (round 5 fetched a word, or two, per eight bases and waited for each: in the second half, whose buffer is in HBM, the sequence line was
            // a chain of a hundred dependent loads per read).  Reverse strand: record group j = buffer positions [s_out - 16 (j + 1), s_out - 16 j) turned round
            const Buf2 B{lds, nthr, a.lds_words};
            const int full16 = s_out >> 4;
            const uint32_t sh2 = ((uint32_t)s_out & 15u) * 2u;
            for (int g0 = 0; g0 < full16; g0 += 4) {
                uint32_t r[5];
                const int base = flow_reversed ? ((s_out - 16 * g0) >> 4) - 4 : g0;
#pragma unroll
                for (int q = 0; q < 5; ++q) r[q] = (base + q >= 0 && (flow_reversed || q < 4)) ? B.word(base + q) : 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (g0 + k < full16) {
                    uint32_t v;
                    if (!flow_reversed) v = r[k];
                    else {
                        v = sh2 ? __builtin_amdgcn_alignbit(r[4 - k], r[3 - k], sh2) : r[3 - k];
                        v = __builtin_bitreverse32(v); v = ((v & 0x55555555u) << 1) | ((v >> 1) & 0x55555555u);      // the sixteen pairs in reverse order
                    }
                    const uint32_t n0 = pairs_to_nibbles(v & 0xFFFFu), n1 = pairs_to_nibbles(v >> 16);
                    o.put16(base_chars4(n0), base_chars4(n0 >> 16), base_chars4(n1), base_chars4(n1 >> 16));
                }
            }
            for (int w = 2 * full16; w * 8 < s_out; ++w) put_last_word(rec_word(w), s_out - w * 8);      // the last, shorter group
        } else {
        int w = 0;
        for (; (w + 2) * 8 <= s_out; w += 2) {
            const uint32_t w0 = rec_word(w), w1 = rec_word(w + 1);
            o.put16(base_chars4(w0), base_chars4(w0 >> 16), base_chars4(w1), base_chars4(w1 >> 16));
        }
        for (; w * 8 < s_out; ++w) put_last_word(rec_word(w), s_out - w * 8);
        }
        o.put('\n'); o.put('+'); o.put('\n');
        o.rebase();
        }
        if (SPLIT == 0 && DW_PRIO_DROP == 1) wave_priority(0);      // (a wave none of whose lanes has a record)
        DW_PROBE_MARK(a, 5); // sequence line
        // qualities (dwgsim.c:899-918): up to eight characters per Philox block of the read end's try stream, appended as they come
        if (rec) {
            if constexpr (WR != 0 && DW_QUAL_FIFO) quality_line_fifo(o.a, a.p, key, D_QUAL0 + (uint32_t)j, ii, att, s_qb + (j ? a.qb_words : 0), s, s_out);
            else for_each_quality_block(a.p, key, D_QUAL0 + (uint32_t)j, ii, att, s_qb + (j ? a.qb_words : 0), s, s_out, false,
                                        [&](uint64_t blk, uint32_t nb) { o.putn(blk, nb); });
        }
        if (rec) { o.put('\n'); o.flush(); }
    }
    DW_PROBE_MARK(a, 6);     // quality line
    if (GS) {                // the scratch slot goes back to this XCD's free list once every wave's accesses to it are done
        wait_stores();
        __syncthreads();
        if (tid == 0) scratch_slot_release(a.flow_free, a.n_blocks, s_slot);
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (declared in dw_launch.hpp)
// ------------------------------------------------------------------------------------------------
#if DW_HAS(0)
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
        {   // the placement's insert-size normal (draw_pair): radii with full 53-bit operands
            const double p1 = 2.0 * u53(c.x, c.y) - 1.0, p2 = 2.0 * u53(c.z, c.w) - 1.0, pr = p1 * p1 + p2 * p2;
            if (pr < 1.0 && pr >= 0x1p-60) {
                const double l1 = det_log(pr), l2 = det_log<true>(pr);
                bad_log += dbl_bits(l1) != dbl_bits(l2);
                const double q1 = (-2.0 * l1) / pr;
                bad_div += dbl_bits(q1) != dbl_bits(div_mid(-2.0 * l1, pr));
                bad_sqrt += dbl_bits(sqrt(q1)) != dbl_bits(sqrt_mid(q1));
                done += 3;
            }
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
// Self-test of the lazy quality normals.  mode 0: n tries w = first, first + 1, ... (a try is one 32-bit word: n = 2^32 covers EVERY try) -- the
// decision of quality_try_lazy is compared with quality_try_exact; out[0] = offsets that differ, out[1] = accept / reject verdicts that differ,
// out[2] = tries the lazy form decided, out[3] = tries it handed to the exact path, out[4] = tries, out[5] = max |y - x| / eps (double bits;
// estimate against the exact fp64 value, over decided values).  mode 1 / 2 / 3: EVERY float of the operand range of v_log_f32 ([2^-30, 1)),
// v_rcp_f32 ([1, 2^30]) and v_sqrt_f32 ([2^-41, 2^5)) against fp64: out[6] = max |log2_hw - log2| / (2^-23 (1 + |log2|)), out[7], out[8] = max
// relative error / 2^-23.
DW_DEV void atomic_max_pos_double(uint64_t *p, double v) { atomicMax((unsigned long long *)p, (unsigned long long)dbl_bits(v)); }
__global__ void __launch_bounds__(256) k_selftest_lazy(int mode, uint32_t first, uint64_t n, double sigma, float qk, float qeps, float qlmin, int qnear1, uint64_t *out)
{
    uint32_t bad_k = 0, bad_acc = 0, n_fast = 0, n_slow = 0, n_all = 0; double worst = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (mode == 0) {
            const uint32_t w = first + (uint32_t)i;
            const QualLazy ql{qk, qeps, qlmin, qnear1};
            int32_t k0 = 0, k1 = 0, e0 = 0, e1 = 0; bool eok = false;
            const uint32_t r = quality_try_lazy(w, ql, k0, k1);
            quality_try_exact(w, sigma, eok, e0, e1);
            ++n_all;
            if ((r & 1u) != (eok ? 1u : 0u)) ++bad_acc;
            else if (eok) {
                if (r & 2u) ++n_slow;
                else {
                    ++n_fast;
                    bad_k += (k0 != e0) + (k1 != e1);
                    // the estimate itself against the exact fp64 value (recomputed here as the lazy form computes it)
                    const int32_t s1 = (int32_t)(w & 0xFFFFu) - 32768, s2 = (int32_t)(w >> 16) - 32768;
                    const double v1 = (double)s1 * 0x1p-15, v2 = (double)s2 * 0x1p-15, rsq = v1 * v1 + v2 * v2;
                    const double fac = sqrt(-2.0 * det_log(rsq) / rsq);
                    const float Rf = (float)((uint32_t)(s1 * s1) + (uint32_t)(s2 * s2)), Lp = -__builtin_amdgcn_logf(Rf * 0x1p-30f);
                    if (!(Lp < ql.lmin)) {
                        const float f = __builtin_amdgcn_sqrtf(Lp * __builtin_amdgcn_rcpf(Rf));
                        const double y0 = (double)__builtin_fmaf((float)s2 * f, ql.k, 0.5f), y1 = (double)__builtin_fmaf((float)s1 * f, ql.k, 0.5f);
                        const double d0 = fabs(y0 - ((v2 * fac) * sigma + 0.5)), d1 = fabs(y1 - ((v1 * fac) * sigma + 0.5));
                        const double wv = (d0 > d1 ? d0 : d1) / (double)ql.eps;
                        worst = wv > worst ? wv : worst;
                    }      // (else "offset 0 without further work": covered by bad_k above -- the exact value must really truncate to 0)
                }
            }
        } else {
            const uint32_t base = mode == 1 ? 0x30800000u : mode == 2 ? 0x3F800000u : 0x2B000000u;       // 2^-30 : 1 : 2^-41
            union { uint32_t u; float f; } c; c.u = base + (uint32_t)i;
            const double x = (double)c.f;
            double e;
            if (mode == 1) { const double t = det_log(x) * 1.44269504088896340736; e = fabs((double)__builtin_amdgcn_logf(c.f) - t) / (0x1p-23 * (1.0 + fabs(t))); }
            else if (mode == 2) { const double t = 1.0 / x; e = fabs((double)__builtin_amdgcn_rcpf(c.f) - t) / t / 0x1p-23; }
            else { const double t = sqrt(x); e = fabs((double)__builtin_amdgcn_sqrtf(c.f) - t) / t / 0x1p-23; }
            worst = e > worst ? e : worst;
        }
    }
    const uint32_t s0 = wave_sum_u32(bad_k), s1 = wave_sum_u32(bad_acc), s2 = wave_sum_u32(n_fast), s3 = wave_sum_u32(n_slow), s4 = wave_sum_u32(n_all);
    if ((threadIdx.x & 63) == 0) {
        if (s0) atomicAdd((unsigned long long *)&out[0], (unsigned long long)s0);
        if (s1) atomicAdd((unsigned long long *)&out[1], (unsigned long long)s1);
        atomicAdd((unsigned long long *)&out[2], (unsigned long long)s2);
        atomicAdd((unsigned long long *)&out[3], (unsigned long long)s3);
        atomicAdd((unsigned long long *)&out[4], (unsigned long long)s4);
    }
    if (worst > 0.0) atomic_max_pos_double(&out[mode == 0 ? 5 : 5 + mode], worst);
}
void launch_selftest_lazy(hipStream_t st, int mode, uint32_t first, uint64_t n, double sigma, float qk, float qeps, float qlmin, int qnear1, uint64_t *out)
{
    const uint64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_selftest_lazy, dim3((uint32_t)(nb < (1u << 20) ? (nb ? nb : 1) : (1u << 20))), dim3(256), 0, st, mode, first, n, sigma, qk, qeps, qlmin, qnear1, out);
}
// Self-test of the number formatters (dw_read.hpp put_dec / put_hex: digits without a loop per digit) against one division per digit: the values
// first + i * stride, i < n -- their low 32 bits in decimal, all 64 in hexadecimal -- behind a separator, collected byte by byte.
// out[0] / out[1]: decimal / hexadecimal texts that differ, out[2]: values compared.
struct TextCollect {
    uint8_t b[24]; uint32_t n;
    DW_DEV void putn(uint64_t v, uint32_t cnt) { for (uint32_t k = 0; k < cnt; ++k) b[n++] = (uint8_t)(v >> (8 * k)); }
    DW_DEV void put_lead8(uint32_t lead, uint64_t v, uint32_t cnt) { b[n++] = (uint8_t)lead; putn(v, cnt); }
};
__global__ void __launch_bounds__(256) k_selftest_text(uint64_t first, uint64_t n, uint64_t stride, uint64_t *out)
{
    uint32_t bad_dec = 0, bad_hex = 0, done = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t v = first + i * stride;
        uint8_t want[24]; uint32_t wn;
        { uint8_t d[10]; uint32_t nd = 0, t = (uint32_t)v; do { d[nd++] = (uint8_t)('0' + t % 10u); t /= 10u; } while (t); wn = 0; want[wn++] = '_'; while (nd) want[wn++] = d[--nd]; }
        TextCollect c; c.n = 0;
        put_dec(c, (uint32_t)v, '_');
        bool same = c.n == wn && wn == 1u + ndigits10((uint32_t)v);
        for (uint32_t k = 0; k < wn && same; ++k) same = c.b[k] == want[k];
        bad_dec += same ? 0u : 1u;
        { uint8_t d[16]; uint32_t nd = 0; uint64_t t = v; do { const uint32_t h = (uint32_t)(t & 15u); d[nd++] = (uint8_t)(h < 10 ? '0' + h : 'a' + (h - 10)); t >>= 4; } while (t); wn = 0; want[wn++] = ':'; while (nd) want[wn++] = d[--nd]; }
        c.n = 0;
        put_hex(c, v, ':');
        same = c.n == wn && wn == 1u + ndigits16(v);
        for (uint32_t k = 0; k < wn && same; ++k) same = c.b[k] == want[k];
        bad_hex += same ? 0u : 1u;
        ++done;
    }
    if (bad_dec) atomicAdd((unsigned long long *)&out[0], (unsigned long long)bad_dec);
    if (bad_hex) atomicAdd((unsigned long long *)&out[1], (unsigned long long)bad_hex);
    atomicAdd((unsigned long long *)&out[2], (unsigned long long)done);
}
void launch_selftest_text(hipStream_t st, uint64_t first, uint64_t n, uint64_t stride, uint64_t *out)
{
    const uint64_t nb = cdiv(n, 256);
    hipLaunchKernelGGL(k_selftest_text, dim3((uint32_t)(nb < (1u << 16) ? (nb ? nb : 1) : (1u << 16))), dim3(256), 0, st, first, n, stride, out);
}
void launch_selftest_fp64(hipStream_t st, uint32_t seed, uint64_t n, uint64_t *mism)
{
    hipLaunchKernelGGL(k_selftest_fp64, dim3(cdiv(n, 256)), dim3(256), 0, st, seed, n, mism);
}
// ---- the two-kernel form of k_simulate (SPLIT): every block's offsets from the sums its first half left ----
// Exclusive sums over the logical blocks of {random pairs, bytes of stream 1, bytes of stream 2} (agg, 16 bytes per block: the byte counts
// without the hexadecimal digits of random reads' names), in two small kernels.  k_split_scan1: a block per 1024 logical blocks -- their sums
// scanned inside the chunk -> pre[4 t .. 4 t + 2], the chunk's totals -> chunk[4 c ..].  k_split_scan2 (one wave): the chunk totals scanned ->
// chunk[4 c ..] = sums in front of chunk c, and the launch's totals into counters[3 .. 6] as the single kernel leaves them.  The second half
// adds pre + chunk and the digits (split_offsets below).
__global__ void __launch_bounds__(1024) k_split_scan1(SimArgs a)
{
    __shared__ uint32_t sm[3][16];
    const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
    uint4 g = make_uint4(0, 0, 0, 0);
    if (t < a.n_blocks) g = reinterpret_cast<const uint4 *>(a.split_agg)[t];
    const uint32_t v[3] = {g.x, g.y, g.z};
    uint32_t ex[3], tot[3];
    block_excl_scan_n<3>(v, sm, ex, tot);
    if (t < a.n_blocks) { uint64_t *o = a.split_pre + 4 * (size_t)t; o[0] = ex[0]; o[1] = ex[1]; o[2] = ex[2]; }
    if (threadIdx.x == 0) { uint64_t *c = a.split_chunk + 4 * (size_t)blockIdx.x; c[0] = tot[0]; c[1] = tot[1]; c[2] = tot[2]; }
}
__global__ void __launch_bounds__(64) k_split_scan2(SimArgs a, int lpp)
{
    const uint32_t nch = (a.n_blocks + 1023u) / 1024u;
    uint64_t run[3] = {0, 0, 0};
    for (uint32_t base = 0; base < nch; base += 64u) {
        const uint32_t c = base + threadIdx.x;
        uint64_t v[3] = {0, 0, 0};
        if (c < nch) for (int k = 0; k < 3; ++k) v[k] = a.split_chunk[4 * (size_t)c + k];
        uint64_t inc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint64_t x = v[k];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t ol = (uint32_t)__shfl_up((int)(uint32_t)x, d), oh = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d); if (lane_id() >= d) x += ((uint64_t)oh << 32) | ol; }
            inc[k] = x;
        }
        if (c < nch) for (int k = 0; k < 3; ++k) a.split_chunk[4 * (size_t)c + k] = run[k] + inc[k] - v[k];
        for (int k = 0; k < 3; ++k) { const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)inc[k], 63), hi = (uint32_t)__shfl((int)(uint32_t)(inc[k] >> 32), 63); run[k] += ((uint64_t)hi << 32) | lo; }
    }
    if (threadIdx.x == 0) {
        const uint64_t H = hex_digits_sum(a.chain[0], run[0]), G1 = run[1] + H, G2 = run[2] + (lpp == 2 ? H : 0ull);
        const uint64_t nreads = a.n_pairs * (uint64_t)((a.p.len[1] > 0) ? 2 : 1);
        a.counters[3] = run[0];
        a.counters[4] = a.p.has_bwa ? G1 : 0;
        a.counters[5] = a.p.has_bwa ? G2 : 0;
        a.counters[6] = !a.p.has_bfast ? 0 : G1 + G2 - 2 * nreads;
    }
}

// ---- the reference's abort rule (dwgsim.c:635, :833-843): one counter of failed attempts runs over the pairs of a contig in index order,
// a pair that ends as a genomic read resets it, a pair that ends as a random read does not, and the job dies as soon as the counter
// passes 10 000.  Pairs are simulated independently here, so the rule is evaluated afterwards from the per-pair record
// meta[pair] = failed attempts | opens its contig << 30 | random << 31 -- only for batches that had a failed attempt at all.  A launch can cover
// several contigs; the counter is per contig (`int num_failed = 0`, dwgsim.c:635), so a pair that opens its contig first puts a reset in front
// of itself: MARK = {0, 0, 1, 0} (joining it after a run that already passed the limit still flags that run: the monoid's `bad` rule).
// A run of pairs is summarised as {P: fails before its first reset (all of them if it has none), S: fails after its last reset,
// R: has a reset, bad: a run between two of its resets passed the limit}.
struct FailSeg { uint64_t P, S; uint32_t R, bad; };
DW_DEV FailSeg failseg_join(const FailSeg &a, const FailSeg &b)
{
    FailSeg r;
    r.bad = a.bad | b.bad | ((a.R && a.S + b.P > (uint64_t)MAX_ATTEMPTS) ? 1u : 0u);
    r.P = a.R ? a.P : a.P + b.P;                    // (the prefix of a run without a reset extends into the next one)
    r.S = b.R ? b.S : a.S + b.S;
    r.R = a.R | b.R;
    return r;
}
// the join of the segments held by lanes 0 .. cnt-1 of a wave, in lane order (a tree of ordered joins: the monoid is associative); result in lane 0
DW_DEV FailSeg failseg_wave_join(FailSeg x, int cnt)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        FailSeg o;
        o.P = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(x.P >> 32), d) << 32) | (uint32_t)__shfl_down((int)(uint32_t)x.P, d);
        o.S = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(x.S >> 32), d) << 32) | (uint32_t)__shfl_down((int)(uint32_t)x.S, d);
        o.R = (uint32_t)__shfl_down((int)x.R, d); o.bad = (uint32_t)__shfl_down((int)x.bad, d);
        if ((lane & (2 * d - 1)) == 0 && lane + d < cnt) x = failseg_join(x, o);
    }
    return x;
}
constexpr int FAIL_PAIRS_PER_THREAD = 64;
// A: thread = 64 consecutive pairs, block = 256 threads; one FailSeg per block into summ[4 * block .. +4).  A batch without a single
// failed attempt (counters[1] == 0, the usual case) is summarised by B from the counters alone.
__global__ void __launch_bounds__(256) k_failrule_a(const uint32_t *__restrict__ meta, uint64_t n_pairs, const uint64_t *__restrict__ counters, uint64_t *__restrict__ summ)
{
    __shared__ FailSeg seg[4];
    if (counters[1] == 0) return;
    const uint64_t first = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * FAIL_PAIRS_PER_THREAD;
    FailSeg s{0, 0, 0, 0};                                          // the neutral element: nothing failed, no reset (also what a thread past the end holds)
    for (int q4 = 0; q4 < FAIL_PAIRS_PER_THREAD && first + q4 < n_pairs; q4 += 4) {       // 16-byte loads (meta is padded to a multiple of four entries)
        const uint4 v = *reinterpret_cast<const uint4 *>(meta + first + q4);
        const uint32_t mm[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (first + q4 + u >= n_pairs) break;
            const uint32_t m = mm[u];
            if (m & 0x40000000u) s = failseg_join(s, FailSeg{0, 0, 1, 0});      // the pair opens its contig: the counter starts from zero
            FailSeg one{m & 0x3fffffffu, 0, (m >> 31) ? 0u : 1u, 0};           // the pair's fails come before its outcome; a genomic read resets
            if (!one.R) one.S = one.P;                                          // no reset: everything is both prefix and suffix
            s = failseg_join(s, one);
        }
    }
    s = failseg_wave_join(s, 64);
    if ((threadIdx.x & 63) == 0) seg[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        FailSeg t = seg[0];
        for (int k = 1; k < 4; ++k) t = failseg_join(t, seg[k]);
        summ[4 * (uint64_t)blockIdx.x + 0] = t.P; summ[4 * (uint64_t)blockIdx.x + 1] = t.S;
        summ[4 * (uint64_t)blockIdx.x + 2] = t.R; summ[4 * (uint64_t)blockIdx.x + 3] = t.bad;
    }
}
// B, the batch epilogue (one wave): the block summaries are joined in order (64 lanes stage them through LDS, lane 0 joins) into the
// segment of this batch alone -> counters[16..19]; behind the carry of the earlier batches (chain[1]) that gives the verdict
// counters[20] = abort? and the carry out counters[21] = chain[1]; the running random-read count chain[0] moves on by counters[3].
__global__ void __launch_bounds__(64) k_failrule_b(const uint64_t *__restrict__ summ, uint32_t n_blocks, uint64_t n_pairs, uint32_t opens_contig, uint64_t *__restrict__ counters, uint64_t *__restrict__ chain)
{
    FailSeg t{0, 0, (counters[3] < n_pairs || opens_contig) ? 1u : 0u, 0};          // no failed attempt at all: any genomic read -- or the start of a contig inside the batch -- resets the counter
    if (counters[1] != 0) {                                       // lane k joins its slice of the block summaries, the wave joins the slices in order
        const uint32_t per = (n_blocks + 63) / 64, b0 = threadIdx.x * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
        FailSeg x{0, 0, 0, 0};
        for (uint32_t b = b0; b < b1; ++b) {
            const FailSeg nx{summ[4 * (uint64_t)b], summ[4 * (uint64_t)b + 1], (uint32_t)summ[4 * (uint64_t)b + 2], (uint32_t)summ[4 * (uint64_t)b + 3]};
            x = (b == b0) ? nx : failseg_join(x, nx);
        }
        t = failseg_wave_join(x, per ? (int)((n_blocks + per - 1) / per) : 0);
    }
    if (threadIdx.x == 0) {
        const uint64_t carry = chain[1];
        const FailSeg all = failseg_join(FailSeg{carry, carry, 0, 0}, t);
        counters[16] = t.P; counters[17] = t.S; counters[18] = t.R; counters[19] = t.bad;
        counters[20] = (all.bad || all.P > (uint64_t)MAX_ATTEMPTS || all.S > (uint64_t)MAX_ATTEMPTS) ? 1 : 0;
        counters[21] = all.S;
        chain[1] = all.S;
        counters[22] = chain[0];      // what this batch's k_simulate started from (a batch is run again when an Ion Torrent read outgrew its buffers: dw_host.cpp)
        chain[0] += counters[3];
    }
}
void launch_failrule(hipStream_t st, const uint32_t *meta, uint64_t n_pairs, uint32_t opens_contig, uint64_t *summ, uint64_t *counters, uint64_t *chain)
{
    const uint32_t nb = cdiv(n_pairs, 256ull * FAIL_PAIRS_PER_THREAD);
    hipLaunchKernelGGL(k_failrule_a, dim3(nb), dim3(256), 0, st, meta, n_pairs, counters, summ);
    hipLaunchKernelGGL(k_failrule_b, dim3(1), dim3(64), 0, st, summ, nb, n_pairs, opens_contig, counters, chain);
}
// chain[0] (random reads before the next batch) and / or chain[1] (the abort rule's carry) set in stream order
__global__ void __launch_bounds__(64) k_chain_set(uint64_t *chain, uint64_t rand_base, int set_rand, uint64_t carry, int set_carry)
{
    if (threadIdx.x != 0) return;
    if (set_rand) chain[0] = rand_base;
    if (set_carry) chain[1] = carry;
}
void launch_chain_set(hipStream_t st, uint64_t *chain, uint64_t rand_base, int set_rand, uint64_t carry, int set_carry)
{
    hipLaunchKernelGGL(k_chain_set, dim3(1), dim3(64), 0, st, chain, rand_base, set_rand, carry, set_carry);
}
// Everything a launch needs reset in front of it, in ONE stream operation: its counters, the look-back words of its blocks (and the scratch slots' free lists), the running
// values it starts from.  (Rounds 2-5: two or three memsets and k_chain_set -- four operations of ~10 us each in front of a kernel that takes 400 us for an E. coli-sized contig.)
__global__ void __launch_bounds__(256) k_launch_init(uint64_t *counters, uint32_t n_counters, uint64_t *z0, uint64_t n0, uint64_t *z1, uint64_t n1, uint64_t *chain, uint64_t rand_base, int set_rand, uint64_t carry, int set_carry)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    if (i < n_counters) counters[i] = 0;
    for (uint64_t k = i; k < n0; k += stride) z0[k] = 0;
    for (uint64_t k = i; k < n1; k += stride) z1[k] = 0;
    if (i == 0) { if (set_rand) chain[0] = rand_base; if (set_carry) chain[1] = carry; }
}
void launch_init(hipStream_t st, uint64_t *counters, uint32_t n_counters, uint64_t *z0, uint64_t n0, uint64_t *z1, uint64_t n1, uint64_t *chain, uint64_t rand_base, int set_rand, uint64_t carry, int set_carry)
{
    const uint64_t most = n0 > n1 ? n0 : n1, want = (most + 255) / 256;
    hipLaunchKernelGGL(k_launch_init, dim3((uint32_t)(want < 1 ? 1 : want > 2048 ? 2048 : want)), dim3(256), 0, st, counters, n_counters, z0, n0, z1, n1, chain, rand_base, set_rand, carry, set_carry);
}
// test / analysis hook: how many bytes of text[0 .. n) equal `byte` (size-independent checks of whole outputs without copying them out)
__global__ void __launch_bounds__(256) k_count_byte(const uint8_t *__restrict__ text, uint64_t n, uint32_t byte, uint64_t *out)
{
    const uint32_t pat = byte * 0x01010101u;
    uint32_t cnt = 0;
    const uint64_t nvec = n / 16;
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (uint64_t)gridDim.x * 256) {
        const uint4 q = reinterpret_cast<const uint4 *>(text)[v];
        const uint32_t w[4] = {q.x ^ pat, q.y ^ pat, q.z ^ pat, q.w ^ pat};
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt += (uint32_t)__popc(~(((w[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[k] | 0x7F7F7F7Fu));      // zero bytes of w[k]
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) for (uint64_t q = nvec * 16; q < n; ++q) cnt += text[q] == byte;
    const uint32_t s = wave_sum_u32(cnt);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long *)out, (unsigned long long)s);
}
void launch_count_byte(hipStream_t st, const uint8_t *text, uint64_t n, uint32_t byte, uint64_t *out)
{
    hipLaunchKernelGGL(k_count_byte, dim3(4096), dim3(256), 0, st, text, n, byte, out);
}
// a.segs / a.n_blocks laid out for PLACE_PAIRS pairs per block; a.block_rand[n_blocks], a.range_rand[n_seg] (zeroed), a.place_list_n (zeroed).
// Behind it: a.block_rand scanned (exclusive) with its total in a.counters[3], a.range_rand[q] = random reads of range q.
void launch_place(hipStream_t st, const SimArgs &a)
{
    hipLaunchKernelGGL(k_place, dim3(a.n_blocks), dim3(PLACE_PAIRS), 0, st, a);
    const uint32_t per_list = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, ((uint64_t)a.place_list_cap + 1023) / 1024));      // blocks of 128 lanes per list, sized for a tenth of the capacity in one sweep
    hipLaunchKernelGGL(k_place_rest, dim3(PLACE_LISTS * per_list), dim3(128), 0, st, a);
    launch_scan_excl(st, a.block_rand, a.n_blocks, &a.counters[3]);
    hipLaunchKernelGGL(k_range_counts, dim3(cdiv((uint64_t)a.n_seg, 256)), dim3(256), 0, st, a);
}
// one launcher per (LPP, DT) family, each defined in its own part
void launch_sim_2_0(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_1_0(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_2_2(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_1_2(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_2_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_1_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_long_2_0(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_long_1_0(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_long_2_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_long_1_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_2_3(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_1_3(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_long_2_3(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_long_1_3(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out);
void launch_sim_split_2(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds_a, size_t lds_b, int out);
void launch_sim_split_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds_a, size_t lds_b, int out);
void launch_sim_split_ion_2(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds_a, size_t lds_b, int out);
void launch_sim_split_ion_1(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds_a, size_t lds_b, int out);
void launch_split_scan(hipStream_t st, const SimArgs &a, int lpp)
{
    hipLaunchKernelGGL(k_split_scan1, dim3(cdiv(a.n_blocks, 1024)), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_split_scan2, dim3(1), dim3(64), 0, st, a, lpp);
}
void launch_simulate(hipStream_t st, const SimArgs &a)
{
    const bool pe = a.p.len[1] > 0, ion = a.p.data_type == 2;
    const uint32_t nthr = (uint32_t)a.sim_threads;
    const uint32_t nb = a.n_blocks;                                  // a.segs is laid out for nthr / (pe ? 2 : 1) pairs per block
    const int out = (a.p.has_bwa ? 1 : 0) | (a.p.has_bfast ? 2 : 0);
    if (a.split) {      // 256-lane blocks: first half | offsets | second half (k_simulate<.., SPLIT>)
        const size_t lds_a = sim_lds_bytes((size_t)(a.lds_words + (ion ? a.flow_stack_words : 0)), nthr, 0, false), lds_b = sim_lds_bytes(0, nthr, (size_t)a.qb_words, a.fifo != 0, SIM_FIFO_BYTES_WIDE);      // (the second half stages no bases)
        if (ion) { if (pe) launch_sim_split_ion_2(st, a, nb, lds_a, lds_b, out); else launch_sim_split_ion_1(st, a, nb, lds_a, lds_b, out); }
        else if (pe) launch_sim_split_2(st, a, nb, lds_a, lds_b, out); else launch_sim_split_1(st, a, nb, lds_a, lds_b, out);
        return;
    }
    // staged bases (Ion Torrent: the read buffers when LDS holds them + the pass-2 run stack; otherwise they are, like the long reads of the one-wave blocks, in a.flow_scratch)
    // + the base-quality tables + the text FIFOs
    const size_t lds = sim_lds_bytes((size_t)(ion ? (a.ion_lds ? a.lds_words : 0) + a.flow_stack_words : nthr != (uint32_t)SIM_THREADS ? 0 : a.lds_words), nthr, (size_t)a.qb_words, a.fifo != 0);
    const bool solid = a.p.data_type == 1;
    if (ion && a.ion_lds) {                                          // Ion Torrent, read buffers in LDS: 256-lane blocks, or one-wave blocks
        if (nthr != (uint32_t)SIM_THREADS) { if (pe) launch_sim_long_2_3(st, a, nb, lds, out); else launch_sim_long_1_3(st, a, nb, lds, out); }
        else { if (pe) launch_sim_2_3(st, a, nb, lds, out); else launch_sim_1_3(st, a, nb, lds, out); }
        return;
    }
    if (nthr != (uint32_t)SIM_THREADS) {                             // long Illumina / SOLiD reads: one-wave blocks
        if (pe) { if (solid) launch_sim_long_2_1(st, a, nb, lds, out); else launch_sim_long_2_0(st, a, nb, lds, out); }
        else { if (solid) launch_sim_long_1_1(st, a, nb, lds, out); else launch_sim_long_1_0(st, a, nb, lds, out); }
        return;
    }
    if (pe) { if (ion) launch_sim_2_2(st, a, nb, lds, out); else if (solid) launch_sim_2_1(st, a, nb, lds, out); else launch_sim_2_0(st, a, nb, lds, out); }
    else { if (ion) launch_sim_1_2(st, a, nb, lds, out); else if (solid) launch_sim_1_1(st, a, nb, lds, out); else launch_sim_1_0(st, a, nb, lds, out); }
}
#endif // DW_HAS(0): launchers

#define DW_SIM_FAMILY(LPP, DT)                                                                                   \
    void launch_sim_##LPP##_##DT(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out)             \
    {                                                                                                            \
        const uint32_t nthr = SIM_THREADS;                                                                       \
        if (DT == 0 && !a.fifo) {                                                                                \
            if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, DT == 0 ? 0 : DT, SIM_THREADS, DT == 0 ? 0 : 1>), dim3(nb), dim3(nthr), lds, st, a);      \
            else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, DT == 0 ? 0 : DT, SIM_THREADS, DT == 0 ? 0 : 1>), dim3(nb), dim3(nthr), lds, st, a); \
            else hipLaunchKernelGGL((k_simulate<LPP, 3, DT == 0 ? 0 : DT, SIM_THREADS, DT == 0 ? 0 : 1>), dim3(nb), dim3(nthr), lds, st, a);               \
            return;                                                                                              \
        }                                                                                                        \
        if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, DT>), dim3(nb), dim3(nthr), lds, st, a);            \
        else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, DT>), dim3(nb), dim3(nthr), lds, st, a);       \
        else hipLaunchKernelGGL((k_simulate<LPP, 3, DT>), dim3(nb), dim3(nthr), lds, st, a);                     \
    }
#define DW_SIM_FAMILY_LONG(LPP, DT)                                                                                             \
    void launch_sim_long_##LPP##_##DT(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out)                          \
    {                                                                                                                               \
        const uint32_t nthr = SIM_THREADS_LONG;                                                                                     \
        if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, DT, SIM_THREADS_LONG>), dim3(nb), dim3(nthr), lds, st, a);             \
        else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, DT, SIM_THREADS_LONG>), dim3(nb), dim3(nthr), lds, st, a);        \
        else hipLaunchKernelGGL((k_simulate<LPP, 3, DT, SIM_THREADS_LONG>), dim3(nb), dim3(nthr), lds, st, a);                      \
    }
// the two-kernel form (Illumina, 256-lane blocks): first half, k_split_scan, second half
void launch_split_scan(hipStream_t st, const SimArgs &a, int lpp);
#define DW_SIM_SPLIT(LPP)                                                                                                                  \
    void launch_sim_split_##LPP(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds_a, size_t lds_b, int out)                        \
    {                                                                                                                                      \
        hipLaunchKernelGGL((k_simulate<LPP, 1, 0, SIM_THREADS, 1, 1>), dim3(nb), dim3(SIM_THREADS), lds_a, st, a);                         \
        launch_split_scan(st, a, LPP);                                                                                                     \
        if (a.fifo) {                                                                                                                      \
            if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, 0, SIM_THREADS, 2, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);       \
            else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, 0, SIM_THREADS, 2, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);  \
            else hipLaunchKernelGGL((k_simulate<LPP, 3, 0, SIM_THREADS, 2, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);                \
        } else {                                                                                                                           \
            if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, 0, SIM_THREADS, 0, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);       \
            else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, 0, SIM_THREADS, 0, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);  \
            else hipLaunchKernelGGL((k_simulate<LPP, 3, 0, SIM_THREADS, 0, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);                \
        }                                                                                                                                  \
    }
// ... of Ion Torrent with its buffers in LDS: the first half is the flow model (no text FIFOs: a block more per CU, and no block waits for the record
// sizes of the blocks in front of it), the second half qualities and text from the finished reads in HBM
#define DW_SIM_SPLIT_ION(LPP)                                                                                                              \
    void launch_sim_split_ion_##LPP(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds_a, size_t lds_b, int out)                    \
    {                                                                                                                                      \
        hipLaunchKernelGGL((k_simulate<LPP, 1, 3, SIM_THREADS, 1, 1>), dim3(nb), dim3(SIM_THREADS), lds_a, st, a);                         \
        launch_split_scan(st, a, LPP);                                                                                                     \
        if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, 3, SIM_THREADS, 2, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);           \
        else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, 3, SIM_THREADS, 2, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);      \
        else hipLaunchKernelGGL((k_simulate<LPP, 3, 3, SIM_THREADS, 2, 2>), dim3(nb), dim3(SIM_THREADS), lds_b, st, a);                    \
    }
#if DW_HAS(14)
DW_SIM_SPLIT_ION(2)
#endif
#if DW_HAS(15)
DW_SIM_SPLIT_ION(1)
#endif
#if DW_HAS(1)
DW_SIM_FAMILY(2, 0)
#endif
#if DW_HAS(9)
DW_SIM_SPLIT(2)
#endif
#if DW_HAS(10)
DW_SIM_SPLIT(1)
#endif
#if DW_HAS(7)
DW_SIM_FAMILY_LONG(2, 0)
DW_SIM_FAMILY_LONG(1, 0)
#endif
#if DW_HAS(8)
DW_SIM_FAMILY_LONG(2, 1)
DW_SIM_FAMILY_LONG(1, 1)
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
// the streams of (D_CALIB + end, read, attempt 1).  The read buffers (dw_read.hpp flow_errors: a.lds_words words per lane) are in a.scratch.
__global__ void __launch_bounds__(PAIRS_PER_BLOCK) k_calibrate(CalibArgs a)
{
    DW_DYN_SHARED(uint32_t, dyn_lds);
    __shared__ FlowTables s_ft;
    const int tid = (int)threadIdx.x, nthr = PAIRS_PER_BLOCK;
    if (tid < 64) s_ft.flow[tid] = a.flow[tid];
    __syncthreads();
    fill_flow_tables(s_ft, a.flow_len, tid, nthr, a.flow);
    __syncthreads();
    const uint64_t jj = a.first_read + (uint64_t)blockIdx.x * PAIRS_PER_BLOCK + (uint64_t)tid;
    uint32_t *buf = a.scratch + (size_t)blockIdx.x * ((size_t)a.lds_words * nthr) + tid;
    const int capb = 16 * a.lds_words;
    int32_t n_err = 0; int s_out = 0;
    const bool live = jj < a.n_reads;
    const RngKey key{a.seed, 0u};
    const uint32_t dom = D_CALIB + (uint32_t)a.end;
    if (live) {
        const int h0 = ((capb - a.len) & ~7) >> 3;             // a forward read: where FlowSink would put it
        for (int w = 0; w * 8 < a.len; ++w) {
            const U4 q0 = rng_block(key, dom, jj, 0, 0, (uint32_t)(2 * w)), q1 = rng_block(key, dom, jj, 0, 0, (uint32_t)(2 * w + 1));
            const uint32_t rw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            uint32_t pairs = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) if (w * 8 + b < a.len) pairs |= (rw[b] >> 30) << (2 * b);      // (int)(u * 4.0) & 3
            const int h = h0 + w;
            reinterpret_cast<uint16_t *>(buf + (h >> 1) * nthr)[h & 1] = (uint16_t)pairs;
        }
    }
    {   // every lane calls (the loops of the model are wave-uniform)
        FlowRng rg; rg.seed = a.seed; rg.contig = 0; rg.dom = dom; rg.att = 1; rg.evt = 0; rg.s = 0; rg.ii = jj; rg.w0 = rg.w1 = rg.w2 = rg.w3 = 0;
        const int so = flow_errors(live, rg, s_ft, a.flow_len, a.thr, a.gap_r, a.gap_s, buf, dyn_lds + tid, nthr, 2 * a.stack_words, a.len, 0, capb, &n_err);
        if (live) { s_out = so; if (s_out < 0) { atomicOr((unsigned long long *)&a.counters[2], 2ull); s_out = 0; n_err = 0; } }
    }
    const uint32_t es = wave_sum_u32((uint32_t)n_err), ls = wave_sum_u32((uint32_t)s_out);
    if ((tid & 63) == 0) { atomicAdd((unsigned long long *)&a.counters[8], (unsigned long long)es); atomicAdd((unsigned long long *)&a.counters[9], (unsigned long long)ls); }
}
void launch_calibrate(hipStream_t st, const CalibArgs &a)
{
    hipLaunchKernelGGL(k_calibrate, dim3(cdiv(std::min(a.chunk_reads, a.n_reads - a.first_read), (uint64_t)PAIRS_PER_BLOCK)), dim3(PAIRS_PER_BLOCK), (size_t)a.stack_words * PAIRS_PER_BLOCK * 4, st, a);
}
#endif
#if DW_HAS(11)
DW_SIM_FAMILY(2, 3)
#endif
#if DW_HAS(12)
DW_SIM_FAMILY(1, 3)
#endif
#if DW_HAS(13)
#define DW_SIM_FAMILY_ION_SMALL(LPP)                                                                                               \
    void launch_sim_long_##LPP##_3(hipStream_t st, const SimArgs &a, uint32_t nb, size_t lds, int out)                              \
    {                                                                                                                               \
        const uint32_t nthr = ION_THREADS_SMALL;                                                                                    \
        if (out == 1) hipLaunchKernelGGL((k_simulate<LPP, 1, 3, ION_THREADS_SMALL>), dim3(nb), dim3(nthr), lds, st, a);             \
        else if (out == 2) hipLaunchKernelGGL((k_simulate<LPP, 2, 3, ION_THREADS_SMALL>), dim3(nb), dim3(nthr), lds, st, a);        \
        else hipLaunchKernelGGL((k_simulate<LPP, 3, 3, ION_THREADS_SMALL>), dim3(nb), dim3(nthr), lds, st, a);                      \
    }
DW_SIM_FAMILY_ION_SMALL(2)
DW_SIM_FAMILY_ION_SMALL(1)
#endif
#if DW_HAS(5)
DW_SIM_FAMILY(2, 1)
#endif
#if DW_HAS(6)
DW_SIM_FAMILY(1, 1)
#endif

} // namespace dw
