// dw_read.hpp -- device code of one simulated read end, shared by k_place and k_simulate (dw_simulate.hip):
//   placement / geometry / base extraction through indels (dwgsim.c:649-843, :75-153), the Ion Torrent flow-space error
//   model (dwgsim.c:246-417), and the FASTQ text assembly (64-byte burst writer, packed decimal / hex fields).
#pragma once
#include "dw_device.hpp"
#include <dw_probe.hpp>

namespace dw {

// ------------------------------------------------------------------------------------------------
// Read simulation
// ------------------------------------------------------------------------------------------------
struct ReadRes { int32_t ext_coor, n_sub, n_indel, num_n, n_ins; };   // n_ins: INSERT cells crossed (the reference's n_indel_first, dwgsim.c:98)

// dwgsim.c:75-153 __gen_read.  STORE: packed 4-bit bases go to lds[word * stride].
// The haplotype is read through its 4-bit view (HapDev::view).  A read's window of the view sits at a random place of a contig that no cache
// of the chip holds for the ~10^5 lanes in flight: what the extraction costs is the memory round trips it takes one after the other, not
// its arithmetic.  So the window of the next 23 staged words (184 cells) is fetched by up to six 16-byte loads issued TOGETHER -- one round
// trip -- into 24 registers, in travel order (a reverse-strand lane loads downwards and turns each block around), and the words are then cut
// out of neighbouring registers with a funnel shift at compile-time register indices: nibble-parallel substitution / N counts, nibble
// reversal and complement for the reverse strand.  A word that holds an escape (an INSERT / DELETE cell or a '-': nibble >= 9), and any word
// too close to a contig end for the window to be loaded, goes through an EPISODE of the reference's per-cell logic on the byte cells, which
// runs on until the output stands at a word boundary again (insertions can carry it over several words); then a new window is loaded from
// the cell the episode stopped at.  Indel cells are rare (one word in a thousand at dwgsim's default rates).
DW_DEV uint64_t reverse_nibbles(uint64_t x)
{
    x = __builtin_bswap64(x);
    return ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
}
DW_DEV uint32_t reverse_nibbles32(uint32_t x)
{
    x = __builtin_bswap32(x);
    return ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
}
struct __attribute__((packed, aligned(4))) ViewQuad { uint32_t a, b, c, d; };      // four consecutive words of the view at a 4-byte aligned address: one dwordx4 load
constexpr int WIN_CHUNKS = 6, WIN_WORDS = 4 * WIN_CHUNKS - 1;                    // 16-byte blocks per window; staged words it serves
template <bool STORE>
DW_DEV ReadRes gen_read(const HapDev &h, int64_t l, int64_t start, int step, int s, int strand, uint32_t *lds, int stride)
{
    ReadRes r{-10, 0, 0, 0, 0};
    const bool fwd = step > 0;
    const int32_t li = (int32_t)l;                     // (contigs are shorter than 2^31)
    int k = 0;                                         // bases produced; a multiple of 8 whenever a window or an episode starts
    int32_t i = (start < -0x7fffffffll || start > 0x7fffffffll) ? -1 : (int32_t)start;      // next cell in travel order
    uint32_t accw = 0; int na = 0;                     // (episodes) the word being filled
    auto emit = [&](uint32_t v) {                      // one base from the per-cell logic
        if (strand) v = v < 4 ? 3 - v : 4;             // dwgsim.c:150-152
        r.num_n += (v == 4);                            // dwgsim.c:824-831
        accw |= v << (4 * na); ++k;
        if (++na == 8) { if (STORE) lds[((k >> 3) - 1) * stride] = accw; accw = 0; na = 0; }
    };
    const uint32_t *const vw = reinterpret_cast<const uint32_t *>(h.view);
    while (k < s) {
        // ---- a window: the next nw words = the next `cells` cells in travel order, all inside the contig
        const int words_left = (s - k + 7) >> 3, nw = words_left < WIN_WORDS ? words_left : WIN_WORDS;
        const int cells = (s - k) < 8 * nw ? (s - k) : 8 * nw;
        const int nch = (nw + 4) >> 2;                                   // 16-byte blocks that hold the nw + 1 words a window of nw staged words is cut from
        // forward: words b0, b0 + 1, ... of the view; reverse: words eb, eb - 1, ... (eb holds cell i)
        const int32_t b0 = i >> 3, eb = i >> 3;
        const bool win = i >= 0 && i < li && (fwd ? i <= li - cells : (i - cells + 1 >= 0 && eb - 4 * nch + 1 >= 0));
        bool esc = !win;
        if (win) {
            uint32_t X[4 * WIN_CHUNKS];
#pragma unroll
            for (int c = 0; c < WIN_CHUNKS; ++c) {
                ViewQuad q{0, 0, 0, 0};
                if (c < nch) q = *reinterpret_cast<const ViewQuad *>(vw + (fwd ? b0 + 4 * c : eb - 4 * c - 3));
                X[4 * c] = fwd ? q.a : q.d; X[4 * c + 1] = fwd ? q.b : q.c; X[4 * c + 2] = fwd ? q.c : q.b; X[4 * c + 3] = fwd ? q.d : q.a;      // travel order
            }
            // forward: word w = cells from nibble (i & 7) of X[w], continued in X[w + 1].  Reverse: the eight cells ending at cell i - 8 w start at
            // nibble ((i + 1) & 7) of view word eb - w - d (d = 1 unless cell i is the last nibble of its word): X[w + d], continued in X[w + d - 1]
            const bool d1 = !fwd && ((uint32_t)i & 7u) != 7u;
            const uint32_t sh = 4u * ((fwd ? (uint32_t)i : (uint32_t)i + 1u) & 7u);
            bool go = true;
#pragma unroll
            for (int w = 0; w < WIN_WORDS; ++w) {
                if (!(go && w < nw)) continue;                                             // (no early exit: the loop unrolls into straight-line code on fixed registers)
                const int want = s - k < 8 ? s - k : 8;
                const uint32_t lo = d1 ? X[w + 1] : X[w], hi = d1 ? X[w] : X[w + 1];
                uint32_t x = __builtin_amdgcn_alignbit(hi, lo, sh);
                if (!fwd) x = reverse_nibbles32(x);                                        // travel order: base 0 of the word in nibble 0
                if (want < 8) x &= (1u << (4 * want)) - 1u;
                const uint32_t n8 = x & 0x88888888u;                                        // nibble >= 8
                if (n8 & ((x & 0x77777777u) + 0x77777777u)) { esc = true; go = false; }     // ... and >= 9: an escape among them
                else {
                    if (r.ext_coor < 0) { r.ext_coor = i; if (strand) r.ext_coor -= s - 1; }
                    r.n_sub += __popc(x & 0x44444444u);                                     // nibbles 4-7: substituted cells
                    r.num_n += __popc(n8);                                                  // nibble 8: an N
                    uint32_t codes = x & 0x33333333u;
                    if (strand) codes = (codes ^ 0x33333333u) & ~((n8 >> 3) * 3u);          // complement, dwgsim.c:150-152 (N stays N)
                    if (want < 8) codes &= (1u << (4 * want)) - 1u;
                    codes |= n8 >> 1;                                                        // N = code 4
                    if (STORE) lds[(k >> 3) * stride] = codes;
                    k += want; i += fwd ? want : -want;
                }
            }
        }
        if (!esc) continue;
        // ---- an episode of the reference's per-cell logic, until the output stands at a word boundary again
        for (;;) {
            if (i < 0 || i >= li) { k = -1; break; }                 // walked off the contig before the read was complete
            const uint32_t c = h.cells[i], mt = c & TMASK;
            if (r.ext_coor < 0) {
                if (mt != T_NONE && mt != T_SUB) { i += step; continue; }
                r.ext_coor = i;
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
            if (k >= s || na == 0) break;
        }
        if (k < 0) break;
    }
    if (k != s) { r.ext_coor = -10; return r; }
    if (STORE && na) lds[(k >> 3) * stride] = accw;      // the ragged last word of an episode
    return r;
}

struct PairDraw { bool is_rand; int32_t pos, d; int hap, strand0, strand1; };

// select-by-value accessors: dynamic indexing into the by-value kernel argument block would force a
// private copy of the whole struct (promoted to LDS by the backend)
// The block's range of the launch: the fields of its SimSeg (block-uniform, scalar registers) in the form the read code uses
struct SegCtx {
    int64_t l, l_place;            // contig length; the `l` of fragment placement (region length with -x)
    int32_t start;                 // the contig's first cell in the group's coordinate space (a multiple of GROUP_ALIGN)
    const int32_t *reg_start, *reg_end; int32_t n_reg;
};
// haplotype h of the block's contig: cells / view pointers moved to the contig's first cell (start is a multiple of 32 cells: view chunks stay 16-byte
// aligned), the insertion tables stay those of the group (ins_find adds pos_off)
DW_DEV HapDev sel_hap(const SimArgs &a, const SegCtx &sc, int h)
{
    HapDev r;
    r.cells = (h ? a.hap[1].cells : a.hap[0].cells) + sc.start;
    r.view = (h ? a.hap[1].view : a.hap[0].view) + (sc.start >> 1);
    r.ins_pos = h ? a.hap[1].ins_pos : a.hap[0].ins_pos;
    r.ins_len = h ? a.hap[1].ins_len : a.hap[0].ins_len;
    r.ins_off = h ? a.hap[1].ins_off : a.hap[0].ins_off;
    r.ins_bases = h ? a.hap[1].ins_bases : a.hap[0].ins_bases;
    r.n_ins = h ? a.hap[1].n_ins : a.hap[0].n_ins;
    r.pos_off = sc.start;
    return r;
}
// The range (SimSeg) that logical block t of the launch belongs to: the last one whose first_block <= t.  t is block-uniform.
typedef const DW_CONST_AS SimSeg *SegPtr;       // the range table of a launch: uploaded before it, read with scalar loads
DW_DEV uint32_t seg_of_block(SegPtr segs, int32_t n_seg, uint32_t t)
{
    uint32_t lo = 0, hi = (uint32_t)n_seg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (segs[mid].first_block <= t) lo = mid; else hi = mid; }
    return lo;
}
DW_DEV SegCtx seg_ctx(const SimArgs &a, SegPtr sg)
{
    SegCtx sc; sc.l = sg->l; sc.l_place = sg->l_place; sc.start = sg->start;
    sc.reg_start = a.reg ? a.reg + sg->reg_off : nullptr; sc.reg_end = a.reg ? a.reg + sg->reg_off + sg->n_reg : nullptr; sc.n_reg = sg->n_reg;
    return sc;
}
DW_DEV int sel_len(const SimArgs &a, int j) { return j ? a.p.len[1] : a.p.len[0]; }

// dwgsim.c:649-742: random-read test, fragment size + position, haplotype, strands
DW_DEV PairDraw draw_pair(const SimArgs &a, const SegCtx &sc, RngKey key, uint64_t ii, uint32_t att)
{
    PairDraw pd; pd.pos = 0; pd.d = 0; pd.hap = 0; pd.strand0 = pd.strand1 = 0;
    const U4 b0 = rng_block(key, D_PAIR, ii, att, 0, 0);
    pd.is_rand = !(a.p.rand_read < u_lo(b0));
    if (pd.is_rand) return pd;
    const int s0 = a.p.len[0], s1 = a.p.len[1];
    const int64_t l = sc.l_place, sl = sc.l;          // placement length (region length with -x) vs contig length
    if (a.p.amplicons) { pd.pos = 0; pd.d = (int32_t)sl; }
    else {
        uint32_t t = 0; int32_t pos, d; bool continue_flag = false;
        do {
            if (s1 > 0 && probe::off(32)) d = a.p.dist;      // (analysis: the pair without its insert-size normal)
            else if (s1 > 0) {
                double v1, v2, rsq; uint32_t r = 0;
                do {
                    const U4 b = rng_block(key, D_PLACE_NORM, ii, att, r, t);
                    v1 = 2.0 * u_lo(b) - 1.0; v2 = 2.0 * u_hi(b) - 1.0;
                    rsq = v1 * v1 + v2 * v2; ++r;
                } while (rsq >= 1.0 || rsq == 0.0);
                // (the range-restricted forms give the same bits on their operand range -- dwgsim_hip_selftest_fp64; a radius below 2^-60 never
                // occurs in practice and takes the general ones)
                const double fac = rsq >= 0x1p-60 ? sqrt_mid(div_mid(-2.0 * det_log<true>(rsq), rsq)) : sqrt(-2.0 * det_log(rsq) / rsq);
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
                for (int q = 0; q < sc.n_reg; ++q) {
                    const int32_t jl = sc.reg_end[q] - sc.reg_start[q];
                    if (pos < jl) { pos = sc.reg_start[q] + pos - 1; break; }
                    pos -= jl;
                }
                inside = false;                           // regions are sorted and disjoint: "some region contains [pos, pos + d)" (regions_bed.c:130-156)
                int lo = 0, hi = sc.n_reg - 1;
                const uint32_t qs = (uint32_t)pos, qe = (uint32_t)(pos + d);
                while (lo <= hi) {
                    const int mid = lo + (hi - lo) / 2;
                    if (qs < (uint32_t)sc.reg_start[mid]) hi = mid - 1;
                    else if ((uint32_t)sc.reg_end[mid] < qe) lo = mid + 1;
                    else { inside = true; break; }
                }
            }
            ++t;
            // the reference would never terminate here; reported as an error by the caller.  Once one pair of the batch has given up the
            // call is lost anyway: the others stop at their next 1024th try instead of spinning to 2^20 each
            if (t > (1u << 20) || ((t & 1023u) == 0 && (__hip_atomic_load(&a.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4ull))) { pd.hap = -1; break; }
            continue_flag = !inside;
        } while (continue_flag || pos < 0 || pos >= sl || (int64_t)pos + d - 1 >= sl
                 || (s1 > 0 && !a.p.is_inner && ((s0 > 0 && d <= s1) || (d <= s0 && s1 > 0))));
        pd.pos = pos; pd.d = d;
    }
    // no placement satisfied the target regions: the call will return an error; this pair ends here as a (discarded) random read so that
    // the lane neither retries 10 000 times nor writes anything irregular
    if (pd.hap < 0) { pd.hap = 0; pd.pos = 0; pd.d = 0; pd.is_rand = true; atomicOr((unsigned long long *)&a.counters[2], 4ull); return pd; }
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
DW_DEV void read_geom(const SimArgs &a, const SegCtx &sc, const PairDraw &pd, int j, int64_t *start, int *step)
{
    const int64_t pos = pd.pos, d = pd.d, s0 = a.p.len[0], s1 = a.p.len[1], sl = sc.l;
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

// ---- Ion Torrent flow-space errors: dwgsim.c:246-417 generate_errors_flows ----
// The reference edits the read in place; both passes only ever insert/delete at the position being
// examined, so they are replayed as transducers over packed arrays in the block's global scratch (word w of a
// lane at base[w * stride]; flow_errors below says how).  Draws: narrow uniforms of domain D_FLOW0 + read end.  The
// flow mask is per read (the reference's persistent mask is fully rewritten by every read's pass 1).
// word-cached access to a lane's packed array (BITS = 4: codes 0-5): the event code looks at single bases
template <int BITS>
struct PackReader {
    static constexpr int PER = 32 / BITS, SH = BITS == 4 ? 3 : 4; static constexpr uint32_t M = (1u << BITS) - 1;
    const uint32_t *base; int stride, cw; uint32_t word;
    DW_DEV void init(const uint32_t *b, int st) { base = b; stride = st; cw = -1; word = 0; }
    DW_DEV uint32_t get(int i) { const int w = i >> SH; if (w != cw) { cw = w; word = base[w * stride]; } return (word >> ((i & (PER - 1)) * BITS)) & M; }
};
// First draws of the flow model's events as bits: bit k of the result = (first uniform of event 8 * blk + k) < e, e as thr = e * 2^32
// (dw_common.hpp D_FLOW0: sixteen-bit halves, the low halves drawn only when a high half ties with thr's).
DW_DEV uint32_t flow_hits8(RngKey key, uint32_t dom, uint64_t ii, uint32_t att, uint32_t blk, uint64_t thr)
{
    const uint32_t t_hi = (uint32_t)(thr >> 16), t_lo = (uint32_t)thr & 0xFFFFu;       // t_hi <= 0x10000
    const U4 b = rng_block(key, dom, ii, att, 0, blk);
    const uint32_t hw[8] = {b.x & 0xFFFFu, b.x >> 16, b.y & 0xFFFFu, b.y >> 16, b.z & 0xFFFFu, b.z >> 16, b.w & 0xFFFFu, b.w >> 16};
    uint32_t lt = 0, closest = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 8; ++k) { lt |= (hw[k] < t_hi ? 1u : 0u) << k; const uint32_t d = hw[k] ^ t_hi; closest = d < closest ? d : closest; }
    if (closest == 0 && t_lo) {      // a high half ties with the threshold's (probability 2^-13 per block): the low halves decide
        uint32_t eq = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) eq |= (hw[k] == t_hi ? 1u : 0u) << k;
        const U4 r = rng_block(key, dom + D_FLOW_REF, ii, att, 0, blk);
        const uint32_t lw[8] = {r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16, r.z & 0xFFFFu, r.z >> 16, r.w & 0xFFFFu, r.w >> 16};
#pragma unroll
        for (int k = 0; k < 8; ++k) lt |= (((eq >> k) & 1u) && lw[k] < t_lo ? 1u : 0u) << k;
    }
    return lt;
}
struct FlowRng {             // scalar members + value selects only: keeps the generator state in registers
    // The private stream of one event (its draws AFTER the first): draw s = word s & 3 of the block (retry s >> 2, block evt) of dom + D_FLOW_EV.
    uint32_t seed, contig, dom, att, evt, s, w0, w1, w2, w3; uint64_t ii;
    DW_DEV void open(uint32_t event) { evt = event; s = 0; }
    DW_DEV uint32_t next()
    {
        if ((s & 3) == 0) { const U4 b = rng_block(RngKey{seed, contig}, dom + D_FLOW_EV, ii, att, s >> 2, evt); w0 = b.x; w1 = b.y; w2 = b.z; w3 = b.w; }
        const uint32_t k = s & 3; ++s;
        const uint32_t lo = (k & 1) ? w1 : w0, hi = (k & 1) ? w3 : w2;
        return (k & 2) ? hi : lo;
    }
    // the rest of `while (drand48() < e) n_err++` (dwgsim.c:296, :373) after a first draw below e.  Bounded: with e = 1 the reference never
    // leaves this loop; 2^14 errors in one flow already overflow every buffer, so the caller reports the read as outgrown.
    DW_DEV int more_errors(uint64_t thr) { int n = 1; while ((uint64_t)next() < thr && n < (1 << 14)) ++n; return n; }
};
// A lane's packed array with word-wide access (word w at base[w * stride], nw words; indices outside read as zero):
// get8(p): the eight 4-bit codes at nibble positions p .. p + 7 (p may be negative), get8x2(p): eight 2-bit bases at positions p .. p + 7
struct WordView {
    const uint32_t *base; int stride, nw;
    DW_DEV uint32_t word(int w) const { return (w >= 0 && w < nw) ? base[(size_t)w * stride] : 0u; }
    DW_DEV uint32_t get8(int p) const { const int w = p >> 3; return __builtin_amdgcn_alignbit(word(w + 1), word(w), ((uint32_t)p & 7u) * 4u); }
    DW_DEV uint32_t get8x2(int p) const { const int w = p >> 4; return __builtin_amdgcn_alignbit(word(w + 1), word(w), ((uint32_t)p & 15u) * 2u) & 0xFFFFu; }
};
// Appends BITS-bit elements, one or up to 32 / BITS at a time (v: cnt elements, nothing above them), a word is stored whenever one is full
template <int BITS>
struct BitAppender {
    uint32_t *base; int stride, n, wi; uint32_t fill; uint64_t acc;
    DW_DEV void init(uint32_t *b, int st) { base = b; stride = st; n = 0; wi = 0; fill = 0; acc = 0; }
    DW_DEV void push_many(uint32_t v, int cnt)
    {
        acc |= (uint64_t)v << fill; fill += (uint32_t)cnt * BITS; n += cnt;
        if (fill >= 32u) { base[(size_t)wi * stride] = (uint32_t)acc; ++wi; acc >>= 32; fill -= 32u; }
    }
    DW_DEV void push(uint32_t v) { push_many(v, 1); }
    DW_DEV void flush() { if (fill) base[(size_t)wi * stride] = (uint32_t)acc; }
};
// A lane's bitmap of scoring first draws (word w at base[w * stride], nbits a multiple of 32): the first set bit at or after p, nbits if none
// (nbits: how far the bitmap has been drawn so far; a search that runs into that frontier returns it, or p when p is already beyond it)
struct HitMap {
    const uint32_t *base; int stride; uint32_t nbits;
    DW_DEV uint32_t next(uint32_t p) const
    {
        while (p < nbits) {
            const uint32_t w = base[(size_t)(p >> 5) * stride] >> (p & 31u);
            if (w) return p + (uint32_t)__ffs((int)w) - 1u;
            p = (p | 31u) + 1u;
        }
        return p > nbits ? p : nbits;
    }
};
DW_DEV uint32_t nibbles_reversed(uint32_t v) { v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu); return __builtin_bswap32(v); }
DW_DEV uint32_t nibbles_to_pairs(uint32_t v)      // eight nibbles holding 0 .. 3 -> sixteen bits
{
    v = (v | (v >> 2)) & 0x0F0F0F0Fu; v = (v | (v >> 4)) & 0x00FF00FFu; return (v | (v >> 8)) & 0xFFFFu;
}
DW_DEV uint32_t pairs_to_nibbles(uint32_t v)      // sixteen bits -> eight nibbles holding 0 .. 3
{
    v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu; return (v | (v << 2)) & 0x33333333u;
}
// dist[4 * f + b]: flows from flow f (inclusive) to the first flow of base b, 0 .. F-1 (filled by fill_flow_dist, every base occurs in the order)
DW_DEV void fill_flow_dist(const uint8_t *flow, int F, uint8_t *dist, int tid, int nthr)
{
    for (int q = tid; q < 4 * F; q += nthr) {
        const int f = q >> 2; const uint32_t b = (uint32_t)q & 3u;
        int k = 0, g = f;
        while (flow[g] != b) { ++k; g = g + 1 == F ? 0 : g + 1; }
        dist[q] = (uint8_t)k;
    }
}
// a batch of parked lanes runs once eight have gathered, or as many as are still running (measured on pass 2: 1 / 2 / 4 / 8 / 16 / 24 / 32 lanes ->
// 5.66 / 5.36 / 5.17 / 5.06 / 5.15 / 5.37 / 5.55 ms for 848 k reads of 400 bp at e = 0.01; waiting for the last runners alone costs 5 %)
#ifndef DW_FLOW_BATCH
#define DW_FLOW_BATCH 8
#endif
DW_DEV bool flow_batch_due(bool parked, bool running)
{
    const uint64_t p = __ballot(parked), r = __ballot(running);
    return p && (__popcll(p) >= DW_FLOW_BATCH || __popcll(p) >= __popcll(r));
}
// generate_errors_flows (dwgsim.c:246-417).  Every lane of the wave must call this (both passes regroup the lanes of a wave with ballots);
// lanes without a read pass active = false.  Returns the new length, -1 if a buffer / the pass-2 stack overflowed or the read degenerated.
// The final read is left in bufA (4-bit) in the orientation of the flow model; a reverse-strand read is turned back by the caller when it is
// read (dwgsim.c:408-414).  bufB: pass-1 output at 2 bits per base; stk: FLOW_STACK_RUNS (base, count) runs, two per word; bm: this lane's
// bitmap of scoring first draws (flow_hit_bits(cap) bits), of pass 1 first and then of pass 2.
//
// Both passes are sequential per read and almost always quiet: the first draw of a position (pass 1) or of an empty flow (pass 2) scores with
// probability e.  The first draws are made in step, as a bitmap, a word of 32 at a time just ahead of the lane that is furthest along
// (a read needs about len of pass 1's and 1.6 len of pass 2's; the capacity they are sized for is twice that, and a Philox block per
// eight draws is the largest single cost of the model); a lane then knows where its next scoring draw is and moves EIGHT
// bases per iteration up to it -- one word of the packed read, the flow pointer's chain of eight table look-ups, one append -- and only a lane
// standing on a scoring draw runs the event code.  With 64 lanes some lane scores in almost every iteration, and the event code (a Philox
// block of its own, homopolymer scans, the run stack) is long: such lanes park, the others run on, and the events are handled for a batch
// of parked lanes at once.  Each lane still performs exactly its own sequence of operations; only their interleaving changes.
DW_DEV int flow_errors(bool active, FlowRng &rg, const uint8_t *flow, const uint8_t *dist, int F, int maxk, uint64_t thr, uint32_t *bufA, uint32_t *bufB, uint32_t *bm, uint32_t *stk, int stride,
                       int len, int strand, int cap, int32_t *n_err_out)
{
    int total = 0, flow_i = 0; bool marked = false; bool failed = !active;
    const int G0 = flow_hit_bits(cap);
    auto draw_word = [&](uint32_t dom, uint32_t w) {           // 32 first draws, four Philox blocks, every lane in step
        if (active) {
            uint32_t bits = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) { if (probe::off(256)) break; bits |= flow_hits8(RngKey{rg.seed, rg.contig}, dom, rg.ii, rg.att, 4u * w + q, thr) << (8 * q); }
            if (probe::off(4096)) { probe::keep(bits); bits = 0; }
            bm[(size_t)w * stride] = bits;
        }
    };
    auto step_flow = [&](uint32_t k) { flow_i += (int)k; if (flow_i >= F) flow_i -= F; };

    // ---- pass 1 (dwgsim.c:253-364): one error event per homopolymer start whose first draw scores.  Input = bufA (len bases, read
    // back-to-front when strand == 1, N -> A: dwgsim.c:253-265), output -> bufB.  The output index is the reference's loop index i
    // (dwgsim.c:281): position i's first draw is bit i of the bitmap. ----
    // The reference's flow mask (dwgsim.c:283-333) never has more than one bit set: a deletion marks the flow the pointer stands on, and the
    // mark is cleared as soon as the pointer moves (the range of skipped flows starts at the pointer) or a new homopolymer starts on that flow --
    // and the pointer stands on the previous base's flow, so both mean "this base differs from the one before".  So the mask is one flag, and
    // what pass 2 sees is that flag together with the final pointer.
    const uint32_t D1_MAX = (uint32_t)((cap + 31) >> 5) << 5;
    const WordView inA{bufA, stride, (cap + 7) >> 3};
    PackReader<4> la; la.init(bufA, stride);
    auto in1 = [&](int t) -> uint32_t { const uint32_t v = la.get(strand ? len - 1 - t : t); return v >= 4 ? 0u : v; };
    auto in8 = [&](int t) -> uint32_t {                       // bases t .. t + 7 of the input as nibbles, N -> A
        uint32_t v = inA.get8(strand ? len - 8 - t : t);
        v = strand ? nibbles_reversed(v) : v;
        return v & 0x33333333u & ~(((v >> 2) & 0x11111111u) * 3u);
    };
    HitMap hm1{bm, stride, 0u};
    BitAppender<2> o1; o1.init(bufB, stride);
    int t = 0; uint32_t prev_c = 4, nh = 0;
    if (active) {
        const uint32_t c0 = in1(0);
        while (flow_i < F && c0 != flow[flow_i]) ++flow_i;
        if (flow_i == F) failed = true;
    }
    {
        bool done = failed, parked = false;
        for (;;) {
            // the bitmap stays ahead of every lane: a step looks at up to nine positions (an event that inserts more than that is waited for here)
            while (__ballot(!done && hm1.nbits < D1_MAX && (uint32_t)o1.n + 16u > hm1.nbits)) {
                const uint32_t old = hm1.nbits;
                draw_word(rg.dom, old >> 5);
                hm1.nbits = old + 32u;
                if (!done && nh >= old) nh = hm1.next((uint32_t)o1.n > old ? (uint32_t)o1.n : old);
            }
            if (!done && !parked) {
                if (t >= len) done = true;
                else if (o1.n >= cap) { failed = true; done = true; }
                else {
                    const uint32_t v = in8(t);
                    int n = len - t < 8 ? len - t : 8;
                    if (cap - o1.n < n) n = cap - o1.n;
                    if ((int)(nh - (uint32_t)o1.n) < n) n = (int)(nh - (uint32_t)o1.n);
                    uint32_t pc = prev_c; bool differs = false;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t c = (v >> (4 * i)) & 3u;
                        if (i < n) { step_flow(dist[4 * flow_i + (int)c]); differs = differs || c != pc; pc = c; }
                    }
                    if (differs) marked = false;
                    o1.push_many(nibbles_to_pairs(v) & ((1u << (2 * n)) - 1u), n);
                    t += n; prev_c = pc;
                    if (n < 8 && (uint32_t)o1.n == nh && t < len && o1.n < cap) {       // standing on a position whose first draw scored
                        if (((v >> (4 * n)) & 3u) != prev_c) parked = true;            // a homopolymer starts here: the event happens
                        else nh = hm1.next(nh + 1u);
                    }
                }
            }
            if (flow_batch_due(parked, !done && !parked)) {
                if (parked) {
                    const uint32_t c = in1(t);
                    step_flow(dist[4 * flow_i + (int)c]); marked = false;
                    rg.open((uint32_t)o1.n);
                    int n_err = rg.more_errors(thr);
                    if (n_err >= (1 << 14)) failed = true;
                    else if (rg.next() < 0x80000000u) {              // insert n_err copies in front of the homopolymer (whose own bases follow unexamined: prev_c == c)
                        if (o1.n + n_err > cap) failed = true;       // (the reference runs out of room at the latest when it reaches the homopolymer itself)
                        else { for (int q = 0; q < n_err; ++q) o1.push(c); total += n_err; prev_c = c; }
                    }
                    else {                                          // delete: bounded by the homopolymer length
                        int hp_l = 0; uint32_t next_c = c;
                        while (t + hp_l < len) { next_c = in1(t + hp_l); if (next_c != c) break; ++hp_l; }
                        if (n_err > hp_l) n_err = hp_l;
                        t += n_err; marked = true; total += n_err;
                        if (n_err == hp_l && (o1.n == 0 || prev_c == next_c)) {   // dot-fill (dwgsim.c:342-358)
                            if (next_c == c) failed = true;         // the whole read was one deleted homopolymer (the reference asserts)
                            else {
                                const int jj = dist[4 * flow_i + (int)next_c];
                                const int kk = (int)(((uint64_t)rg.next() * (uint64_t)jj) >> 32);   // (int)(drand48() * j)
                                int f = flow_i + kk; if (f >= F) f -= F;
                                o1.push(flow[f]);
                            }
                        } else if (t < len) { o1.push(in1(t)); ++t; }   // the base now at this position is not examined
                        prev_c = c;
                    }
                    if (failed) done = true; else nh = hm1.next((uint32_t)o1.n);
                    parked = false;
                }
            }
            if (__ballot(!done) == 0) break;
        }
    }
    o1.flush();
    const int n1 = o1.n;
    if (probe::off(1024)) { *n_err_out += total; return failed ? -1 : n1; }
    const int marked_flow = marked ? flow_i : -1;      // the one flow of the (persistent) mask that pass 2 finds set

    // ---- pass 2 (dwgsim.c:367-406): insertions in empty flows; inserted bases are examined again later, the examined base
    // itself stays behind them: a stack of (base, count) runs on top of the pass-1 output reproduces the in-place order.
    // g counts the empty flows examined so far: flow g's first draw is bit g of the bitmap (flows beyond the bitmap -- long cascades -- are
    // drawn one by one), nh is the next scoring flow: a base whose empty flows end at or before nh is quiet.  Lanes with an empty stack
    // move up to eight quiet bases per iteration, lanes with pending runs one; a base with a scoring flow in front of it parks. ----
    const uint32_t dom2 = rg.dom + D_FLOW_PASS2;
    HitMap hm2{bm, stride, 0u};
    const uint32_t margin2 = 8u * (uint32_t)maxk + 64u;      // a step moves g by at most eight bases' worth of flows
    const WordView inB{bufB, stride, (cap + 15) >> 4};
    BitAppender<4> o2; o2.init(bufA, stride);
    auto stk_get = [&](int k) -> uint32_t { return (stk[(k >> 1) * stride] >> ((k & 1) * 16)) & 0xffffu; };
    auto stk_set = [&](int k, uint32_t v) { const uint32_t sh = (uint32_t)(k & 1) * 16; uint32_t w = stk[(k >> 1) * stride]; stk[(k >> 1) * stride] = (w & ~(0xffffu << sh)) | (v << sh); };
    int t2 = 0, sp = 0;
    auto settle = [&](uint32_t x) {                        // the position's final base: the examined base, or the first base of the top run
        if (sp == 0) { o2.push(x); ++t2; }
        else {
            const uint32_t top = stk_get(sp - 1);
            o2.push(top >> 14);
            if ((top & 0x3fffu) <= 1) --sp; else stk_set(sp - 1, top - 1);
        }
    };
    rg.dom = dom2;
    {
        bool done = failed, parked = false; uint32_t g = 0, x = 0;
        nh = 0;
        for (;;) {
            while (__ballot(!done && hm2.nbits < (uint32_t)G0 && g + margin2 > hm2.nbits)) {
                const uint32_t old = hm2.nbits;
                draw_word(dom2, old >> 5);
                hm2.nbits = old + 32u;
                if (!done && nh >= old) nh = hm2.next(old);
            }
            if (!done && !parked) {
                if (sp == 0 && t2 >= n1) done = true;
                else if (o2.n >= cap) { failed = true; done = true; }
                else if (sp > 0) {                                  // runs pending: one base
                    x = stk_get(sp - 1) >> 14;
                    const uint32_t k = dist[4 * flow_i + (int)x];
                    if (g + k <= nh) { step_flow(k); g += k; settle(x); } else parked = true;
                }
                else {
                    const uint32_t v = inB.get8x2(t2);
                    int n = n1 - t2 < 8 ? n1 - t2 : 8;
                    if (cap - o2.n < n) n = cap - o2.n;
                    int m = 0; bool quiet = true;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t k = dist[4 * flow_i + (int)((v >> (2 * i)) & 3u)];
                        quiet = quiet && i < n && g + k <= nh;
                        if (quiet) { step_flow(k); g += k; ++m; }
                    }
                    o2.push_many(pairs_to_nibbles(v) & (m == 8 ? 0xFFFFFFFFu : (1u << (4 * m)) - 1u), m);
                    t2 += m;
                    if (m < n) { x = (v >> (2 * m)) & 3u; parked = true; }
                }
            }
            if (flow_batch_due(parked, !done && !parked)) {
                if (parked) {
                    uint32_t left = dist[4 * flow_i + (int)x];      // empty flows in front of x still to examine
                    while (!failed && left > 0) {
                        uint32_t skip; bool scores;                 // quiet flows before the next scoring one
                        if (g < (uint32_t)G0) { const uint32_t q = nh - g; skip = q < left ? q : left; scores = q < left && nh < hm2.nbits; }
                        else {                                      // beyond the bitmap (a long cascade): flow by flow
                            skip = 0;
                            while (skip < left && !((flow_hits8(RngKey{rg.seed, rg.contig}, dom2, rg.ii, rg.att, (g + skip) >> 3, thr) >> ((g + skip) & 7u)) & 1u)) ++skip;
                            scores = skip < left;
                        }
                        step_flow(skip);
                        g += skip; left -= skip;
                        if (!scores) continue;                      // (all of them looked at, or the bitmap ended: on beyond it)
                        rg.open(g);                                 // flow g scores: while (drand48() < e) n_err++ goes on in its private stream
                        const int n_err = rg.more_errors(thr);
                        if (flow_i != marked_flow) {
                            if (sp >= FLOW_STACK_RUNS || n_err >= (1 << 14)) failed = true;
                            else { stk_set(sp, ((uint32_t)flow[flow_i] << 14) | (uint32_t)n_err); ++sp; total += n_err; }
                        }
                        step_flow(1);
                        ++g; --left;
                        if (g <= (uint32_t)G0) nh = hm2.next(g);
                    }
                    if (failed) done = true; else settle(x);
                    parked = false;
                }
            }
            if (__ballot(!done) == 0) break;
        }
    }
    if (failed) return -1;
    o2.flush();
    *n_err_out += total;
    return o2.n;
}

// ---- FASTQ text assembly ----
// (probe::off / probe::keep: dw_probe.hpp -- nothing in the product build)
struct __attribute__((packed, aligned(1))) Unal16 { uint64_t a, b; };      // stores at any byte address: gfx950 global memory takes unaligned
struct __attribute__((packed, aligned(1))) Unal8 { uint64_t v; };           // dword / dwordx2 / dwordx4 accesses as they are (one instruction)
struct __attribute__((packed, aligned(1))) Unal4 { uint32_t v; };
struct __attribute__((packed, aligned(1))) Unal2 { uint16_t v; };
struct Writer {               // sequential byte stream of one record -> 16-byte stores at the record's own (arbitrary) byte phase
    // The bytes are gathered in lo / hi RELATIVE TO THE SECTION START p, so whole groups (16 bases, 16 qualities) go out with no funnel
    // shift at all: put16 is one unaligned dwordx4 store.  flush() writes the ragged rest (8 + 4 + 2 + 1 bytes) and starts a new section
    // at the next byte, which is how the callers re-phase before the base line and before the quality line.
    uint8_t *p; uint64_t lo, hi; uint32_t n;
    DW_DEV void init(uint8_t *q) { p = q; lo = hi = 0; n = 0; }
    DW_DEV void init(uint8_t *, uint8_t *q) { init(q); }
    DW_DEV void emit()
    {
        if (probe::off(2)) probe::keep(lo, hi, p);
        else { Unal16 v; v.a = lo; v.b = hi; *reinterpret_cast<Unal16 *>(p) = v; }
        p += 16; lo = hi = 0; n = 0;
    }
    DW_DEV void put(uint32_t b)
    {
        if (probe::off(1)) return;
        if (probe::off(64)) { probe::keep(b); return; }
        const uint64_t v = (uint64_t)b << (8 * (n & 7));
        if (n < 8) lo |= v; else hi |= v;
        if (++n == 16) emit();
    }
    DW_DEV void putn(uint64_t v, uint32_t cnt)   // cnt (1..8) bytes, little-endian in v, upper bytes zero
    {
        if (probe::off(1)) return;
        if (probe::off(64)) { probe::keep(v, cnt); return; }
        const uint32_t sh = 8 * (n & 7);
        if (n < 8) { lo |= v << sh; if (sh) hi |= v >> (64 - sh); }
        else hi |= v << sh;
        const uint32_t total = n + cnt;
        if (total >= 16) {
            const uint32_t over = total - 16;       // bytes that belong to the next 16 (0..7)
            const uint64_t carry = over ? v >> (8 * (cnt - over)) : 0;
            emit();
            lo = carry; n = over;
        } else n = total;
    }
    DW_DEV void put4(uint32_t w) { putn((uint64_t)w, 4); }
    DW_DEV void put16(uint32_t a, uint32_t b, uint32_t c, uint32_t d)       // sixteen bytes; one store when the section stands at a multiple of 16
    {
        if (probe::off(1)) return;
        if (probe::off(64)) { probe::keep(a, b, c, d); return; }
        const uint64_t x = (uint64_t)a | ((uint64_t)b << 32), y = (uint64_t)c | ((uint64_t)d << 32);
        if (n == 0) { lo = x; hi = y; emit(); }
        else { putn(x, 8); putn(y, 8); }
    }
    DW_DEV void flush()                          // the n < 16 gathered bytes; the next byte starts a new section
    {
        if (probe::off(2)) { probe::keep(lo, hi, p, n); p += n; lo = hi = 0; n = 0; return; }
        if (n & 8) { Unal8 v; v.v = lo; *reinterpret_cast<Unal8 *>(p) = v; p += 8; lo = hi; }
        if (n & 4) { Unal4 v; v.v = (uint32_t)lo; *reinterpret_cast<Unal4 *>(p) = v; p += 4; lo >>= 32; }
        if (n & 2) { Unal2 v; v.v = (uint16_t)lo; *reinterpret_cast<Unal2 *>(p) = v; p += 2; lo >>= 16; }
        if (n & 1) { *p = (uint8_t)lo; p += 1; }
        lo = hi = 0; n = 0;
    }
};
// The record writer of the primary output: a 40-byte FIFO per lane in LDS.  Bytes are appended at any byte position with plain (unaligned)
// LDS stores of up to 8 bytes -- the LDS does the byte shifting, no VALU -- and leave as 32-byte ALIGNED bursts (two dwordx4 stores): the L2
// of gfx950 does not merge a lane's pieces over time (41 k lanes per XCD write 41 k different lines), so what leaves the L2 is 32 bytes per
// touched sector per burst: 16-byte pieces cost 2.5x-3.3x the text, aligned 32-byte bursts 1.1x (tools/ubench_write_bursts.hip,
// profiles/r02_ubench_write_bursts.txt).  FIFO position 0 always stands for the 32-byte aligned address dst; a record starts at position
// skip = its address mod 32.  40 bytes, not 48 with 16-byte appends: LDS is what limits the blocks per CU (5 at 2 x 150 bp).
// BURST = 64 (a 72-byte FIFO): where LDS does not bound residency -- the second half of the two-kernel form, which stages no bases -- the text
// leaves as aligned 64-byte bursts: whole pairs of sectors, 1.02x the text instead of 1.15x-1.5x.
template <uint32_t BURST = 32u>
struct FifoWriter {
    uint8_t *f, *dst; uint32_t wp, skip;
    DW_DEV void init(uint8_t *fifo, uint8_t *rec) { const uint32_t h = (uint32_t)((uintptr_t)rec & (BURST - 1u)); f = fifo; dst = rec - h; wp = skip = h; }
    DW_DEV uint64_t ld8(uint32_t b) const { return *reinterpret_cast<const uint64_t *>(f + b); }
    DW_DEV void st16(uint32_t b) const { *reinterpret_cast<uint4 *>(dst + b) = make_uint4((uint32_t)ld8(b), (uint32_t)(ld8(b) >> 32), (uint32_t)ld8(b + 8), (uint32_t)(ld8(b + 8) >> 32)); }
    DW_DEV void store_range(uint32_t from, uint32_t upto)        // bytes [from, upto) of the unit (ragged first / last unit of a record), from LDS
    {
        uint32_t b = from;       // rising sizes until b is aligned (or the next piece would pass upto), then falling sizes
        if ((b & 1u) && b + 1 <= upto) { dst[b] = f[b]; b += 1; }
        if ((b & 2u) && b + 2 <= upto) { *reinterpret_cast<uint16_t *>(dst + b) = *reinterpret_cast<const uint16_t *>(f + b); b += 2; }
        if ((b & 4u) && b + 4 <= upto) { *reinterpret_cast<uint32_t *>(dst + b) = *reinterpret_cast<const uint32_t *>(f + b); b += 4; }
        if ((b & 8u) && b + 8 <= upto) { *reinterpret_cast<uint64_t *>(dst + b) = ld8(b); b += 8; }
        if (b + 16 <= upto) { st16(b); b += 16; }
        if (BURST == 64u) { if (b + 16 <= upto) { st16(b); b += 16; } if (b + 16 <= upto) { st16(b); b += 16; } }
        if (b + 8 <= upto) { *reinterpret_cast<uint64_t *>(dst + b) = ld8(b); b += 8; }
        if (b + 4 <= upto) { *reinterpret_cast<uint32_t *>(dst + b) = *reinterpret_cast<const uint32_t *>(f + b); b += 4; }
        if (b + 2 <= upto) { *reinterpret_cast<uint16_t *>(dst + b) = *reinterpret_cast<const uint16_t *>(f + b); b += 2; }
        if (b + 1 <= upto) { dst[b] = f[b]; }
    }
    DW_DEV void drain()                          // wp >= BURST: one unit leaves
    {
        if (!(probe::off(2))) {
            if (skip == 0) { st16(0); st16(16); if (BURST == 64u) { st16(32); st16(48); } }
            else store_range(skip, BURST);
        }
        skip = 0;
        *reinterpret_cast<uint64_t *>(f) = ld8(BURST);      // the (< 8) bytes past the unit move to the front
        dst += BURST; wp -= BURST;
    }
    DW_DEV void put(uint32_t b) { if (probe::off(1)) return; f[wp] = (uint8_t)b; if (++wp >= BURST) drain(); }
    DW_DEV void putn(uint64_t v, uint32_t cnt)   // cnt (1..8) bytes; the bytes above them are overwritten by the next put
    {
        if (probe::off(1)) return;
        Unal8 x; x.v = v; *reinterpret_cast<Unal8 *>(f + wp) = x;
        wp += cnt; if (wp >= BURST) drain();
    }
    DW_DEV void put4(uint32_t w) { if (probe::off(1)) return; Unal4 x; x.v = w; *reinterpret_cast<Unal4 *>(f + wp) = x; wp += 4; if (wp >= BURST) drain(); }
    DW_DEV void put16(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { putn((uint64_t)a | ((uint64_t)b << 32), 8); putn((uint64_t)c | ((uint64_t)d << 32), 8); }
    DW_DEV void flush() { if (wp > skip && !(probe::off(2))) store_range(skip, wp); dst += wp; wp = skip = 0; }
};
// OUT bit 0: the bwa stream of this read end, bit 1: the interleaved bfast stream.  The (first) output goes through the FIFO writer (WR = 1)
// or the register writer (WR = 0: the host found that the FIFO's LDS would cost a resident block per CU); with both outputs (-o 0) the
// bfast stream always takes the register writer.
template <int WR> struct PrimaryWriter { typedef FifoWriter<32u> type; };      // WR = 1
template <> struct PrimaryWriter<0> { typedef Writer type; };
template <> struct PrimaryWriter<2> { typedef FifoWriter<64u> type; };
template <int OUT, int WR = 1>
struct Out2 {
    typename PrimaryWriter<WR>::type a; Writer b;
    DW_DEV void init(uint8_t *fifo, uint8_t *bwa, uint8_t *bfast) { a.init(fifo, (OUT & 1) ? bwa : bfast); if (OUT == 3) b.init(bfast); }
    DW_DEV void put(uint32_t c) { a.put(c); if (OUT == 3) b.put(c); }
    DW_DEV void put4(uint32_t w) { a.put4(w); if (OUT == 3) b.put4(w); }
    DW_DEV void putn(uint64_t v, uint32_t cnt) { a.putn(v, cnt); if (OUT == 3) b.putn(v, cnt); }
    DW_DEV void put16(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { a.put16(x, y, z, w); if (OUT == 3) b.put16(x, y, z, w); }
    // the end of the name line differs between the two families: "/1\n" (bwa) and "\n" (bfast)
    DW_DEV void put_suffix(uint64_t v_bwa, uint32_t n_bwa, uint64_t v_bf, uint32_t n_bf)
    {
        if (OUT & 1) a.putn(v_bwa, n_bwa); else a.putn(v_bf, n_bf);
        if (OUT == 3) b.putn(v_bf, n_bf);
    }
    DW_DEV void rebase() { if (WR == 0) a.flush(); if (OUT == 3) b.flush(); }      // a new section of the register writer(s); the FIFO needs none
    DW_DEV void flush() { a.flush(); if (OUT == 3) b.flush(); }
};
DW_DEV uint32_t ndigits10(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}
DW_DEV uint32_t ndigits16(uint64_t v) { return v ? (uint32_t)(67 - __clzll((long long)v)) >> 2 : 1u; }
// decimal digits of v as packed ASCII, most significant digit in the lowest byte (stream order);
// lead = one separator byte to emit in front (0 = none).  Numbers above 10^7 take the two-part path.
template <class O>
DW_DEV void put_dec(O &o, uint32_t v, uint32_t lead)
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
template <class O>
DW_DEV void put_hex(O &o, uint64_t v)
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
DW_DEV uint32_t base_chars4(uint32_t nibbles) { return lut8(0x4E4E4E4Eu, 0x54474341u, spread4(nibbles)); }        // four codes (<= 7) -> "ACGTNNNN"[code]
DW_DEV uint32_t colour_digits4(uint32_t nibbles) { return lut8(0x34343434u, 0x33323130u, spread4(nibbles)); }     // four colours -> "01234444"[colour]

} // namespace dw
