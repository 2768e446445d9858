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

// dwgsim.c:75-153 __gen_read.  The packed 4-bit bases go to `sink`, eight at a time.
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
// where the extracted bases go: put(kw, codes, n) receives word kw of the read -- bases 8 kw .. 8 kw + n - 1 as 4-bit codes (0-3 ACGT, 4 N), nothing
// above them -- in rising order of kw, each word once
struct NoSink { DW_DEV void put(int, uint32_t, int) const {} };                                           // k_place: only the verdict is wanted
struct WordSink { uint32_t *lds; int stride; DW_DEV void put(int kw, uint32_t codes, int) const { lds[kw * stride] = codes; } };      // word kw of this lane at lds[kw * stride]
DW_DEV uint32_t nibbles_to_pairs(uint32_t v)      // eight nibbles holding 0 .. 3 -> sixteen bits
{
    v = (v | (v >> 2)) & 0x0F0F0F0Fu; v = (v | (v >> 4)) & 0x00FF00FFu; return (v | (v >> 8)) & 0xFFFFu;
}
DW_DEV uint32_t pairs_to_nibbles(uint32_t v)      // sixteen bits -> eight nibbles holding 0 .. 3
{
    v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu; return (v | (v << 2)) & 0x33333333u;
}
DW_DEV uint32_t reverse_pairs16(uint32_t v)       // the eight 2-bit fields of v[15:0] in reverse order
{
    v = __builtin_bitreverse32(v) >> 16;          // bits reversed: the fields are in place, each with its two bits swapped
    return ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
}
// Ion Torrent: the flow model (flow_errors below) works on 2-bit bases with N read as A (dwgsim.c:253-257), in place, on ONE buffer per lane: base
// position p of the buffer = pair p & 15 of word p >> 4 (at buf[(p >> 4) * stride]).  A forward read goes to positions h0 * 8 .. in order; a
// reverse-strand read is stored turned round (dwgsim.c:259-265: reversed, not complemented) with its last extracted base at the buffer's top.
struct FlowSink {
    uint32_t *buf; int stride, h0; bool rev;      // h0: the 16-bit piece (eight bases) that word 0 goes to; rev: pieces fall from there, each turned round
    DW_DEV void put(int kw, uint32_t codes, int) const
    {
        uint32_t p = nibbles_to_pairs(codes & 0x33333333u & ~(((codes >> 2) & 0x11111111u) * 3u));
        if (rev) p = reverse_pairs16(p);
        const int h = rev ? h0 - kw : h0 + kw;
        reinterpret_cast<uint16_t *>(buf + (h >> 1) * stride)[h & 1] = (uint16_t)p;
    }
};
template <class Sink>
DW_DEV ReadRes gen_read(const HapDev &h, int64_t l, int64_t start, int step, int s, int strand, const Sink &sink)
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
        accw |= (v > 4u ? 4u : v) << (4 * na); ++k;     // (a '-' cell, code 5: not an N for the filter above, clamped as the error loop does: `if (c >= 4) c = 4`, dwgsim.c:235)
        if (++na == 8) { sink.put((k >> 3) - 1, accw, 8); accw = 0; na = 0; }
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
                    sink.put(k >> 3, codes, want);
                    k += want; i += fwd ? want : -want;
                }
            }
        }
        if (!esc) continue;
        // ---- an episode of the reference's per-cell logic, until the output stands at a word boundary again
        // (episodes) the sixteen byte cells around the one in hand, fetched together: the per-cell logic below walks a handful of neighbouring cells, and a load per cell
        // was a memory round trip per cell -- on the path of every block behind this one (dw_simulate.hip, "one look-back").  h.cells is 16-byte aligned (a contig starts at
        // a multiple of GROUP_ALIGN cells), its buffer padded behind the last contig (CELL_PAD)
        // (fetched ONCE, where the episode starts: sixteen cells in travel order from cell i -- an episode rarely leaves them, and then falls back to a load per cell)
        const int32_t ck_base = (i < 0 || i >= li) ? 0 : fwd ? (i & ~3) : ((i | 3) - 15 < 0 ? 0 : (i | 3) - 15);      // 4-byte aligned, inside the buffer (padded behind, CELL_PAD)
        const ViewQuad ckq = *reinterpret_cast<const ViewQuad *>(h.cells + ck_base);
        const uint32_t ck0 = ckq.a, ck1 = ckq.b, ck2 = ckq.c, ck3 = ckq.d;
        const uint8_t *const cellp = h.cells;
        auto cell_at = [=](int32_t q) -> uint32_t {      // (captures by VALUE: a choice between references is a choice between addresses, and the four words would live in memory)
            const uint32_t o = (uint32_t)(q - ck_base);
            if (o >= 16u) return cellp[q];
            const uint32_t w01 = (o & 4u) ? ck1 : ck0, w23 = (o & 4u) ? ck3 : ck2;
            return (((o & 8u) ? w23 : w01) >> (8u * (o & 3u))) & 0xffu;
        };
        for (;;) {
            if (i < 0 || i >= li) { k = -1; break; }                 // walked off the contig before the read was complete
            const uint32_t c = cell_at(i), mt = c & TMASK;
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
    if (na) sink.put(k >> 3, accw, na);      // the ragged last word of an episode
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
// The reference edits the read in place with memmoves; both passes only ever insert / delete at the position being examined, so they are replayed
// as transducers over ONE packed buffer per lane, 2 bits per base (the model reads N as A, dwgsim.c:253-257, and never produces anything but
// ACGT), IN PLACE: the input of a pass stands right-aligned at the buffer's top, its output grows from position 0 and can never catch up with the
// read pointer while the read fits the buffer (output - input = net growth <= room).  So a lane's whole state is cap / 4 bytes -- in LDS (word w of
// a lane at buf[w * stride]) -- where rounds 1-4 kept a 4-bit buffer, a 2-bit buffer and two bitmaps of first draws per lane in a global scratch of
// 150 KB per block (5 x the algorithmic HBM traffic, profiles/r04_ion_*).
// Draws: dw_common.hpp D_FLOW0.  The flow mask is per read (the reference's persistent mask is fully rewritten by every read's pass 1).
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
// A lane's 2-bit packed buffer (word w at base[w * stride], nw words; a word beyond the end reads as zero)
struct Buf2 {
    const uint32_t *base; int stride, nw;
    DW_DEV uint32_t word(int w) const { return w < nw ? base[w * stride] : 0u; }
    DW_DEV uint32_t get8(int p) const      // the eight bases at positions p .. p + 7 (p >= 0) as sixteen bits
    {
        const int w = p >> 4; const uint32_t sh = ((uint32_t)p & 15u) * 2u;
        const uint32_t lo = word(w), hi = sh > 16u ? word(w + 1) : 0u;
        return __builtin_amdgcn_alignbit(hi, lo, sh) & 0xFFFFu;
    }
    DW_DEV uint32_t get16(int p) const     // the sixteen bases at positions p .. p + 15
    {
        const int w = p >> 4; const uint32_t sh = ((uint32_t)p & 15u) * 2u;
        const uint32_t lo = word(w), hi = sh ? word(w + 1) : 0u;
        return __builtin_amdgcn_alignbit(hi, lo, sh);
    }
    DW_DEV uint32_t get1(int p) const { return (word(p >> 4) >> (((uint32_t)p & 15u) * 2u)) & 3u; }
};
// Appends BITS-bit elements, one or up to 32 / BITS at a time (v: cnt elements, nothing above them), a word is stored whenever one is full
template <int BITS>
struct BitAppender {
    uint32_t *base; int stride, n, wi; uint32_t fill; uint64_t acc;
    DW_DEV void init(uint32_t *b, int st) { base = b; stride = st; n = 0; wi = 0; fill = 0; acc = 0; }
    DW_DEV void push_many(uint32_t v, int cnt)
    {
        acc |= (uint64_t)v << fill; fill += (uint32_t)cnt * BITS; n += cnt;
        if (fill >= 32u) { base[wi * stride] = (uint32_t)acc; ++wi; acc >>= 32; fill -= 32u; }
    }
    DW_DEV void push(uint32_t v) { push_many(v, 1); }
    DW_DEV void flush() { if (fill) base[wi * stride] = (uint32_t)acc; }
};
// The FIRST draws of a pass -- `drand48() < e` once per homopolymer start (pass 1, dwgsim.c:296) / per empty flow (pass 2, :373) -- are a Bernoulli(e')
// sequence, e' = thr / 2^32; 99 % of them say "no".  Rounds 2-5 drew them one by one (a Philox block per eight, a 64-event window of hit bits per lane):
// 170 blocks per 400-base read, two or three per loop iteration, and most of the flow model's 69.6 k VALU instructions per wave went into drawing and
// comparing numbers that say "no" (profiles/r05_ion_chr20_kernel_stats_pmc.txt).  In law such a sequence IS its gaps: the quiet draws in front of each
// scoring one are Geometric(e'), independent.  So the gaps are drawn (dw_common.hpp D_FLOW0: gap m = word m & 3 of block m >> 2 of the pass's domain),
//     G = floor(-log2(U) / -log2(1 - e')),  U = (2 w + 1) / 2^33,
// in integer arithmetic only -- a 257-entry table of log2 with linear interpolation, fixed point, one 64 x 64 -> 128 multiplication by the reciprocal
// the host made (dw_kernels.hpp flow_gap_params; the tests' CPU restatement has the same text) -- and a lane knows the ORDINAL of its next
// scoring first draw: a dozen gaps per read instead of 1 400 uniforms.  (The unmodified reference replays it all the same: tests/replay_common.py.)
struct FlowGap {
    uint32_t next, m, w0, w1, w2, w3;       // ordinal of the next scoring first draw (FLOW_NEVER: none); gaps drawn so far; the words of the current block
    // the next gap: the scoring draw it leads to has ordinal base + G (base = the ordinal after the one that just scored, 0 at the start of a pass)
    DW_DEV void draw(bool go, RngKey key, uint32_t dom, uint64_t ii, uint32_t att, uint32_t base, uint64_t thr, const uint32_t *lg, uint64_t R, int sR)
    {
        if (!go) return;
        if (thr == 0) { next = FLOW_NEVER; return; }
        if ((m & 3u) == 0u) { const U4 b = rng_block(key, dom, ii, att, 0, m >> 2); w0 = b.x; w1 = b.y; w2 = b.z; w3 = b.w; }
        const uint32_t k = m & 3u; ++m;
        const uint32_t w = (k & 2u) ? ((k & 1u) ? w3 : w2) : ((k & 1u) ? w1 : w0);
        next = base + (thr >= 0x100000000ull ? 0u : geom_gap(w, lg, R, sR));
    }
};
// what a scoring event goes on to draw (dw_common.hpp D_FLOW_EV: further errors, insert-or-delete, the dot-fill flow): one Philox block, block `pos`
// oc: n_err (1 or 2) | 0x100 insert | 0x200 the event draws on (n_err >= 3): taken from FlowRng where it happens
struct FlowEvent {
    uint32_t oc, oc_dot;
    DW_DEV void resolve(RngKey key, uint32_t dom, uint64_t ii, uint32_t att, uint32_t pos, uint64_t thr)
    {
        const U4 b = rng_block(key, dom + D_FLOW_EV, ii, att, 0, pos);
        const bool m0 = (uint64_t)b.x < thr, m1 = (uint64_t)b.y < thr;                // while (drand48() < e) n_err++ (dwgsim.c:296, :373): draws 0, 1, ...
        const uint32_t insw = m0 ? b.z : b.y;                                         // the draw after the first failing one: insert or delete (dwgsim.c:299)
        oc_dot = m0 ? b.w : b.z;                                                      // ... and the one after that: the dot-fill flow (dwgsim.c:352)
        oc = (m0 ? 2u : 1u) | (insw < 0x80000000u ? 0x100u : 0u) | ((m0 && m1) ? 0x200u : 0u);
    }
};
// position of the r-th (0-based) set bit of a sixteen-bit mask that has more than r set bits
DW_DEV int select_bit16(uint32_t m, uint32_t r)
{
    int pos = 0;
    uint32_t c = (uint32_t)__popc(m & 0xFFu); if (r >= c) { r -= c; m >>= 8; pos += 8; }
    c = (uint32_t)__popc(m & 0xFu); if (r >= c) { r -= c; m >>= 4; pos += 4; }
    c = (uint32_t)__popc(m & 0x3u); if (r >= c) { r -= c; m >>= 2; pos += 2; }
    c = m & 1u; if (r >= c) pos += 1;
    return pos;
}
// The flow order as tables (LDS, filled once per block by fill_flow_tables; every base occurs in the order, F <= 64):
//   dist[4 f + b]   flows from flow f (inclusive) to the first flow of base b, 0 .. F-1;   next1[4 f + b] = that flow
//   pair[16 f + (b0 | b1 << 2)]   TWO bases at once: the flow after both in bits 0-5, the flows passed over in front of them (k0 + k1) in bits 8-14
// The flow pointer is a chain of dependent table look-ups, one per base of the read (the only part of the model that is serial base by base); the
// pair table halves its length.
//   lg[257]         floor(2^32 log2(1 + i / 256)): what FlowGap::gap_of interpolates in (made by the host, behind the flow order in the same device buffer)
struct FlowTables { uint8_t flow[64], dist[256], next1[256]; uint16_t pair[1024]; uint32_t lg[FLOW_LG_ENTRIES]; };
DW_DEV void fill_flow_tables(FlowTables &T, int F, int tid, int nthr, const uint8_t *flow_dev)      // (T.flow is in place; a barrier before and after)
{
    for (int q = tid; q < FLOW_LG_ENTRIES; q += nthr) T.lg[q] = reinterpret_cast<const uint32_t *>(flow_dev + 64)[q];
    for (int q = tid; q < 4 * F; q += nthr) {
        const int f = q >> 2; const uint32_t b = (uint32_t)q & 3u;
        int k = 0, g = f;
        while (T.flow[g] != b) { ++k; g = g + 1 == F ? 0 : g + 1; }
        T.dist[q] = (uint8_t)k; T.next1[q] = (uint8_t)g;
    }
    for (int q = tid; q < 16 * F; q += nthr) {
        const int f = q >> 4; const uint32_t b0 = (uint32_t)q & 3u, b1 = ((uint32_t)q >> 2) & 3u;
        int k = 0, g = f;
        while (T.flow[g] != b0) { ++k; g = g + 1 == F ? 0 : g + 1; }
        while (T.flow[g] != b1) { ++k; g = g + 1 == F ? 0 : g + 1; }
        T.pair[q] = (uint16_t)((uint32_t)g | ((uint32_t)k << 8));
    }
}
DW_DEV uint32_t even_bits16(uint32_t x)      // bits 0, 2, 4, .. 30 of x gathered into sixteen bits
{
    x &= 0x55555555u; x = (x | (x >> 1)) & 0x33333333u; x = (x | (x >> 2)) & 0x0F0F0F0Fu; x = (x | (x >> 4)) & 0x00FF00FFu; return (x | (x >> 8)) & 0xFFFFu;
}
// generate_errors_flows (dwgsim.c:246-417).  Every lane of the wave must call this (the loops are wave-uniform); lanes without a read pass
// active = false.  buf: this lane's buffer of capb bases (a multiple of 16; capb / 16 words).  On entry the read stands where FlowSink put it:
// forward at positions ((capb - len) & ~7) .. , reverse-strand -- already turned round -- at capb - len .. capb - 1.  On return the read after
// errors stands at positions 0 .. (result - 1) in the orientation of the flow model; a reverse-strand read is turned back by the caller when it
// is written (dwgsim.c:408-414).  Returns the new length, or -1 if the read outgrew the buffer / the pass-2 stack (stack_runs (base, count) runs,
// two per word of stk) or degenerated.
//
// Both passes are sequential per read and almost always quiet.  A lane knows the ordinal of its next scoring first draw (FlowGap) and moves SIXTEEN
// bases per step up to it -- one word of the packed read, the flow pointer's chain of eight pair look-ups, one append.  A lane that reaches its
// scoring draw PARKS; the event code (the event's further draws, insert / delete / dot-fill, the run stack of pass 2, the next gap) runs for the
// parked lanes together, when a quarter of the wave stands parked or no lane can step any more: in rounds 3-5 some lane stood on an event in nearly
// every iteration and the whole wave went through the event code every time.  Each lane performs exactly its own sequence of operations; only their
// interleaving changes.
#ifndef DW_FLOW_EVENT_BATCH
#define DW_FLOW_EVENT_BATCH 16
#endif
constexpr int FLOW_EVENT_BATCH = DW_FLOW_EVENT_BATCH;      // parked lanes of a wave that make an event round worth its instructions
DW_DEV int flow_errors(bool active, FlowRng &rg, const FlowTables &T, int F, uint64_t thr, uint64_t gap_R, int gap_s, uint32_t *buf, uint32_t *stk, int stride, int stack_runs,
                       int len, int strand, int capb, int32_t *n_err_out)
{
    const RngKey key{rg.seed, rg.contig};
    int total = 0; uint32_t flow_i = 0; bool marked = false; bool failed = !active;
    const int capw = capb >> 4;
    const Buf2 B{buf, stride, capw};
    FlowGap G; FlowEvent E;

    // ---- pass 1 (dwgsim.c:253-364): one error event per homopolymer start whose first draw scores.  k1 counts the homopolymer starts examined so far:
    // the next one's first draw is first draw number k1 of the pass. ----
    // The reference's flow mask (dwgsim.c:283-333) never has more than one bit set: a deletion marks the flow the pointer stands on, and the
    // mark is cleared as soon as the pointer moves (the range of skipped flows starts at the pointer) or a new homopolymer starts on that flow --
    // and the pointer stands on the previous base's flow, so both mean "this base differs from the one before".  So the mask is one flag, and
    // what pass 2 sees is that flag together with the final pointer.
    const int in0 = strand ? capb - len : (capb - len) & ~7;      // where base 0 of the input stands (FlowSink)
    BitAppender<2> o1; o1.init(buf, stride);
    int t = 0; uint32_t prev_c = 4;
    if (active) {
        const uint32_t c0 = B.get1(in0);
        while (flow_i < (uint32_t)F && c0 != T.flow[flow_i]) ++flow_i;
        if (flow_i == (uint32_t)F) failed = true;
    }
    G.m = 0; G.next = FLOW_NEVER; G.w0 = G.w1 = G.w2 = G.w3 = 0;
    G.draw(__ballot(!failed) != 0, key, rg.dom, rg.ii, rg.att, 0u, thr, T.lg, gap_R, gap_s);
    {
        bool done = failed, parked = false; uint32_t k1 = 0;
        for (;;) {
            // ---- a step: up to sixteen positions, up to the homopolymer start whose first draw scores
            const bool act = !done && !parked;
            if (act && t >= len) done = true;
            const bool go = act && t < len;
            if (__ballot(go)) {
                const uint32_t v = B.get16(in0 + t);
                int n = !go ? 0 : len - t < 16 ? len - t : 16;
                // homopolymer starts among the sixteen: base i differs from the one before it (the first one from prev_c)
                const uint32_t x = v ^ ((v << 2) | (prev_c & 3u));
                const uint32_t starts = even_bits16(x | (x >> 1)) | (prev_c > 3u ? 1u : 0u);
                const uint32_t below = n < 16 ? (1u << n) - 1u : 0xFFFFu;
                const uint32_t sb = starts & below, rem = G.next - k1;                     // starts that may be passed before the one that scores
                const bool ev = go && rem < (uint32_t)__popc(sb);
                if (ev) n = select_bit16(sb, rem);                                        // ... which stops the step
                const uint32_t taken = n < 16 ? (1u << n) - 1u : 0xFFFFu;
                k1 += (uint32_t)__popc(sb & taken);
                // the flow pointer over the n bases: pairs, then the odd one
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const uint32_t e = T.pair[(flow_i << 4) | ((v >> (4 * p)) & 15u)];
                    if (2 * p + 2 <= n) flow_i = e & 63u;
                }
                if (n & 1) flow_i = T.next1[(flow_i << 2) | ((v >> (2 * (n - 1))) & 3u)];
                if (starts & taken) marked = false;
                o1.push_many(n < 16 ? v & ((1u << (2 * n)) - 1u) : v, n);
                if (n) prev_c = (v >> (2 * (n - 1))) & 3u;
                t += n;
                if (ev) parked = true;                                                   // standing on a homopolymer start whose first draw scored: the event happens
            }
            // ---- an event round: the parked lanes draw what their events draw beyond the first uniform (one Philox block), act on it, and draw their next gap
            const uint64_t pm = __ballot(parked);
            if (pm && (__popcll(pm) >= FLOW_EVENT_BATCH || __ballot(!done && !parked) == 0)) {
                if (parked) E.resolve(key, rg.dom, rg.ii, rg.att, (uint32_t)o1.n, thr);
                if (parked) {
                    const uint32_t c = B.get1(in0 + t);
                    flow_i = T.next1[(flow_i << 2) | c]; marked = false;
                    int n_err; bool ins = false;
                    const bool slow = (E.oc & 0x200u) != 0;
                    if (slow) { rg.open((uint32_t)o1.n); n_err = rg.more_errors(thr); if (n_err >= (1 << 14)) failed = true; else ins = rg.next() < 0x80000000u; }
                    else { n_err = (int)(E.oc & 0xffu); ins = (E.oc & 0x100u) != 0; }
                    if (failed) {}
                    else if (ins) {                                     // insert n_err copies in front of the homopolymer (whose own bases follow unexamined: prev_c == c)
                        if (o1.n + n_err > in0 + t) failed = true;      // the output would run into the input: the read has outgrown the buffer
                        else { for (int q = 0; q < n_err; ++q) o1.push(c); total += n_err; prev_c = c; }
                    } else {                                            // delete: bounded by the homopolymer length
                        int hp_l = 0; uint32_t next_c = c;
                        while (t + hp_l < len && hp_l <= n_err) { next_c = B.get1(in0 + t + hp_l); if (next_c != c) break; ++hp_l; }
                        if (n_err > hp_l) n_err = hp_l;
                        t += n_err; marked = true; total += n_err;
                        if (n_err == hp_l && (o1.n == 0 || prev_c == next_c)) {   // dot-fill (dwgsim.c:342-358)
                            if (next_c == c) failed = true;             // the whole read was one deleted homopolymer (the reference asserts)
                            else {
                                const int jj = T.dist[(flow_i << 2) | next_c];
                                const uint32_t dw = slow ? rg.next() : E.oc_dot;
                                const int kk = (int)(((uint64_t)dw * (uint64_t)jj) >> 32);   // (int)(drand48() * j)
                                int f = (int)flow_i + kk; if (f >= F) f -= F;
                                o1.push(T.flow[f]);
                            }
                        } else if (t < len) { o1.push(B.get1(in0 + t)); ++t; }   // the base now at this position is not examined
                        prev_c = c;
                    }
                    ++k1;                                               // the start's own first draw was number k1
                    if (failed) done = true;
                }
                G.draw(parked && !failed, key, rg.dom, rg.ii, rg.att, k1, thr, T.lg, gap_R, gap_s);
                parked = false;
            }
            if (__ballot(!done) == 0) break;
        }
    }
    o1.flush();
    const int n1 = o1.n;
    const int marked_flow = marked ? (int)flow_i : -1;      // the one flow of the (persistent) mask that pass 2 finds set

    // ---- the output of pass 1 moves up to the buffer's top (positions capb - n1 ..), from the top word down: one read and one write per word ----
    if (!failed && n1 > 0 && n1 < capb) {
        const int D = capb - n1, dw = D >> 4; const uint32_t sh = ((uint32_t)D & 15u) * 2u;
        uint32_t hi = buf[(capw - 1 - dw) * stride];
        for (int j = capw - 1; j >= dw; --j) {
            const uint32_t lo = j - dw - 1 >= 0 ? buf[(j - dw - 1) * stride] : 0u;
            buf[j * stride] = sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;
            hi = lo;
        }
    }

    // ---- pass 2 (dwgsim.c:367-406): insertions in empty flows; inserted bases are examined again later, the examined base
    // itself stays behind them: a stack of (base, count) runs on top of the pass-1 output reproduces the in-place order.
    // g counts the empty flows examined so far: flow g's first draw is first draw number g of the pass.  A base whose empty flows end before the next
    // scoring flow is quiet: lanes with an empty stack move up to sixteen quiet bases per step; a base with the scoring flow in front of it, and every
    // base examined while runs are pending, parks and goes through the flow-by-flow code of the event rounds. ----
    const uint32_t dom2 = rg.dom + D_FLOW_PASS2;
    const int in2 = capb - n1;
    BitAppender<2> o2; o2.init(buf, stride);
    auto stk_get = [&](int k) -> uint32_t { return (stk[(k >> 1) * stride] >> ((k & 1) * 16)) & 0xffffu; };
    auto stk_set = [&](int k, uint32_t v) { const uint32_t sh = (uint32_t)(k & 1) * 16; uint32_t w = stk[(k >> 1) * stride]; stk[(k >> 1) * stride] = (w & ~(0xffffu << sh)) | (v << sh); };
    int t2 = 0, sp = 0;
    auto settle = [&](uint32_t x) {                        // the position's final base: the examined base, or the first base of the top run
        if (sp == 0) { o2.push(x); ++t2; }
        else if (o2.n >= in2 + t2) failed = true;          // (the output would run into the input: outgrown)
        else {
            const uint32_t top = stk_get(sp - 1);
            o2.push(top >> 14);
            if ((top & 0x3fffu) <= 1) --sp; else stk_set(sp - 1, top - 1);
        }
    };
    rg.dom = dom2;
    G.m = 0; G.next = FLOW_NEVER;
    G.draw(__ballot(!failed) != 0, key, dom2, rg.ii, rg.att, 0u, thr, T.lg, gap_R, gap_s);
    {
        bool done = failed, parked = false; uint32_t g = 0, x = 0;
        for (;;) {
            const bool act = !done && !parked;
            if (act && sp == 0 && t2 >= n1) done = true;
            const bool go = act && !done;
            if (__ballot(go)) {
                const uint32_t v = B.get16(in2 + t2);
                const int n = !go ? 0 : n1 - t2 < 16 ? n1 - t2 : 16;
                uint32_t rem = G.next - g;                                   // quiet flows in front of the scoring one: they may be passed
                int m = 0; bool ok = true;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const uint32_t e = T.pair[(flow_i << 4) | ((v >> (4 * p)) & 15u)], kt = e >> 8;
                    ok = ok && 2 * p + 2 <= n && kt <= rem;
                    if (ok) { rem -= kt; flow_i = e & 63u; m += 2; }
                }
                if (m < n) {                                                 // the pair that did not fit: its first base alone may
                    const uint32_t i1 = (flow_i << 2) | ((v >> (2 * m)) & 3u), k = T.dist[i1];
                    if (k <= rem) { rem -= k; flow_i = T.next1[i1]; ++m; }
                }
                g = G.next - rem;
                o2.push_many(m < 16 ? v & ((1u << (2 * m)) - 1u) : v, m);
                t2 += m;
                if (m < n) { x = (v >> (2 * m)) & 3u; parked = true; }        // the scoring flow lies in front of this base
            }
            // ---- an event round: flow by flow up to the parked base's own flow -- quiet flows are skipped together, the scoring one inserts (dwgsim.c:373-383) --
            // and on through the runs it leaves pending, until the lane can step again
            const uint64_t pm = __ballot(parked);
            if (pm && (__popcll(pm) >= FLOW_EVENT_BATCH || __ballot(!done && !parked) == 0)) {
                if (parked) {
                    for (;;) {
                        uint32_t left = T.dist[(flow_i << 2) | x];
                        const uint32_t q = G.next - g, skip = q < left ? q : left;
                        { uint32_t f = flow_i + skip; if (f >= (uint32_t)F) f -= (uint32_t)F; flow_i = f; }
                        g += skip; left -= skip;
                        if (left == 0) {
                            settle(x);
                            if (failed) { done = true; break; }
                            if (sp == 0) break;
                            x = stk_get(sp - 1) >> 14;                  // the next base to examine is the first of the top run
                            continue;
                        }
                        E.resolve(key, dom2, rg.ii, rg.att, g, thr);    // the scoring flow: what the event draws beyond its first uniform
                        int n_err;
                        if (E.oc & 0x200u) { rg.open(g); n_err = rg.more_errors(thr); } else n_err = (int)(E.oc & 0xffu);
                        if ((int)flow_i != marked_flow) {
                            if (sp >= stack_runs || n_err >= (1 << 14)) { failed = true; done = true; break; }
                            stk_set(sp, ((uint32_t)T.flow[flow_i] << 14) | (uint32_t)n_err); ++sp; total += n_err;
                        }
                        flow_i = flow_i + 1u == (uint32_t)F ? 0u : flow_i + 1u;
                        ++g;
                        G.draw(true, key, dom2, rg.ii, rg.att, g, thr, T.lg, gap_R, gap_s);
                    }
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
    DW_DEV void put_lead8(uint32_t lead, uint64_t v, uint32_t cnt)          // one character, then cnt (1 .. 8) more
    {
        if (cnt < 8) putn((uint64_t)lead | (v << 8), cnt + 1);
        else { putn((uint64_t)lead | (v << 8), 8); putn(v >> 56, 1); }
    }
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
// DUAL (-o 0, both output families): the BFAST record is the BWA record without the two characters "/1" (or "/2") in front of the name line's
// end (dwgsim.c:919-981), so the second stream is written FROM THE SAME IMAGE: every burst that leaves for the BWA file leaves for the BFAST file too
// (the registers are loaded once), at the BFAST record's own byte phase -- unaligned 16-byte stores, two per burst.  Rounds 1-4 assembled the BFAST
// record a second time in registers and wrote it in 16-byte pieces (2.3-2.4 x the text in HBM writes, two records' worth of assembly per lane:
// profiles/r04_variants_pmc.txt).  dstb + p is where FIFO position p goes in the BFAST stream; bdone = the first position the BFAST stream has
// not written yet (it differs from skip only between the suffix and the next drain: the suffix's own two characters are jumped over).
// bytes [from, upto) of a unit (the ragged first / last unit of a record) from the FIFO at f to dst: rising sizes until the position is aligned (or the
// next piece would pass upto), then falling sizes.  (DW_DEV_NOINLINE: dw_intrin.hpp)
template <uint32_t BURST>
DW_DEV_NOINLINE void fifo_store_range(const uint8_t *f, uint8_t *dst, uint32_t from, uint32_t upto)
{
    auto ld8 = [&](uint32_t b) { return *reinterpret_cast<const uint64_t *>(f + b); };
    auto st16 = [&](uint32_t b) { *reinterpret_cast<uint4 *>(dst + b) = make_uint4((uint32_t)ld8(b), (uint32_t)(ld8(b) >> 32), (uint32_t)ld8(b + 8), (uint32_t)(ld8(b + 8) >> 32)); };
    uint32_t b = from;
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
// the same for a destination of any alignment (the second stream of FifoWriter DUAL): falling sizes by the bits of the count
template <uint32_t BURST>
DW_DEV_NOINLINE void fifo_store_range_unaligned(const uint8_t *f, uint8_t *dstb, uint32_t from, uint32_t upto)
{
    if (from >= upto) return;
    auto ld8u = [&](uint32_t b) { return reinterpret_cast<const Unal8 *>(f + b)->v; };
    uint32_t b = from; const uint32_t n = upto - from;
    if (BURST == 64u && (n & 64u)) { for (int q = 0; q < 4; ++q) { Unal16 v; v.a = ld8u(b); v.b = ld8u(b + 8); *reinterpret_cast<Unal16 *>(dstb + b) = v; b += 16; } }
    if (n & 32u) { for (int q = 0; q < 2; ++q) { Unal16 v; v.a = ld8u(b); v.b = ld8u(b + 8); *reinterpret_cast<Unal16 *>(dstb + b) = v; b += 16; } }
    if (n & 16u) { Unal16 v; v.a = ld8u(b); v.b = ld8u(b + 8); *reinterpret_cast<Unal16 *>(dstb + b) = v; b += 16; }
    if (n & 8u) { Unal8 v; v.v = ld8u(b); *reinterpret_cast<Unal8 *>(dstb + b) = v; b += 8; }
    if (n & 4u) { Unal4 v = *reinterpret_cast<const Unal4 *>(f + b); *reinterpret_cast<Unal4 *>(dstb + b) = v; b += 4; }
    if (n & 2u) { Unal2 v = *reinterpret_cast<const Unal2 *>(f + b); *reinterpret_cast<Unal2 *>(dstb + b) = v; b += 2; }
    if (n & 1u) { dstb[b] = f[b]; }
}
template <uint32_t BURST = 32u, bool DUAL = false>
struct FifoWriter {
    uint8_t *f, *dst; uint32_t wp, skip;
    uint8_t *dstb; uint32_t bdone;
    DW_DEV void init(uint8_t *fifo, uint8_t *rec) { const uint32_t h = (uint32_t)((uintptr_t)rec & (BURST - 1u)); f = fifo; dst = rec - h; wp = skip = h; }
    DW_DEV void init(uint8_t *fifo, uint8_t *rec, uint8_t *rec_b) { init(fifo, rec); dstb = rec_b - skip; bdone = skip; }
    DW_DEV uint64_t ld8(uint32_t b) const { return *reinterpret_cast<const uint64_t *>(f + b); }
    DW_DEV uint64_t ld8u(uint32_t b) const { return reinterpret_cast<const Unal8 *>(f + b)->v; }
    DW_DEV void st16(uint32_t b) const { *reinterpret_cast<uint4 *>(dst + b) = make_uint4((uint32_t)ld8(b), (uint32_t)(ld8(b) >> 32), (uint32_t)ld8(b + 8), (uint32_t)(ld8(b + 8) >> 32)); }
    DW_DEV void st16_both(uint32_t b) const          // sixteen bytes of a whole unit: aligned in the first stream, wherever they fall in the second
    {
        const uint64_t lo = ld8(b), hi = ld8(b + 8);
        *reinterpret_cast<uint4 *>(dst + b) = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        Unal16 v; v.a = lo; v.b = hi; *reinterpret_cast<Unal16 *>(dstb + b) = v;
    }
    DW_DEV void store_range(uint32_t from, uint32_t upto) const { fifo_store_range<BURST>(f, dst, from, upto); }      // (ragged first / last unit of a record)
    DW_DEV void store_range_b(uint32_t from, uint32_t upto) const { fifo_store_range_unaligned<BURST>(f, dstb, from, upto); }
    DW_DEV void drain()                          // wp >= BURST: one unit leaves
    {
        if (!(probe::off(2))) {
            if (!DUAL) {
                if (skip == 0) { st16(0); st16(16); if (BURST == 64u) { st16(32); st16(48); } }
                else store_range(skip, BURST);
            } else {
                if ((skip | bdone) == 0) { st16_both(0); st16_both(16); if (BURST == 64u) { st16_both(32); st16_both(48); } }
                else {          // (store_range covers fewer than BURST bytes: a whole unit of the first stream goes as it does above)
                    if (skip == 0) { st16(0); st16(16); if (BURST == 64u) { st16(32); st16(48); } } else store_range(skip, BURST);
                    store_range_b(bdone, BURST);
                }
            }
        }
        skip = 0;
        if (DUAL) { bdone = bdone > BURST ? bdone - BURST : 0u; dstb += BURST; }
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
    // two characters written `off` (<= 6) places past the write position without moving it, then advance(n): how the quality line lays down the (up to
    // four) pairs of characters a Philox block delivers -- a rejected try's pair is simply overwritten by the next one's
    DW_DEV void poke2(uint32_t off, uint32_t v) { if (probe::off(1)) return; Unal2 x; x.v = (uint16_t)v; *reinterpret_cast<Unal2 *>(f + wp + off) = x; }
    DW_DEV void advance(uint32_t n) { if (probe::off(1)) return; wp += n; if (wp >= BURST) drain(); }
    // one character, then cnt (1 .. 8) more: two overlapping stores, one step (the nine bytes end at position wp + 8 <= BURST + 7: inside the FIFO)
    DW_DEV void put_lead8(uint32_t lead, uint64_t v, uint32_t cnt)
    {
        if (probe::off(1)) return;
        f[wp] = (uint8_t)lead;
        Unal8 x; x.v = v; *reinterpret_cast<Unal8 *>(f + wp + 1) = x;
        wp += cnt + 1; if (wp >= BURST) drain();
    }
    // DUAL: the next `only_a` characters exist in the first stream only (they are the head of the cnt characters of v).  What the second stream has
    // not written yet leaves now; from here on its bytes stand only_a places further down
    DW_DEV void putn_first_only(uint64_t v, uint32_t cnt, uint32_t only_a)
    {
        if (probe::off(1)) return;
        if (!(probe::off(2))) store_range_b(bdone, wp);
        bdone = wp + only_a; dstb -= only_a;
        putn(v, cnt);
    }
    DW_DEV void put4(uint32_t w) { if (probe::off(1)) return; Unal4 x; x.v = w; *reinterpret_cast<Unal4 *>(f + wp) = x; wp += 4; if (wp >= BURST) drain(); }
    DW_DEV void put16(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { putn((uint64_t)a | ((uint64_t)b << 32), 8); putn((uint64_t)c | ((uint64_t)d << 32), 8); }
    DW_DEV void flush()
    {
        if (!(probe::off(2))) { if (wp > skip) store_range(skip, wp); if (DUAL) store_range_b(bdone, wp); }
        dst += wp; if (DUAL) dstb += wp;
        wp = skip = 0; bdone = 0;
    }
};
// OUT bit 0: the bwa stream of this read end, bit 1: the interleaved bfast stream.  The (first) output goes through the FIFO writer (WR = 1, 2)
// or the register writer (WR = 0: the host found that the FIFO's LDS would cost a resident block per CU); with both outputs (-o 0) the
// bfast stream leaves from the same FIFO image (FifoWriter<.., DUAL>), or, beside the register writer, through a register writer of its own.
template <int WR, bool DUAL> struct PrimaryWriter { typedef FifoWriter<32u, DUAL> type; };      // WR = 1
template <bool DUAL> struct PrimaryWriter<0, DUAL> { typedef Writer type; };
template <bool DUAL> struct PrimaryWriter<2, DUAL> { typedef FifoWriter<64u, DUAL> type; };
template <int OUT, int WR = 1>
struct Out2 {
    static constexpr bool DUAL = OUT == 3 && WR != 0;       // both streams from one FIFO image
    static constexpr bool TWO = OUT == 3 && WR == 0;        // both streams, a register writer each
    typename PrimaryWriter<WR, DUAL>::type a; Writer b;
    DW_DEV void init(uint8_t *fifo, uint8_t *bwa, uint8_t *bfast)
    {
        if constexpr (DUAL) a.init(fifo, bwa, bfast);
        else { a.init(fifo, (OUT & 1) ? bwa : bfast); if (TWO) b.init(bfast); }
    }
    DW_DEV void put(uint32_t c) { a.put(c); if (TWO) b.put(c); }
    DW_DEV void put4(uint32_t w) { a.put4(w); if (TWO) b.put4(w); }
    DW_DEV void putn(uint64_t v, uint32_t cnt) { a.putn(v, cnt); if (TWO) b.putn(v, cnt); }
    DW_DEV void put_lead8(uint32_t lead, uint64_t v, uint32_t cnt) { a.put_lead8(lead, v, cnt); if (TWO) b.put_lead8(lead, v, cnt); }
    DW_DEV void put16(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { a.put16(x, y, z, w); if (TWO) b.put16(x, y, z, w); }
    // the end of the name line differs between the two families: "/1\n" (bwa) and "\n" (bfast): v_bf is the tail of v_bwa
    DW_DEV void put_suffix(uint64_t v_bwa, uint32_t n_bwa, uint64_t v_bf, uint32_t n_bf)
    {
        if constexpr (DUAL) a.putn_first_only(v_bwa, n_bwa, n_bwa - n_bf);
        else {
            if (OUT & 1) a.putn(v_bwa, n_bwa); else a.putn(v_bf, n_bf);
            if (TWO) b.putn(v_bf, n_bf);
        }
    }
    DW_DEV void rebase() { if (WR == 0) a.flush(); if (TWO) b.flush(); }      // a new section of the register writer(s); the FIFO needs none
    DW_DEV void flush() { a.flush(); if (TWO) b.flush(); }
};
DW_DEV uint32_t ndigits10(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}
DW_DEV uint32_t ndigits16(uint64_t v) { return v ? (uint32_t)(67 - __clzll((long long)v)) >> 2 : 1u; }
// ---- numbers as text, without a loop per digit (rounds 1-4 divided by ten once per digit: the name line was 0.3 of 5.9 ms) ----
// the four decimal digits of x < 10 000 as ASCII, most significant digit in the lowest byte (stream order), leading zeros kept: x = 100 a + b by one
// multiplication, then both two-digit numbers at once in the halves of a word (n / 10 = n * 103 >> 10 for n < 100)
DW_DEV uint32_t dec4(uint32_t x)
{
    const uint32_t a = (x * 5243u) >> 19;                        // x / 100 for x < 43 699
    const uint32_t p = a | ((x - a * 100u) << 16);
    const uint32_t t = ((p * 103u) >> 10) & 0x000F000Fu;         // the tens of both (99 * 103 < 2^16: the halves do not mix)
    return (t | ((p - t * 10u) << 8)) + 0x30303030u;
}
// eight digits of v < 10^8, leading zeros kept
DW_DEV uint64_t dec8(uint32_t v)
{
    const uint32_t hi = (uint32_t)(((uint64_t)v * 0xD1B71759ull) >> 45);      // v / 10 000 for every 32-bit v
    return (uint64_t)dec4(hi) | ((uint64_t)dec4(v - hi * 10000u) << 32);
}
// ... with the leading zeros taken off: the digits of v < 10^8 in the low bytes of the result, *nd of them (1 .. 8)
DW_DEV uint64_t dec8_stripped(uint32_t v, uint32_t *nd)
{
    const uint64_t w = dec8(v);
    const uint32_t z = (uint32_t)__builtin_ctzll((w - 0x3030303030303030ull) | (1ull << 56)) >> 3;      // leading zero digits (at most seven)
    *nd = 8u - z;
    return w >> (8u * z);
}
// the hexadecimal digits of the low 32 bits of v, most significant first, leading zeros kept: nibbles spread to bytes, 'a' - '0' - 10 added where a digit passes 9
DW_DEV uint64_t hex8(uint32_t v)
{
    auto four = [](uint32_t n) -> uint32_t {                     // digits 3 .. 0 of n (16 bits), digit 3 in the lowest byte
        uint32_t x = ((n & 0xF000u) >> 12) | ((n & 0x0F00u) << 0) | ((n & 0x00F0u) << 12) | ((n & 0x000Fu) << 24);
        const uint32_t over = ((x + 0x06060606u) >> 4) & 0x01010101u;      // 1 where the digit is 10 .. 15
        return x + 0x30303030u + over * 39u;
    };
    return (uint64_t)four(v >> 16) | ((uint64_t)four(v & 0xFFFFu) << 32);
}
// lead (one separator character) + the decimal digits of v
template <class O>
DW_DEV void put_dec(O &o, uint32_t v, uint32_t lead)
{
    if (v >= 100000000u) {                     // nine or ten digits: the first one or two with the separator, then eight
        const uint32_t hi = v / 100000000u;    // 1 .. 42
        const uint32_t t = (hi * 103u) >> 10;
        o.putn(t ? (uint64_t)lead | ((uint64_t)('0' + t) << 8) | ((uint64_t)('0' + hi - 10u * t) << 16) : (uint64_t)lead | ((uint64_t)('0' + hi) << 8), t ? 3u : 2u);
        o.putn(dec8(v - hi * 100000000u), 8);
        return;
    }
    uint32_t nd; const uint64_t w = dec8_stripped(v, &nd);
    o.put_lead8(lead, w, nd);
}
// lead + the hexadecimal digits of v
template <class O>
DW_DEV void put_hex(O &o, uint64_t v, uint32_t lead)
{
    const uint32_t nd = ndigits16(v);
    if (nd > 8) {
        o.put_lead8(lead, hex8((uint32_t)(v >> 32)) >> (8u * (16u - nd)), nd - 8u);
        o.putn(hex8((uint32_t)v), 8);
    } else o.put_lead8(lead, hex8((uint32_t)v) >> (8u * (8u - nd)), nd);
}
// "_e:u:i": the three counts of a read end behind their separators; one piece when all three are single digits (almost always)
template <class O>
DW_DEV void put_counts(O &o, uint32_t e, uint32_t u, uint32_t i)
{
    if (e < 10u && u < 10u && i < 10u)
        o.putn((uint64_t)'_' | ((uint64_t)('0' + e) << 8) | ((uint64_t)':' << 16) | ((uint64_t)('0' + u) << 24) | ((uint64_t)':' << 32) | ((uint64_t)('0' + i) << 40), 6);
    else { put_dec(o, e, '_'); put_dec(o, u, ':'); put_dec(o, i, ':'); }
}
DW_DEV uint32_t counts_len(uint32_t e, uint32_t u, uint32_t i)
{
    return (e < 10u && u < 10u && i < 10u) ? 6u : 3u + ndigits10(e) + ndigits10(u) + ndigits10(i);
}
DW_DEV uint32_t base_chars4(uint32_t nibbles) { return lut8(0x4E4E4E4Eu, 0x54474341u, spread4(nibbles)); }        // four codes (<= 7) -> "ACGTNNNN"[code]
DW_DEV uint32_t colour_digits4(uint32_t nibbles) { return lut8(0x34343434u, 0x33323130u, spread4(nibbles)); }     // four colours -> "01234444"[colour]

} // namespace dw
