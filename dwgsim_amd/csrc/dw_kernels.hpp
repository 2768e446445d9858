// dw_kernels.hpp -- argument blocks shared by the HIP kernels (dw_walk.hip, dw_simulate.hip) and the
// C-ABI host code (dw_host.cpp).
#pragma once
#include <stdint.h>

namespace dw {

constexpr int PAIRS_PER_BLOCK = 128;     // k_calibrate: reads per block
constexpr int PLACE_PAIRS = 256;         // k_place: pairs per block (one lane per pair)
constexpr int PLACE_LISTS = 64;          // k_place hands the pairs it cannot decide to k_place_rest through this many lists (one counter each: no hot word)
#ifndef DW_SIM_THREADS
#define DW_SIM_THREADS 256
#endif
constexpr int SIM_THREADS = DW_SIM_THREADS;   // k_simulate: threads per block (one lane per read end)
constexpr int FLOW_STACK_WORDS = 4;           // Ion Torrent pass 2: LDS words per lane for the (base, count) runs that can be pending in front of the examined base (two per word), times the
constexpr uint32_t FLOW_NEVER = 0xFFFFFFFFu;     // Ion Torrent: "no first draw of this pass scores any more" (dw_read.hpp FlowGap)
constexpr int FLOW_LG_ENTRIES = 257;              // ... entries of the log2 table the gaps between scoring first draws are computed with (behind the 64 bytes of the flow order)
constexpr int FLOW_CAP_MAX = 1 << 20;           // Ion Torrent: bases a read may grow to in the flow model (capacity re-runs double the buffers up to this)
constexpr int FLOW_STACK_WORDS_MAX = 32;      // ... capacity multiplier of the job (a read that outgrows the stack is run again like one that outgrows its buffer), up to this many
#ifndef DW_ION_THREADS_SMALL
#define DW_ION_THREADS_SMALL 128
#endif
constexpr int ION_THREADS_SMALL = DW_ION_THREADS_SMALL;      // Ion Torrent with its read buffers in LDS: the smaller block form (fill_sim_args "ion_lds" = 2)
constexpr int SIM_THREADS_LONG = 64;          // ... one-wave blocks for reads of ~650 bases and more: their bases are staged in scratch slots, not LDS
constexpr int SIM_FIFO_BYTES = 40;            // per lane: the text FIFO of the record writer (one 32-byte burst + the overshoot of an 8-byte put)
constexpr int SIM_FIFO_BYTES_WIDE = 72;       // ... with 64-byte bursts (second half of the two-kernel form: no staged bases compete for LDS)
// dynamic LDS of a k_simulate block: [words_per_lane][lanes] staged bases | the two base-quality tables | [lanes] text FIFOs (16-byte aligned)
inline size_t sim_lds_bytes(size_t words_per_lane, size_t lanes, size_t qb_words, bool fifo, size_t fifo_bytes = SIM_FIFO_BYTES) { return ((words_per_lane * lanes + 2 * qb_words + 3) & ~(size_t)3) * 4 + (fifo ? lanes * fifo_bytes : 0); }
// blocks of SIM_THREADS lanes a CU holds at this much dynamic LDS (160 KB per CU in granules of 1280 bytes, ~0.6 KB static per block), at most `cap` (the register limit)
inline int sim_blocks_per_cu(size_t dyn_lds, int cap) { const size_t per = (dyn_lds + 1700 + 1279) / 1280 * 1280;      /* (+ the kernel's static LDS: scan scratch, name lines, the log2 table of the gap draws) */ const int b = (int)(163840 / per); return b < cap ? b : cap; }
constexpr size_t SIM_LDS_BUDGET = 150 * 1024; // dynamic LDS a block may ask for (160 KB per CU minus the static part)
constexpr int SCAN_POS_PER_THREAD = 16;  // k_site_scan / k_collect: 16 positions (one 16-B load) per thread
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_POS_PER_BLOCK = SCAN_POS_PER_THREAD * SCAN_THREADS;   // 4096
constexpr int CELL_PAD = 64;             // cells are allocated with 64 pad bytes after the last contig: vector loads run past a contig end, and k_dirty_chunks handles whole 64-cell chunks (a group total is a multiple of 16 only)
constexpr int GROUP_ALIGN = SCAN_POS_PER_BLOCK;   // contigs that are resident together share ONE coordinate space ("group"): contig k starts at a multiple of this many cells,
                                         // so that a block of the position-parallel walk kernels, a 32-cell chunk of the read view and a 64-cell summary all lie inside one contig
constexpr int MAX_ATTEMPTS = 10000;      // dwgsim.c:837

// mutation type bits of a cell (low byte of the reference's mut_t, mut.h:25-30)
constexpr uint8_t T_NONE = 0x00, T_INS = 0x10, T_SUB = 0x20, T_DEL = 0x30, TMASK = 0x30, BTMASK = 0x3f;

// a candidate mutation site resolved speculatively (k_events) and then marked live/dead (k_resolve)
struct Event {
    int32_t  pos;       // group coordinate
    uint8_t  type;      // 0 dead, 1 substitution, 2 deletion, 3 insertion, 4 patched from a mutation-input file
    uint8_t  hap;       // haplotype mask 1|2
    uint8_t  base;      // substitution: new base; else reference code at pos
    uint8_t  live;
    uint32_t len;       // deletion run length / insertion length
    uint32_t seg;       // which contig of the group
};

// The contigs of a group: contig k occupies group coordinates [start[k], start[k] + len[k]); start[] is ascending, multiples of GROUP_ALIGN,
// start[n] = the padded total.  Cells between two contigs are unmutated N.
struct SegTab { const int32_t *start; const int32_t *len; const uint32_t *cindex; int32_t n; };

// one haplotype of one contig, resident in HBM
struct HapDev {
    uint8_t *cells;         // [l + CELL_PAD]  bits 0-3 base code (0-3 ACGT, 4 N, 5 '-'), bits 4-5 type
    const uint8_t *view;    // the read view of the cells, one NIBBLE per cell (cell i in byte i >> 1, even cells low): 0-3 unmutated A C G T, 4-7 substituted
                            // to A C G T, 8 unmutated N, >= 9 "look at the byte cell" (INSERT / DELETE cells, '-').  Built by k_make_view after the walk;
                            // base extraction reads 32 cells per 16-byte load from it and touches the byte cells only for the escapes
    int32_t *ins_pos;       // sorted positions of INSERT cells
    uint32_t *ins_len;
    uint32_t *ins_off;      // offset of the inserted bases P[0..n) (printed order) in ins_bases
    uint8_t *ins_bases;
    uint32_t n_ins;
    int32_t pos_off;        // added to a position before it is looked up in ins_pos (which holds group coordinates): 0 in the walk kernels, the contig's start
                            // in the read kernels, whose cells / view pointers are those of the contig
};

struct ContigDev {             // a group of contigs in its coordinate space (a single contig: one segment starting at 0)
    HapDev hap[2];
    const uint8_t *ref;     // [total + CELL_PAD] reference base codes
    int64_t l;              // padded total of the group
    SegTab seg;
    const uint32_t *tot4;   // walk kernels only (else null): device copy of {n_ins[0], n_ins_bases[0], n_ins[1], n_ins_bases[1]} still being produced
    uint32_t cap_bases[2];  // ... and the capacity of the inserted-base pools
};

// An element count that is either known on the host or still being produced on the device: kernels are launched for `host`
// elements (an exact count, or the capacity the buffers were sized for) and work on min(*dev, host).
struct Count { const uint64_t *dev; uint32_t host; };

struct WalkParams {
    double mut_rate, indel_frac, indel_extend;
    uint64_t mut_thr;              // ceil(mut_rate * 2^32) (2^32 for a rate of 1): u < mut_rate  <=>  the 32-bit draw < mut_thr
    uint64_t gap_r; int32_t gap_s; // the gaps between candidate sites (dw_common.hpp geom_gap): flow_gap_params(mut_thr) ...
    const uint32_t *lg;            // ... and the log2 table they interpolate in (device memory, FLOW_LG_ENTRIES words)
    int32_t indel_min, is_hap;
    uint32_t seed;
};

struct SimParams {
    double std_dev, mut_freq, rand_read, quality_std;
    int32_t dist, is_inner, len[2], max_n, strandedness, read_one_strand, amplicons, fixed_quality, data_type;
    int32_t has_bwa, has_bfast;
    uint32_t seed;
    // lazy quality normals (dw_simulate.hip quality_try_lazy, host: lazy_quality_params): k = sqrt(2 ln 2) * quality_std as a float,
    // eps = the proven error bound of the fp32 estimate, lmin / near1 = the treatment of polar radii next to 1
    float q_k, q_eps, q_lmin; int32_t q_near1;
};

// -B: per-base calibration of the Ion Torrent flow error (dwgsim_opt.c:415-457): n_reads random reads of `len` bases of read end `end`
// ---- Ion Torrent: the gaps between scoring first draws of the flow model (dw_read.hpp FlowGap; dw_common.hpp D_FLOW0), host side.  Integer arithmetic
// only, so that the kernels, this file and the CPU restatement the tests check them against agree bit for bit: no libm. ----
// floor(log2(y) * 2^fb) for y >= 1, fb <= 56: repeated squaring in Q1.63
inline uint64_t flow_ilog2_fixed(uint64_t y, int fb)
{
    const int p = 63 - __builtin_clzll(y);
    uint64_t m = y << (63 - p), frac = 0;
    for (int k = 0; k < fb; ++k) {
        const unsigned __int128 sq = (unsigned __int128)m * m;       // Q2.126
        if ((uint64_t)(sq >> 127)) { m = (uint64_t)(sq >> 64); frac = (frac << 1) | 1u; }
        else { m = (uint64_t)(sq >> 63); frac <<= 1; }
    }
    return ((uint64_t)p << fb) | frac;
}
inline void flow_log2_table(uint32_t *lg)      // FLOW_LG_ENTRIES words: floor(2^32 log2(1 + i / 256)), the last one 2^32 - 1
{
    for (int i = 0; i < 256; ++i) lg[i] = (uint32_t)flow_ilog2_fixed(256u + (uint64_t)i, 32);
    lg[256] = 0xFFFFFFFFu;
}
// G = floor(-log2(U) / -log2(1 - thr / 2^32)) as mulhi64(-log2(U) in Q8.56, R) >> s: R = 2^127 / (normalised -log2(1 - e') in Q8.56), s = 63 - its shift
inline void flow_gap_params(uint64_t thr, uint64_t *R, int32_t *s)
{
    *R = 0; *s = 0;
    if (thr == 0 || thr >= 0x100000000ull) return;      // never / always: no gap is computed
    const uint64_t Lq = (32ull << 56) - flow_ilog2_fixed(0x100000000ull - thr, 56);
    const int sh = __builtin_clzll(Lq);
    const unsigned __int128 q = ((unsigned __int128)1 << 127) / (Lq << sh);
    *R = (uint64_t)(q >> 64) ? ~0ull : (uint64_t)q; *s = 63 - sh;
}

struct CalibArgs {
    uint32_t seed; int32_t end, len; uint64_t n_reads, first_read, chunk_reads;      // reads [first_read, min(first_read + chunk_reads, n_reads)) in this launch (the scratch is sized for a chunk)
    uint64_t thr;                   // ceil(e * 2^32) of the uncalibrated -e
    uint64_t gap_r; int32_t gap_s;  // ... and what the gaps between scoring first draws are computed with (flow_gap_params)
    const uint8_t *flow; int32_t flow_len, lds_words, stack_words;       // flow: 64 bytes of flow order + the log2 table (flow_log2_table)       // lds_words: words per lane of the read buffer (16 bases each)
    uint32_t *scratch;              // per block [lds_words][PAIRS_PER_BLOCK] words
    uint64_t *counters;             // [8] += errors, [9] += read lengths after errors, [2] |= 2 on a buffer overflow
};

constexpr int SUMM_CELLS = 64;          // cells per haplotype-summary word (k_summarize / k_place)
constexpr int SUMM2_CELLS = 1024;       // ... of the coarse level: bits 0-14 INSERT / DELETE cells, bit 15 a base code >= 4 (16 fine words each)

// Ion Torrent read buffers in a scratch slot (DT = 2): words per lane, forced odd so that the blocks' areas do not all start on the same HBM
// channels (a 96 KB stride cost 14 % against 89 KB)
inline constexpr int flow_words_per_lane(int lds_words) { return lds_words | 1; }

// One read-index range of one contig inside a k_simulate / k_place launch.  A launch covers n_seg of them, in file order; a block never spans two.
struct SimSeg {
    uint32_t first_block;          // logical block of the range's first pair (exclusive prefix of ceil(n_pairs / pairs per block))
    uint32_t contig_index;         // RNG key
    int32_t  start, l;             // the contig inside its group
    uint64_t first_ii, n_pairs;    // the range
    uint64_t pair_off;             // pairs of the launch in front of this range (index into meta)
    int64_t  l_place;              // the `l` that sizes fragment placement: contig length, or the contig's region length with -x (dwgsim.c:552)
    int32_t  reg_off, n_reg;       // -x: this contig's merged target regions, reg[reg_off .. +n_reg) = starts, the next n_reg = ends (regions_bed.c)
    uint32_t name_off; int32_t name_fixed_len;     // "@[prefix_]contig" in the group's name pool (zero padded to >= 272 bytes)
    uint32_t contig_start;         // 1: the range begins its contig (first_ii == 0): the abort rule's counter starts from zero there (dwgsim.c:635)
    uint32_t pad_;
};

struct SimArgs {
    SimParams p;
    HapDev hap[2];                 // the group (pointers at group coordinate 0)
    const SimSeg *segs; int32_t n_seg; uint32_t n_blocks;     // the ranges of this launch (device memory) and its logical blocks
    uint64_t n_pairs;              // pairs of the launch
    const uint64_t *chain;         // device words chained from batch to batch on the context's stream: [0] random reads emitted before this batch (rand_ii, dwgsim.c:1042,1096), [1] the abort rule's carry
    const int32_t *reg; int32_t have_regions;       // -x: region pool of the group
    const uint64_t *e_thr[2];      // per-position error thresholds ceil((e.start + e.by*i) * 2^32) (dwgsim.c:237): u < e  <=>  w < thr
    const uint16_t *summ[2];       // k_place only (else null): per SUMM_CELLS cells of a haplotype, bits 0-7 = INSERT / DELETE cells, bit 15 = a base code >= 4
    const uint16_t *summ2[2];      // ... and per SUMM2_CELLS cells
    int32_t place_fast, place_k;   // k_place: 1 = a pair can be decided from the coarse summaries under any insert size within dist +- place_k (no -x, no -a); place_k bounds |normal| * std_dev
    uint32_t *place_list, *place_list_n; uint32_t place_list_cap;      // k_place -> k_place_rest: PLACE_LISTS lists of place_list_cap pair numbers, their lengths (16 words apart)
    uint64_t *range_rand;          // k_place: random reads per range of the launch
    const uint32_t *e_thr32[2];    // the same as 32-bit words, zero padded to a multiple of 8 entries; a threshold of 2^32 (e = 1) is stored as
    int32_t e_full;                // 0xFFFFFFFF and flagged here: those positions always err
    const uint32_t *qbase[2];      // per-position base quality characters (dwgsim.c:907; signed-char semantics) packed four to a word: len entries, then the last one
    int32_t qb_words;              // repeated up to qb_words words (>= len + 4 entries, the same for both read ends); the kernel stages both tables in LDS
    const uint8_t *names;          // name pool of the group
    const uint8_t *rand_fixed; int32_t rand_fixed_len;   // "[prefix_]rand"
    uint32_t *meta;                // per pair: failed attempts | contig start << 30 | random read << 31 (input of the abort rule, k_failrule)
    uint32_t *block_rand;          // per 128-pair block: random pairs (k_place), then exclusive prefix (k_scan)
    uint64_t *counters;            // this batch's slot: [0] ticket, [1] retries, [2] fail flags, [3] total random, [4..6] stream bytes, [16..19] abort-rule segment of the batch, [20] abort, [21] carry out
    uint64_t *status[4];           // look-back words: record bytes of stream BWA1 / BWA2, random-read count, (SOLiD) BFAST bytes
    uint8_t *out[3];               // packed FASTQ text: bwa read1, bwa read2, bfast
    int32_t lds_words;             // uint32 words of packed bases per lane: 8 bases per word; Ion Torrent: 16 per word, the flow model's one in-place buffer
    int32_t cap;                   // Ion Torrent: capacity (bases) of a read after flow errors = 16 lds_words
    int32_t ion_lds;               // Ion Torrent: 1 = the read buffers are in LDS (k_simulate<.., 3>), 0 = in scratch slots (k_simulate<.., 2>)
    int32_t flow_stack_words;      // Ion Torrent: LDS words per lane of the pass-2 run stack (two runs per word)
    int32_t fifo;                 // 1: the records leave through the per-lane LDS FIFO (32-byte aligned bursts); 0: 16-byte pieces straight from registers (when the FIFO would cost a block per CU)
    int32_t sim_threads;          // lanes per k_simulate block chosen by the host: SIM_THREADS, or SIM_THREADS_LONG for long reads
    int32_t flow_len;              // Ion Torrent: length of the flow order (<= 64)
    int32_t split;                 // 1: k_simulate runs as two kernels (dw_simulate.hip SPLIT) handing their state over through the arrays below
    uint32_t *split_state;         // per block [lds_words][lanes]: the staged bases after the error phase
    uint32_t *split_hand;          // per lane 16 bytes: ext_coor | n_err, n_sub | n_indel, n_ins | attempt, flags
    uint32_t *split_agg;           // per block 16 bytes: random pairs, bytes of stream 1 / 2 (without random reads' hexadecimal digits)
    uint64_t *split_pre;           // per block 4 words: random reads / bytes of stream 1 / bytes of stream 2 in front of it inside its chunk of 1024 blocks (k_split_scan1)
    uint64_t *split_chunk;         // per chunk 4 words: the chunk's sums (k_split_scan1), then the sums in front of the chunk (k_split_scan2)
    uint32_t *flow_scratch;        // scratch slots: the staged reads of one block per SLOT (Ion Torrent DT = 2: flow_words_per_lane words per lane), word w of lane t at [w * nthr + t]
    uint64_t *flow_free;           // ... the slots' free lists, one per XCD (dw_simulate.hip scratch_slot_take): 256 header words + 8 x n_blocks queue words, zeroed per launch
    int32_t flow_slots;            // ... slots per XCD (8 x flow_slots slots in flow_scratch)
    int32_t lb_shift;              // the single Illumina kernel's one look-back word: random reads << lb_shift | bytes of stream 1 (dw_simulate.hip ONE_LB)
    const uint8_t *flow;           // Ion Torrent: flow order as base codes (dwgsim_opt.c:404-407), device memory, 64 bytes, followed by the log2 table of the gap draws (FLOW_LG_ENTRIES words)
    uint64_t flow_gap_r[2]; int32_t flow_gap_s[2];      // Ion Torrent, per read end: flow_gap_params of its threshold e_thr[j][0]
    uint64_t err_thr_max[2], err_gap_r[2]; int32_t err_gap_s[2], err_ramp[2];      // Illumina / SOLiD, per read end: the largest threshold of the error ramp, flow_gap_params of it, and whether any position's threshold is lower (those sites are thinned)
};

} // namespace dw
