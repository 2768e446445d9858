// dw_gzip.hip -- gzip on the GPU for the packed FASTQ text (replaces the gzprintf / gzputc stream of the reference, src/dwgsim.c:919-981,
// files opened with gzopen at :1150-1158): the text never crosses PCIe uncompressed and the host only writes bytes.
//
// One workgroup = one 32 KiB chunk of a stream = one complete gzip member (RFC 1952), so the members of a batch are independent and their
// concatenation is a valid .gz whose decompressed bytes are exactly the text (what the reference's own test compares, testdata/test.sh:23-25):
//     header with an FNAME field of 0-3 pad characters (so that EVERY member is a multiple of 4 bytes and starts on a word of the output)
//     | one DEFLATE block with dynamic Huffman codes (RFC 1951 3.2.7) | an empty stored block (final, byte-aligns) | CRC-32 and length.
//
// What is coded.  A FASTQ record is three populations of symbols -- name, bases, qualities -- under ONE Huffman table per block, and a block
// header costs more than a read's worth of savings, so the table cannot follow them.  Bases and qualities of simulated reads are random: an
// LZ77 window finds nothing there that a literal does not code as well.  The NAME line is the exception: its symbols are the rare ones of the
// table (6-7 bits each where digits are not also quality characters) and most of it repeats the name line of the record before -- "@contig_",
// the strand / random-read flags, ":0:0_" groups, the leading digits of the running index -- or itself (the second position repeats the leading
// digits of the first).  So name lines, and only they, are searched for matches, at a handful of distances the structure suggests rather than
// through a hash table: the same column and the same distance from the line's end in the line FOUR LINES UP, and the distances 6 .. 11 inside
// the line.  Round 3 coded literals only: 0.493 of the text on the bench workload (0.479 on the oracle's sample); with the name-line matches
// and run-length coded code lengths in the block header: see DESIGN.md section 6b.
//
// A lane owns 128 consecutive bytes and keeps them in registers (8 x dwordx4) for all passes:
//   lines   '\n' masks of the spans -> scan -> the chunk's line starts in LDS
//   parse   name lines inside the span: greedy matches (>= 3 bytes, never across the span's or the line's end), up to GZ_MAXM per span, kept in LDS
//   pass 1  histograms of literals / lengths / distances (LDS atomics into 8 sub-histograms) + CRC-32 of the span (slicing-by-4)
//   codes   two Huffman codes (literals / lengths, distances): used symbols rank-sorted, one lane's two-queue merge per alphabet, depths by walking
//           the parents; counts halved until no code is longer than 15 bits; canonical codes with ranks by ballots; the code lengths of the
//           block header run-length coded (symbols 16 / 17 / 18), every run by the lane that starts it
//   pass 2  bits per span -> block scan -> decoupled look-back over the chunks for the member's byte offset
//   pass 3  every lane packs its span's codes LSB-first into an IMAGE OF THE MEMBER IN LDS (OR for the words two spans share); CRC-32 of
//           the spans joined by a tree of x^(8 L) shifts; then the image leaves with plain, coalesced word stores.
// A chunk whose member would not fit the image (24 KB: text that does not compress to 3/4) leaves as a STORED block instead.
#include "dw_device.hpp"
#include "dw_launch.hpp"

namespace dw {

constexpr int GZ_CHUNK = 32768, GZ_THREADS = 256, GZ_SPAN = GZ_CHUNK / GZ_THREADS;      // 128 bytes per lane
constexpr int GZ_IMG_WORDS = 24576 / 4;                    // the member image; what does not fit is stored
constexpr int GZ_LINES = 1536;                             // line starts kept per chunk (later lines are coded as literals)
constexpr int GZ_MAXM = 8;                                 // matches kept per span
constexpr int GZ_NLIT = 286, GZ_NDIST = 30;
constexpr int GZ_FIXED_HDR_BITS = 3 + 5 + 5 + 4 + 19 * 3;  // BFINAL, BTYPE, HLIT, HDIST, HCLEN, the 19 code-length code lengths
constexpr int GZ_MIN_MATCH = 3;

struct GzArgs {
    const uint8_t *text; const uint64_t *n_dev;   // the stream; its length is still on the device when the kernel is enqueued
    uint32_t *out; uint64_t cap;            // output (4-byte aligned), cap bytes
    uint64_t *flags;                        // |= 8 when a member would not fit
    uint64_t *status;                       // look-back words, one per chunk (zeroed)
    uint64_t *ticket;                       // zeroed
    uint64_t *total;                        // out: compressed bytes of the stream (written by the last chunk)
    const uint32_t *crc_slice;              // [4][256] slicing-by-4 tables of CRC-32 (reflected 0xEDB88320)
    const uint32_t *crc_shift;              // [16][4][256]: multiply by x^(8 * 2^m) mod P, m = 0..15 (append 2^m zero bytes)
};

DW_DEV uint32_t crc_apply_shift(const uint32_t *t, uint32_t v)
{
    return t[v & 255u] ^ t[256 + ((v >> 8) & 255u)] ^ t[512 + ((v >> 16) & 255u)] ^ t[768 + (v >> 24)];
}
DW_DEV uint32_t crc_append_zeros(const uint32_t *tables, uint32_t v, uint32_t nbytes)     // the register after nbytes more zero bytes
{
    for (uint32_t m = 0; nbytes; ++m, nbytes >>= 1) if (nbytes & 1u) v = crc_apply_shift(tables + m * 1024, v);
    return v;
}
DW_DEV uint32_t bit_reverse(uint32_t code, uint32_t len) { return len ? __builtin_bitreverse32(code) >> (32u - len) : 0u; }     // len 0..15

// DEFLATE length / distance symbols (RFC 1951 3.2.5), computed: sym | extra bits << 8 | extra value << 16
DW_DEV uint32_t gz_len_sym(uint32_t len)              // 3 .. 258
{
    if (len <= 10) return len - 3u;
    if (len == 258) return 28u;
    const uint32_t v = len - 3u, e = 29u - (uint32_t)__builtin_clz(v);       // v in [2^(e+2), 2^(e+3)): e extra bits
    return (4u * e + 4u + ((v >> e) & 3u)) | (e << 8) | ((v & ((1u << e) - 1u)) << 16);
}
DW_DEV uint32_t gz_dist_sym(uint32_t dist)            // 1 .. 32768
{
    if (dist <= 4) return dist - 1u;
    const uint32_t v = dist - 1u, e = 30u - (uint32_t)__builtin_clz(v);      // v in [2^(e+1), 2^(e+2)): e extra bits
    return (2u * e + 2u + ((v >> e) & 1u)) | (e << 8) | ((v & ((1u << e) - 1u)) << 16);
}

struct LdsBits {            // LSB-first bit packer into the (zeroed) LDS image; the first and the last word of a run may be shared with a neighbour, so
    uint32_t *img; uint32_t w; uint64_t acc; uint32_t nb;      // every word is ORed in (an LDS OR costs what a store costs; no "first word" case to tell apart)
    DW_DEV void init(uint32_t *image, uint32_t bitpos) { img = image; w = bitpos >> 5; nb = bitpos & 31u; acc = 0; }
    DW_DEV void put(uint32_t code, uint32_t len)
    {
        acc |= (uint64_t)code << nb; nb += len;
        if (nb >= 32) { atomicOr(&img[w], (uint32_t)acc); ++w; acc >>= 32; nb -= 32; }
    }
    DW_DEV void finish() { if (nb) atomicOr(&img[w], (uint32_t)acc); }
};

// bytes of the text at any address (gfx950 takes unaligned dword loads as they are)
struct __attribute__((packed, aligned(1))) GzUnal4 { uint32_t v; };
DW_DEV uint32_t gz_load4(const uint8_t *p) { return reinterpret_cast<const GzUnal4 *>(p)->v; }
// equal bytes of src[p ..] and src[p - d ..], at most lim (p + lim does not pass the line's or the span's end; the buffer is padded for the last dword)
DW_DEV uint32_t gz_match_len(const uint8_t *src, uint32_t p, uint32_t d, uint32_t lim)
{
    uint32_t l = 0;
    while (l < lim) {
        const uint32_t x = gz_load4(src + p + l) ^ gz_load4(src + p + l - d);
        if (x) { l += (uint32_t)__builtin_ctz(x) >> 3; break; }
        l += 4;
    }
    return l < lim ? l : lim;
}

// Huffman code lengths of TWO alphabets at once: A = up to 512 symbols (lane tid owns symbols tid and 256 + tid: literals / lengths), B = up to 64
// symbols owned by the lanes of wave 0 (distances).  h0 / h1 / hb = the counts of this lane's symbols.
// Round 3 merged the two lightest trees once per iteration with a block-wide reduction: three barriers per merge, ~ 60 merges -- half the kernel's
// time.  Now: the used symbols are gathered and RANK-SORTED by weight in parallel (a used symbol counts the keys below its own), ONE lane per
// alphabet runs the classic two-queue merge over the sorted leaves (no barrier, one LDS round trip per merge; lane 0 for A and lane 64 for B at
// the same time), and every leaf then finds its depth by walking its parents -- in parallel again.  Counts are flattened ((h >> scale) | 1) until no
// code is longer than 15 bits (Fibonacci-like counts only).  An alphabet with a single used symbol gives it length 1 (RFC 1951: one distance code is
// sent with one bit), none leaves all lengths 0.  Work arrays (LDS): A: key[288] sorted[288] nodew[288] parent[576]; B: key[32] sorted[32] nodew[32] parent[64].
struct GzTreeMem { uint32_t *keyA, *sortA, *nodeA, *parA, *lenA, *keyB, *sortB, *nodeB, *parB, *lenB; };
DW_DEV void gz_two_queue(const uint32_t *sorted, uint32_t n, uint32_t *nodew, uint32_t *parent)      // leaves 0 .. n-1 (ascending keys: weight << 9 | symbol), internal node k = n + k
{
    uint32_t li = 0, ni = 0, nn = 0;
    uint32_t lw = sorted[0] >> 9;                                 // weight at the head of the leaf queue
    uint32_t nw = 0xFFFFFFFFu;                                    // ... of the node queue (empty)
    for (uint32_t k = 0; k + 1 < n; ++k) {
        uint32_t sum = 0;
#pragma unroll
        for (int pick = 0; pick < 2; ++pick) {                    // a leaf wins a tie: shallower trees
            if (li < n && lw <= nw) { sum += lw; parent[li] = n + nn; ++li; lw = li < n ? sorted[li] >> 9 : 0xFFFFFFFFu; }
            else { sum += nw; parent[n + ni] = n + nn; ++ni; nw = ni < nn ? nodew[ni] : 0xFFFFFFFFu; }
        }
        nodew[nn] = sum;
        if (ni == nn) nw = sum;                                   // the node queue was empty: the new node is its head
        ++nn;
    }
}
DW_DEV void gz_code_lengths2(uint32_t h0, uint32_t h1, uint32_t hb, const GzTreeMem &m, uint32_t *s_scan, uint32_t (*s_min)[4][2])
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t scaleA = 0, scaleB = 0; bool doneA = false, doneB = false;
    for (;;) {
        const uint32_t w0 = h0 ? (scaleA ? ((h0 >> scaleA) | 1u) : h0) : 0u, w1 = h1 ? (scaleA ? ((h1 >> scaleA) | 1u) : h1) : 0u;
        const uint32_t wb = (tid < 64 && hb) ? (scaleB ? ((hb >> scaleB) | 1u) : hb) : 0u;
        // the used symbols of each alphabet, gathered (order does not matter: they are sorted next)
        uint32_t nA, nB;
        {
            uint32_t tot; const uint32_t mine = (w0 ? 1u : 0u) + (w1 ? 1u : 0u) + (wb ? 0x10000u : 0u);
            const uint32_t ex = block_excl_scan(mine, s_scan, &tot);
            nA = tot & 0xFFFFu; nB = tot >> 16;
            if (!doneA) { uint32_t q = ex & 0xFFFFu; if (w0) m.keyA[q++] = (w0 << 9) | (uint32_t)tid; if (w1) m.keyA[q] = (w1 << 9) | (256u + (uint32_t)tid); m.lenA[tid] = 0; m.lenA[256 + tid] = 0; }
            if (!doneB && tid < 64) { if (wb) m.keyB[ex >> 16] = (wb << 9) | (uint32_t)tid; m.lenB[tid] = 0; }
        }
        __syncthreads();
        // rank sort: keys are distinct (the symbol is part of them)
        if (!doneA) for (uint32_t i = (uint32_t)tid; i < nA; i += GZ_THREADS) { const uint32_t key = m.keyA[i]; uint32_t r = 0; for (uint32_t q = 0; q < nA; ++q) r += m.keyA[q] < key ? 1u : 0u; m.sortA[r] = key; }
        if (!doneB && wave == 1 && (uint32_t)lane < nB) { const uint32_t key = m.keyB[lane]; uint32_t r = 0; for (uint32_t q = 0; q < nB; ++q) r += m.keyB[q] < key ? 1u : 0u; m.sortB[r] = key; }
        __syncthreads();
        if (!doneA && tid == 0 && nA > 1) gz_two_queue(m.sortA, nA, m.nodeA, m.parA);
        if (!doneB && tid == 64 && nB > 1) gz_two_queue(m.sortB, nB, m.nodeB, m.parB);
        __syncthreads();
        // depths: leaf i of the sorted order walks up to the root (node 2n - 2)
        uint32_t mxa = 0, mxb = 0;
        if (!doneA) {
            if (nA == 1) { if (tid == 0) m.lenA[m.sortA[0] & 511u] = 1; }
            else for (uint32_t i = (uint32_t)tid; i < nA; i += GZ_THREADS) { uint32_t d = 1, p = m.parA[i]; while (p != 2 * nA - 2) { p = m.parA[p]; ++d; } m.lenA[m.sortA[i] & 511u] = d; mxa = d > mxa ? d : mxa; }
        }
        if (!doneB && wave == 1) {
            if (nB == 1) { if (lane == 0) m.lenB[m.sortB[0] & 511u] = 1; }
            else if ((uint32_t)lane < nB) { uint32_t d = 1, p = m.parB[lane]; while (p != 2 * nB - 2) { p = m.parB[p]; ++d; } m.lenB[m.sortB[lane] & 511u] = d; mxb = d; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const uint32_t oa = (uint32_t)__shfl_xor((int)mxa, off), ob = (uint32_t)__shfl_xor((int)mxb, off); mxa = oa > mxa ? oa : mxa; mxb = ob > mxb ? ob : mxb; }
        if (lane == 0) { s_min[0][wave][0] = mxa; s_min[0][wave][1] = mxb; }
        __syncthreads();
        mxa = s_min[0][0][0]; mxb = s_min[0][0][1];
        for (int w = 1; w < 4; ++w) { mxa = s_min[0][w][0] > mxa ? s_min[0][w][0] : mxa; mxb = s_min[0][w][1] > mxb ? s_min[0][w][1] : mxb; }
        __syncthreads();
        if (!doneA) { if (mxa <= 15) doneA = true; else ++scaleA; }      // else flatten the histogram and build again (equal weights give depth <= 9)
        if (!doneB) { if (mxb <= 15) doneB = true; else ++scaleB; }
        if (doneA && doneB) return;
    }
}
// canonical codes (RFC 1951 3.2.2) of the lengths in len[0 .. 512) (lane tid: symbols tid and 256 + tid), stored bit-reversed for LSB-first packing:
// code[s] = reversed code | length << 16.  A symbol's rank among the symbols of its length: ballots per length inside its group of 64, the groups
// before it through a small table (round 3 counted them in a loop over all smaller symbols: 256 LDS reads per lane).  n_lo / n_hi: codes kept.
DW_DEV void gz_canonical(const uint32_t *len, uint32_t *code, int n_lo, int n_hi, uint32_t *s_blc, uint32_t *s_next, uint32_t (*s_cnt)[8])
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t l0 = len[tid], l1 = len[256 + tid];
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;  // the lanes in front of this one
    uint32_t r0 = 0, r1 = 0;
    for (uint32_t L = 1; L <= 15; ++L) {                          // group g = wave: symbols 64 g .. 64 g + 63; group 4 + wave: symbols 256 + 64 wave ..
        const uint64_t b0 = __ballot(l0 == L), b1 = __ballot(l1 == L);
        if (l0 == L) r0 = (uint32_t)__popcll(b0 & below);
        if (l1 == L) r1 = (uint32_t)__popcll(b1 & below);
        if (lane == 0) { s_cnt[L][wave] = (uint32_t)__popcll(b0); s_cnt[L][4 + wave] = (uint32_t)__popcll(b1); }
    }
    __syncthreads();
    if (tid < 16) { uint32_t c = 0; if (tid) for (int g = 0; g < 8; ++g) c += s_cnt[tid][g]; s_blc[tid] = c; }
    __syncthreads();
    if (tid == 0) { uint32_t c = 0; s_next[0] = 0; for (int bits = 1; bits <= 15; ++bits) { c = (c + (bits > 1 ? s_blc[bits - 1] : 0u)) << 1; s_next[bits] = c; } }
    __syncthreads();
    if (l0) for (int g = 0; g < wave; ++g) r0 += s_cnt[l0][g];
    if (l1) for (int g = 0; g < 4 + wave; ++g) r1 += s_cnt[l1][g];
    if (tid < n_lo) code[tid] = l0 ? (bit_reverse(s_next[l0] + r0, l0) | (l0 << 16)) : 0u;
    if (tid < n_hi) code[256 + tid] = l1 ? (bit_reverse(s_next[l1] + r1, l1) | (l1 << 16)) : 0u;
    __syncthreads();
}

__global__ void __launch_bounds__(GZ_THREADS, 3) k_gzip(GzArgs a)
{
    // one raw array: during the parse it holds the TEXT of the chunk (name lines are compared with the lines above them at LDS latency: from HBM,
    // where the text has just been written by k_simulate, every dependent compare cost a microsecond and the kernel ran at 75 instead of 260 GB/s);
    // afterwards it is the member image (24 KB) followed by the sub-histograms (9 KB)
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[GZ_IMG_WORDS + 8 * 288];
    static_assert((GZ_IMG_WORDS + 8 * 288) * 4 >= GZ_CHUNK + 16, "the chunk's text fits where the image and the histograms will be");
    uint32_t *const s_img = s_raw;
    uint32_t (*const s_sub)[288] = reinterpret_cast<uint32_t (*)[288]>(s_raw + GZ_IMG_WORDS);      // sub-histograms of literals / lengths (lane & 7), then the work arrays of the code construction
    const uint8_t *const s_text = reinterpret_cast<const uint8_t *>(s_raw);
    __shared__ uint32_t s_dsub[64];           // distance histogram, then its code lengths
    __shared__ uint32_t s_code[288];          // bit-reversed canonical code | length << 16 of literals / lengths
    __shared__ uint32_t s_dcode[32];          // ... of distances
    __shared__ uint32_t s_tab[1024];          // CRC slicing tables, later the span CRCs ([0..255])
    __shared__ uint32_t s_tok[GZ_MAXM][GZ_THREADS];      // matches of a span: start in the span | length << 7 | distance << 14
    __shared__ uint16_t s_tokp[GZ_MAXM][GZ_THREADS];     // ... their bits (pass 2 -> pass 3a), then where they start in the lane's bit stream (pass 3a -> 3b)
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_min[2][4][2];       // the two lightest trees of every wave (double-buffered by merge parity)
    __shared__ uint32_t s_wb[4][64];          // work arrays of the distance code: keys + sorted keys, node weights, parents, lengths
    __shared__ uint32_t s_cnt[16][8];         // canonical codes: symbols per code length and group of 64
    __shared__ uint32_t s_blc[16], s_next[16];
    __shared__ uint32_t s_hdr[4];             // bits of the run-length coded code lengths, HLIT, HDIST, their tokens
    __shared__ uint64_t s_mask[6];            // "code length != 0" of literals / lengths (286 bits) and distances (30 bits)
    __shared__ uint32_t s_ticket; __shared__ uint64_t s_base;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = (uint32_t)atomicAdd((unsigned long long *)a.ticket, 1ull);
    uint16_t *const s_ls = reinterpret_cast<uint16_t *>(s_tab);  // line starts of the chunk: until the parse is over (then the CRC tables are loaded there)
    static_assert((GZ_LINES + 2) * 2 <= 1024 * 4, "the line starts fit where the CRC tables will be");
    if (tid < 64) s_dsub[tid] = 0;
    __syncthreads();
    const uint32_t t = s_ticket;                                  // logical chunk: its predecessors have started
    const uint64_t n_text = *a.n_dev, c0 = (uint64_t)t * GZ_CHUNK;
    if (c0 >= n_text) return;                                     // (the grid is sized for the buffer's capacity)
    const uint32_t clen = (uint32_t)(n_text - c0 < (uint64_t)GZ_CHUNK ? n_text - c0 : (uint64_t)GZ_CHUNK);
    const uint8_t *src = a.text + c0;
    const uint32_t s0 = (uint32_t)tid * GZ_SPAN, slen = s0 >= clen ? 0u : (clen - s0 < (uint32_t)GZ_SPAN ? clen - s0 : (uint32_t)GZ_SPAN);

    // ---- the lane's span: 32 words in registers (chunks and spans are 16-byte aligned in the text buffer) ----
    uint32_t d[GZ_SPAN / 4];
#pragma unroll
    for (int q = 0; q < GZ_SPAN / 16; ++q) {
        if ((uint32_t)(16 * q + 16) <= slen) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + s0 + 16 * q);
            d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t w = 0;
                for (uint32_t b = 0; b < 4; ++b) { const uint32_t i = 16u * q + 4u * k + b; if (i < slen) w |= (uint32_t)src[s0 + i] << (8 * b); }
                d[4 * q + k] = w;
            }
        }
    }

    // the chunk's text into LDS: word i * 256 + tid by lane tid (coalesced from the L2, where the loads above have just brought it; conflict-free in LDS)
    {
        const uint32_t nw = (clen + 3u) >> 2;
        if (!probe::off(1 << 26))      // (analysis builds: 2^26 = the text is not brought to LDS: only with 2^25)
        for (uint32_t q = (uint32_t)tid; q < (uint32_t)(GZ_CHUNK / 4 + 4); q += GZ_THREADS) s_raw[q] = q < nw ? reinterpret_cast<const uint32_t *>(src)[q] : 0u;      // (the buffer is padded: whole words)
    }
    // ---- lines: newline masks of the span (bit i of nl[i >> 5] = byte i is '\n'), the chunk's line starts ----
    uint32_t nl[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < GZ_SPAN / 4; ++k) {
        const uint32_t x = d[k] ^ 0x0A0A0A0Au, z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);      // 0x80 in every zero byte of x
        const uint32_t m = ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
        nl[k >> 3] |= m << (4 * (k & 7));
    }
    // (bytes past the text are zero in d[], not '\n')
    uint32_t nlines_before, nl_total;
    {
        const uint32_t cnt = (uint32_t)(__popc(nl[0]) + __popc(nl[1]) + __popc(nl[2]) + __popc(nl[3]));
        nlines_before = block_excl_scan(cnt, s_scan, &nl_total);
    }
    if (tid == 0) s_ls[0] = 0;
    {
        uint32_t k = nlines_before + 1;                            // line k starts behind the k-th '\n'
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t m = nl[w];
            while (m) { const uint32_t b = (uint32_t)__builtin_ctz(m); m &= m - 1; if (k <= (uint32_t)GZ_LINES) s_ls[k] = (uint16_t)(s0 + 32u * (uint32_t)w + b + 1u); ++k; }
        }
    }
    __syncthreads();
    // lines 0 .. n_lines-1 start inside the chunk; line k ends (exclusive, with its '\n') at ls_end(k)
    const uint32_t n_lines_all = nl_total + 1u - ((clen > 0 && nl_total > 0 && s_text[clen - 1] == '\n') ? 1u : 0u);
    const uint32_t n_lines = n_lines_all < (uint32_t)GZ_LINES ? n_lines_all : (uint32_t)GZ_LINES;
    auto ls_start = [&](uint32_t k) -> uint32_t { return s_ls[k]; };
    auto ls_end = [&](uint32_t k) -> uint32_t { return k + 1 < n_lines_all && k + 1 <= (uint32_t)GZ_LINES ? (uint32_t)s_ls[k + 1] : clen; };

    if (probe::off(1 << 20)) return;      // (analysis builds, tools/r04_gz_knock.sh: the kernel up to here -- load, text to LDS, line starts)
    // ---- parse: name lines inside the span ----
    // A line is taken for a name line when it starts with '@' and the line two above (or, at the top of the chunk, two below) starts with '+':
    // a wrong guess costs compares, never correctness -- every match is verified byte by byte.
    // The lanes of a wave parse IN STEP (a first version let every lane walk its own lines and positions, nested loops of different lengths: the wave
    // paid the sum, 160 of the kernel's 250 us per block): (1) every lane finds the pieces of name lines inside its span -- at most two, a few lines to
    // look at; (2) for a piece of up to 64 bytes, the bytes that equal the byte `d` back are gathered as a 64-bit mask per candidate distance
    // (eight candidates, sixteen dword compares each, the same code for every lane); (3) a greedy walk over the masks: the longest run of set bits at
    // the current position over the eight masks is a match if it has three bytes -- count-trailing-zeros, no memory.
    uint32_t cov[4] = {0, 0, 0, 0};                               // bytes of the span covered by matches
    uint32_t mst[4] = {0, 0, 0, 0};                               // ... and where the matches start
    uint32_t n_tok = 0;
    if (!probe::off(1 << 25)) {                                   // (analysis builds: 2^25 = no parse, every byte a literal)
        // (1) the name-line pieces of this span: start (in the chunk), length (<= 64), distance A and B (0 = none)
        uint32_t pa[2] = {0, 0}, pn[2] = {0, 0}, pdA[2] = {0, 0}, pdB[2] = {0, 0};
        {
            const uint32_t span_end = s0 + slen;
            uint32_t k = nlines_before, found = 0;
            for (int it = 0; it < 6; ++it) {                     // (a span of more than six lines holds no FASTQ names worth the search)
                const bool live = slen && k < n_lines && found < 2u && ls_start(k < n_lines ? k : 0u) < span_end;
                if (!__ballot(live)) break;
                if (live) {
                    const uint32_t st = ls_start(k), le = ls_end(k);
                    bool name = s_text[st] == '@';
                    if (name) { if (k >= 2) name = s_text[ls_start(k - 2)] == '+'; else if (k + 2 < n_lines) name = s_text[ls_start(k + 2)] == '+'; }
                    if (name) {
                        const uint32_t a0 = st > s0 ? st : s0, b0 = le < span_end ? le : span_end;
                        uint32_t dA = 0, dB = 0;
                        if (k >= 4) { dA = st - ls_start(k - 4); dB = le - ls_end(k - 4); if (dB == dA) dB = 0; }
                        if (b0 > a0) { const uint32_t nn = b0 - a0 < 64u ? b0 - a0 : 64u; if (found == 0) { pa[0] = a0; pn[0] = nn; pdA[0] = dA; pdB[0] = dB; } else { pa[1] = a0; pn[1] = nn; pdA[1] = dA; pdB[1] = dB; } ++found; }
                    }
                    ++k;
                }
            }
        }
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const uint32_t a0 = pa[seg], nn = pn[seg];
            if (!__ballot(nn != 0)) continue;                     // (2 x 150 bp records: a span never holds two name lines)
            // (2) equality masks: bit i of E[c] = byte a0 + i equals byte a0 + i - d_c (and that byte lies inside the chunk)
            const uint32_t dc[8] = {pdA[seg], pdB[seg], 6u, 7u, 8u, 9u, 10u, 11u};
            uint64_t E[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
            for (int jw = 0; jw < 16; ++jw) {
                const uint32_t pos = a0 + 4u * (uint32_t)jw;
                const uint32_t cur = gz_load4(s_text + pos);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t d = dc[c];
                    uint32_t m = 0;
                    if (d && pos >= d && 4u * (uint32_t)jw < nn) {
                        const uint32_t x = cur ^ gz_load4(s_text + pos - d), z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);      // 0x80 in every zero byte of x
                        m = ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
                    }
                    E[c] |= (uint64_t)m << (4 * jw);
                }
            }
            // (3) greedy walk.  `any3`: the positions where some candidate has three equal bytes in a row -- the walk jumps from one to the next
            uint64_t any3 = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) any3 |= E[c] & (E[c] >> 1) & (E[c] >> 2);
            if (nn < 64u) any3 &= (1ull << nn) - 1ull;
            uint32_t p = 0;
            while (__ballot((any3 >> (p < 63u ? p : 63u)) != 0 && p < nn && n_tok < (uint32_t)GZ_MAXM)) {
                if ((any3 >> (p < 63u ? p : 63u)) != 0 && p < nn && n_tok < (uint32_t)GZ_MAXM) {
                    p += (uint32_t)__builtin_ctzll(any3 >> p);                               // the next position with a match of three bytes or more
                    uint32_t best = 0, bd = 0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const uint64_t m = ~(E[c] >> p);                                     // (zeros shifted in from above end every run)
                        uint32_t r = (uint32_t)__builtin_ctzll(m | (1ull << 63));
                        if (r > nn - p) r = nn - p;
                        if (r > best) { best = r; bd = dc[c]; }
                    }
                    if (best >= (uint32_t)GZ_MIN_MATCH) {
                        const uint32_t o = a0 + p - s0;
                        s_tok[n_tok][tid] = o | (best << 7) | (bd << 14);      // start in the span (7 bits) | length 3 .. 64 (7 bits) | distance
                        ++n_tok;
                        mst[o >> 5] |= 1u << (o & 31u);           // (dynamic word index: four registers, resolved by selects)
                        for (uint32_t b = o; b < o + best;) {     // bits o .. o + best - 1 of cov
                            const uint32_t w = b >> 5, lo = b & 31u, n = (32u - lo < o + best - b) ? 32u - lo : o + best - b;
                            const uint32_t mm = (n == 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << lo;
                            cov[w] |= mm; b += n;
                        }
                        p += best;
                    } else ++p;                                   // (three equal bytes that run past the piece's end)
                }
            }
        }
    }

    __syncthreads();                                              // every lane is done with the text: the array becomes image + histograms
    for (int q = tid; q < GZ_IMG_WORDS + 8 * 288; q += GZ_THREADS) s_raw[q] = 0;
    for (int q = tid; q < 1024; q += GZ_THREADS) s_tab[q] = a.crc_slice[q];      // (the line starts are done with)
    __syncthreads();

    if (probe::off(1 << 21)) return;      // ... + parse
    // ---- pass 1: histograms + CRC-32 of the span (register started from 0; the chunk's init value enters with lane 0) ----
    uint32_t crc = tid == 0 ? 0xFFFFFFFFu : 0u;
    {
        uint32_t *hist = s_sub[tid & 7];
#pragma unroll
        for (int k = 0; k < GZ_SPAN / 4; ++k) {
            const uint32_t w = d[k], base = 4u * (uint32_t)k;
            const uint32_t cv = (cov[k >> 3] >> (4 * (k & 7))) & 15u;
            if (base + 4 <= slen) {
                // (one predicated form for every word: a "no match in this word" fast path beside it made the wave run both, some lane always has a match)
                if (!(cv & 1u)) atomicAdd(&hist[w & 255u], 1u);
                if (!(cv & 2u)) atomicAdd(&hist[(w >> 8) & 255u], 1u);
                if (!(cv & 4u)) atomicAdd(&hist[(w >> 16) & 255u], 1u);
                if (!(cv & 8u)) atomicAdd(&hist[w >> 24], 1u);
                const uint32_t x = crc ^ w;
                crc = s_tab[768 + (x & 255u)] ^ s_tab[512 + ((x >> 8) & 255u)] ^ s_tab[256 + ((x >> 16) & 255u)] ^ s_tab[x >> 24];
            } else if (base < slen) {
                for (uint32_t b = 0; base + b < slen; ++b) { const uint32_t c = (w >> (8 * b)) & 255u; if (!((cv >> b) & 1u)) atomicAdd(&hist[c], 1u); crc = s_tab[(crc ^ c) & 255u] ^ (crc >> 8); }
            }
        }
        for (uint32_t q = 0; q < (uint32_t)GZ_MAXM; ++q) {
            if (!__ballot(q < n_tok)) break;
            if (q < n_tok) {
                const uint32_t tk = s_tok[q][tid];
                atomicAdd(&hist[257u + (gz_len_sym((tk >> 7) & 127u) & 255u)], 1u);
                atomicAdd(&s_dsub[gz_dist_sym(tk >> 14) & 255u], 1u);
            }
        }
    }
    __syncthreads();

    if (probe::off(1 << 22)) return;      // ... + histograms and CRC
    // ---- Huffman code lengths: lane k owns symbols k and 256 + k (256 = end of block, 257 .. 285 = lengths) ----
    // work arrays of the code construction, where the sub-histograms were: A (literals / lengths) key 288 | sorted 288 | node weights 288 | parents 576 | lengths 512
    uint32_t *const wk = &s_sub[0][0];
    const GzTreeMem tm{wk, wk + 288, wk + 576, wk + 864, wk + 1440, s_wb[0], s_wb[0] + 32, s_wb[1], s_wb[2], s_wb[3]};
    uint32_t *const len = wk + 1440;
    uint32_t h0 = 0, h1 = 0;
    for (int q = 0; q < 8; ++q) { h0 += s_sub[q][tid]; if (tid < GZ_NLIT - 256) h1 += s_sub[q][256 + tid]; }
    if (tid == 0) h1 = 1;                                         // end of block: once
    const uint32_t hd = tid < GZ_NDIST ? s_dsub[tid] : 0u;
    __syncthreads();                                              // (the sub-histograms become work arrays)
    gz_code_lengths2(h0, h1, hd, tm, s_scan, s_min);
    __syncthreads();
    gz_canonical(len, s_code, 256, 32, s_blc, s_next, s_cnt);
    if (tid < 64) s_dsub[tid] = tid < GZ_NDIST ? s_wb[3][tid] : 0u;      // the distance code lengths: s_dsub[0 .. 30), and as a 512-entry array for gz_canonical
    const uint32_t my_len = len[tid], my_len1 = len[256 + tid];
    __syncthreads();
    len[tid] = tid < GZ_NDIST ? s_dsub[tid] : 0u; len[256 + tid] = 0;
    __syncthreads();
    gz_canonical(len, s_dcode, 32, 0, s_blc, s_next, s_cnt);
    len[tid] = my_len; len[256 + tid] = my_len1;                  // literal / length code lengths back into len[] for the header
    __syncthreads();

    if (probe::off(1 << 23)) return;      // ... + the two codes
    // ---- block header: the code lengths, run-length coded (RFC 1951 3.2.7) under a FIXED code-length code: symbols 0 .. 12 take 4 bits (codes
    // 0 .. 12), 13 .. 18 take 5 bits (codes 26 .. 31) ----
    // Coded in parallel: a RUN of equal lengths is the unit (RFC 1951's repeat codes never look across one).  Every symbol that starts a run -- ballots
    // of "differs from the symbol before" -- finds the run's end in the ballot masks, counts its tokens and bits, a scan in symbol order gives it
    // its place, and it writes its own tokens (code-length code + extra bits, <= 12 bits each, with their bit offsets) to LDS for pass 3.  The literal /
    // length lengths are one sequence (first symbols by all lanes, then symbols 256 .. by wave 0), the distance lengths another (wave 0 again).
    // (One lane walking the sequence, even from non-zero to non-zero by count-trailing-zeros, was a tenth of the kernel: profiles/r04_gzip.txt.)
    uint32_t *const s_clt = s_tab + 256;                          // <= 320 tokens: value | bits << 12 | bit offset << 16 (the slicing tables are done with; the span CRCs use s_tab[0 .. 256))
    {
        if (wave == 0) { const uint64_t m1 = __ballot(lane < GZ_NLIT - 256 && len[256 + lane] != 0), m2 = __ballot(lane < GZ_NDIST && s_dsub[lane] != 0); if (lane == 0) { s_mask[4] = m1; s_mask[5] = m2; } }
    }
    __syncthreads();
    const uint32_t hlit = (s_mask[4] >> 1) ? 256u + 64u - (uint32_t)__builtin_clzll(s_mask[4]) : 257u;      // literal / length codes sent: up to the last used one (end-of-block always is)
    const uint32_t hdist = s_mask[5] ? 64u - (uint32_t)__builtin_clzll(s_mask[5]) : 1u;
    __syncthreads();                                              // (s_mask is reused for the run starts)
    {
        const uint32_t v0 = len[tid], p0 = tid ? len[tid - 1] : 0xFFFFu;
        const uint64_t st0 = __ballot(v0 != p0);
        if (lane == 0) s_mask[wave] = st0;
        if (wave == 0) {
            const uint64_t st1 = __ballot(lane < GZ_NLIT - 256 && len[256 + lane] != len[255 + lane]);
            const uint64_t st2 = __ballot(lane < GZ_NDIST && (lane == 0 || s_dsub[lane] != s_dsub[lane ? lane - 1 : 0]));
            if (lane == 0) { s_mask[4] = st1; s_mask[5] = st2; }
        }
    }
    __syncthreads();
    auto cl_code = [](uint32_t sym, uint32_t &nbits) -> uint32_t { nbits = sym <= 12u ? 4u : 5u; return bit_reverse(sym <= 12u ? sym : 26u + (sym - 13u), nbits); };
    // the end (exclusive) of the run that starts at symbol i of a sequence of n symbols whose run starts are the bits of mask[0 .. nwords)
    auto run_end = [&](const uint64_t *mask, uint32_t nwords, uint32_t i, uint32_t n) -> uint32_t {
        uint32_t e = n;
        for (uint32_t w = (i + 1) >> 6; w < nwords; ++w) {
            const uint64_t m = w == ((i + 1) >> 6) ? (mask[w] >> ((i + 1) & 63u)) << ((i + 1) & 63u) : mask[w];
            if (m) { const uint32_t q = 64u * w + (uint32_t)__builtin_ctzll(m); e = q < n ? q : n; break; }
        }
        return e;
    };
    // the tokens of one run of r lengths v: counted (write = false) or written from token `ntok` / bit `bits` on; returns tokens | bits << 16
    auto code_run = [&](uint32_t v, uint32_t run, bool write, uint32_t ntok, uint32_t bits) -> uint32_t {
        const uint32_t n0 = ntok, b0 = bits;
        auto emit = [&](uint32_t sym, uint32_t ebits, uint32_t eval) { uint32_t nb; const uint32_t c = cl_code(sym, nb); if (write) s_clt[ntok] = (c | (eval << nb)) | ((nb + ebits) << 12) | (bits << 16); ++ntok; bits += nb + ebits; };
        if (v) {
            emit(v, 0, 0); --run;
            while (run >= 3) { const uint32_t r = run < 6u ? run : 6u; emit(16, 2, r - 3u); run -= r; }
            while (run) { emit(v, 0, 0); --run; }
        } else {
            while (run >= 11) { const uint32_t r = run < 138u ? run : 138u; emit(18, 7, r - 11u); run -= r; }
            if (run >= 3) { emit(17, 3, run - 3u); run = 0; }
            while (run) { emit(0, 0, 0); --run; }
        }
        return (ntok - n0) | ((bits - b0) << 16);
    };
    {
        // first symbols: run starts below hlit
        const bool is0 = (uint32_t)tid < hlit && ((s_mask[wave] >> lane) & 1ull);
        const uint32_t v0 = len[tid];
        uint32_t r0 = 0, c0 = 0;
        if (is0) { r0 = run_end(s_mask, 5, (uint32_t)tid, hlit) - (uint32_t)tid; c0 = code_run(v0, r0, false, 0, 0); }
        uint32_t tot0;
        const uint32_t ex0 = block_excl_scan(c0, s_scan, &tot0);      // tokens in the low half, bits in the high half (<= 320 / <= 3 800)
        if (is0) (void)code_run(v0, r0, true, ex0 & 0xFFFFu, ex0 >> 16);
        if (wave == 0) {
            // symbols 256 .. hlit - 1, then the distance lengths: lanes 0 .. 29 of this wave, in order behind the first symbols
            const uint32_t i1 = 256u + (uint32_t)lane;
            const bool is1 = lane < GZ_NLIT - 256 && i1 < hlit && ((s_mask[4] >> lane) & 1ull);
            const uint32_t v1 = len[256 + (lane < GZ_NLIT - 256 ? lane : 0)];
            uint32_t r1 = 0, c1 = 0;
            if (is1) { r1 = run_end(s_mask, 5, i1, hlit) - i1; c1 = code_run(v1, r1, false, 0, 0); }
            const uint32_t in1 = wave_incl_scan(c1), tot1 = (uint32_t)__shfl((int)in1, 63);
            const uint32_t at1 = tot0 + in1 - c1;
            if (is1) (void)code_run(v1, r1, true, at1 & 0xFFFFu, at1 >> 16);
            const bool is2 = (uint32_t)lane < hdist && ((s_mask[5] >> lane) & 1ull);
            const uint32_t v2 = s_dsub[lane < 64 ? lane : 0];
            uint32_t r2 = 0, c2 = 0;
            if (is2) { r2 = run_end(s_mask + 5, 1, (uint32_t)lane, hdist) - (uint32_t)lane; c2 = code_run(v2, r2, false, 0, 0); }
            const uint32_t in2 = wave_incl_scan(c2), tot2 = (uint32_t)__shfl((int)in2, 63);
            const uint32_t at2 = tot0 + tot1 + in2 - c2;
            if (is2) (void)code_run(v2, r2, true, at2 & 0xFFFFu, at2 >> 16);
            if (lane == 0) { const uint32_t all = tot0 + tot1 + tot2; s_hdr[0] = all >> 16; s_hdr[1] = hlit - 257u; s_hdr[2] = hdist - 1u; s_hdr[3] = all & 0xFFFFu; }
        }
    }
    if (probe::off(1 << 27)) { __syncthreads(); return; }         // (analysis builds: ... + the block header's tokens)

    // ---- pass 2: bits of this lane's span; scan; this member's size and its byte offset by look-back over the chunks ----
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < GZ_SPAN / 4; ++k) {
        const uint32_t w = d[k], base = 4u * (uint32_t)k;
        const uint32_t cv = (cov[k >> 3] >> (4 * (k & 7))) & 15u;
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) { const uint32_t l = s_code[(w >> (8 * b)) & 255u] >> 16; bits += (base + b < slen && !((cv >> b) & 1u)) ? l : 0u; }
    }
    for (uint32_t q = 0; q < (uint32_t)GZ_MAXM; ++q) {           // a match: length code + extra bits + distance code + extra bits
        if (!__ballot(q < n_tok)) break;
        if (q < n_tok) {
            const uint32_t tk = s_tok[q][tid], ls = gz_len_sym((tk >> 7) & 127u), ds = gz_dist_sym(tk >> 14);
            const uint32_t L = (s_code[257u + (ls & 255u)] >> 16) + ((ls >> 8) & 255u) + (s_dcode[ds & 255u] >> 16) + ((ds >> 8) & 255u);
            s_tokp[q][tid] = (uint16_t)L;
            bits += L;
        }
    }
    uint32_t tot_bits;
    const uint32_t before = block_excl_scan(bits, s_scan, &tot_bits);      // (its barriers publish s_hdr)
    const uint32_t hdr_sym_bits = s_hdr[0];
    const uint32_t eob = s_code[256];
    const uint32_t body_bits = (uint32_t)GZ_FIXED_HDR_BITS + hdr_sym_bits + tot_bits + (eob >> 16) + 3u;      // ... + the final stored block's 3 header bits
    uint32_t body_bytes = ((body_bits + 7u) >> 3) + 4u;                                                      // ... + LEN, NLEN
    const bool coded = 10u + 4u + body_bytes + 8u <= (uint32_t)GZ_IMG_WORDS * 4u && body_bytes <= 5u + clen;
    if (!coded) body_bytes = 5u + clen;                                                                      // one stored block: header byte, LEN, NLEN, the text
    const uint32_t pad = (4u - ((10u + 1u + body_bytes + 8u) & 3u)) & 3u;                                    // FNAME = pad characters + NUL
    const uint32_t hdr_bytes = 10u + pad + 1u, member_bytes = hdr_bytes + body_bytes + 8u;
    if (wave == 0) { const uint64_t g = lookback_excl(a.status, t, member_bytes, 0); if (lane == 0) { s_base = g; if (c0 + clen >= n_text) *a.total = g + member_bytes; } }
    __syncthreads();
    if (s_base + member_bytes > a.cap) { if (tid == 0) atomicOr((unsigned long long *)a.flags, 8ull); return; }      // (cannot happen: gz_capacity covers stored members)

    if (probe::off(1 << 24)) return;      // ... + header tokens, bit counts, look-back
    // CRC-32 of the chunk: the spans' registers joined by a tree; the right half of a node is shifted in by its true length (the last chunk of
    // a stream is short), so one rule serves every node: left' = left * x^(8 * bytes of the right half) + right
    s_tab[tid] = crc;                                             // (the slicing tables are no longer needed: the barrier above)
    __syncthreads();
    for (int k = 0; k < 8; ++k) {
        uint32_t v = 0; const bool act = (tid & ((2 << k) - 1)) == 0;
        if (act) {
            const uint32_t r0 = ((uint32_t)tid + (1u << k)) * (uint32_t)GZ_SPAN, full = (uint32_t)GZ_SPAN << k;
            const uint32_t lr = r0 >= clen ? 0u : (clen - r0 < full ? clen - r0 : full);
            v = crc_append_zeros(a.crc_shift, s_tab[tid], lr) ^ s_tab[tid + (1 << k)];
        }
        __syncthreads();
        if (act) s_tab[tid] = v;
        __syncthreads();
    }
    const uint32_t chunk_crc = s_tab[0] ^ 0xFFFFFFFFu;
    const uint8_t hd10[10] = {0x1f, 0x8b, 8, 8, 0, 0, 0, 0, 0, 255};      // gzip header: magic, deflate, FLG = FNAME, mtime 0, xfl 0, OS 255, then the pad name

    if (!coded) {
        // ---- a stored member (text that does not compress to 3/4): header | 01 LEN NLEN text | CRC, length; written straight from the registers ----
        uint8_t *dst = reinterpret_cast<uint8_t *>(a.out) + s_base;
        if (tid == 0) {
            for (uint32_t q = 0; q < 10; ++q) dst[q] = hd10[q];
            for (uint32_t q = 0; q < pad; ++q) dst[10 + q] = 'x';
            dst[10 + pad] = 0;
            uint8_t *b = dst + hdr_bytes;
            b[0] = 1; b[1] = (uint8_t)(clen & 255u); b[2] = (uint8_t)(clen >> 8); b[3] = (uint8_t)(~clen & 255u); b[4] = (uint8_t)((~clen >> 8) & 255u);      // BFINAL = 1, BTYPE = 00
            uint8_t *e = b + 5 + clen;
            for (uint32_t q = 0; q < 4; ++q) { e[q] = (uint8_t)((chunk_crc >> (8 * q)) & 255u); e[4 + q] = (uint8_t)((clen >> (8 * q)) & 255u); }
        }
        uint8_t *body = dst + hdr_bytes + 5u + s0;
#pragma unroll
        for (int k = 0; k < GZ_SPAN / 4; ++k) {
            const uint32_t base = 4u * (uint32_t)k;
            if (base + 4 <= slen) { GzUnal4 v; v.v = d[k]; *reinterpret_cast<GzUnal4 *>(body + base) = v; }
            else for (uint32_t b = 0; base + b < slen; ++b) body[base + b] = (uint8_t)((d[k] >> (8 * b)) & 255u);
        }
        return;
    }

    // ---- pass 3: the member image ----
    const uint32_t data0 = hdr_bytes * 8u;                        // bit position of the DEFLATE data
    {
        // pass 3a: the literals, in one branch-free form per byte (a covered byte is a code of no bits, the first byte of a match a GAP of the match's
        // bits whose position in the lane's stream is noted); pass 3b, behind a barrier: the matches are ORed into their gaps.  Emitting a match where
        // it stands made every wave run the match path at every byte -- some lane always has one there.
        const uint32_t start_abs = data0 + (uint32_t)GZ_FIXED_HDR_BITS + hdr_sym_bits + before;
        LdsBits bs; bs.init(s_img, start_abs);
        uint32_t q = 0;
#pragma unroll
        for (int k = 0; k < GZ_SPAN / 4; ++k) {
            const uint32_t w = d[k], base = 4u * (uint32_t)k;
            const uint32_t cv = (cov[k >> 3] >> (4 * (k & 7))) & 15u, ms = (mst[k >> 3] >> (4 * (k & 7))) & 15u;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                const uint32_t c = s_code[(w >> (8 * b)) & 255u];
                const bool live = base + b < slen, isms = live && ((ms >> b) & 1u), iscv = (cv >> b) & 1u;
                uint32_t len = (live && !iscv) ? c >> 16 : 0u, code = (live && !iscv) ? c & 0xFFFFu : 0u;
                if (isms) {
                    len = s_tokp[q][tid];                                                    // the gap
                    s_tokp[q][tid] = (uint16_t)(((bs.w << 5) + bs.nb) - start_abs);          // ... and where it starts
                    ++q;
                    if (len > 32u) { bs.put(0, 16); len -= 16; }                             // (a match of more than 32 bits: next to never)
                }
                bs.put(code, len);
            }
        }
        bs.finish();
        __syncthreads();
        for (uint32_t t2 = 0; t2 < (uint32_t)GZ_MAXM; ++t2) {
            if (!__ballot(t2 < n_tok)) break;
            if (t2 < n_tok) {
                const uint32_t tk = s_tok[t2][tid], ls = gz_len_sym((tk >> 7) & 127u), ds = gz_dist_sym(tk >> 14);
                const uint32_t lc = s_code[257u + (ls & 255u)], dc = s_dcode[ds & 255u];
                uint64_t v = lc & 0xFFFFu; uint32_t nb = lc >> 16;
                v |= (uint64_t)(ls >> 16) << nb; nb += (ls >> 8) & 255u;
                v |= (uint64_t)(dc & 0xFFFFu) << nb; nb += dc >> 16;
                v |= (uint64_t)(ds >> 16) << nb;                                              // <= 15 + 5 + 15 + 13 = 48 bits
                const uint32_t pos = start_abs + (uint32_t)s_tokp[t2][tid], sh = pos & 31u, w0 = pos >> 5;
                const uint64_t lo = (v & 0xFFFFFFFFull) << sh, hi = (v >> 32) << sh;          // two halves: each fits 64 bits after the shift
                if ((uint32_t)lo) atomicOr(&s_img[w0], (uint32_t)lo);
                const uint32_t mid = (uint32_t)(lo >> 32) | (uint32_t)hi;
                if (mid) atomicOr(&s_img[w0 + 1], mid);
                if ((uint32_t)(hi >> 32)) atomicOr(&s_img[w0 + 2], (uint32_t)(hi >> 32));
            }
        }
    }
    if (tid == 0) {
        auto put_byte = [&](uint32_t pos, uint32_t b) { atomicOr(&s_img[pos >> 2], b << (8 * (pos & 3u))); };
        for (uint32_t q = 0; q < 10; ++q) put_byte(q, hd10[q]);
        for (uint32_t q = 0; q < pad; ++q) put_byte(10u + q, (uint32_t)'x');
        // block header: BFINAL = 0, BTYPE = 10, HLIT, HDIST, HCLEN = 15 (19 code-length code lengths: 4 for 0..12, 5 for 13..18), then the coded lengths
        LdsBits bh; bh.init(s_img, data0);
        bh.put(0, 1); bh.put(2, 2); bh.put(s_hdr[1], 5); bh.put(s_hdr[2], 5); bh.put(15, 4);
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int q = 0; q < 19; ++q) bh.put(order[q] <= 12 ? 4u : 5u, 3);
        bh.finish();                                              // (the coded lengths behind it: every lane ORs in its share of the tokens, below)
        // end of block + the final empty stored block (BFINAL = 1, BTYPE = 00, pad to a byte, LEN = 0, NLEN = 0xFFFF)
        LdsBits be; be.init(s_img, data0 + (uint32_t)GZ_FIXED_HDR_BITS + hdr_sym_bits + tot_bits);
        be.put(eob & 0xFFFFu, eob >> 16); be.put(1, 1); be.put(0, 2);
        be.finish();
        const uint32_t tail = hdr_bytes + ((body_bits + 7u) >> 3);
        put_byte(tail + 2, 0xFF); put_byte(tail + 3, 0xFF);
        for (uint32_t q = 0; q < 4; ++q) { put_byte(tail + 4 + q, (chunk_crc >> (8 * q)) & 255u); put_byte(tail + 8 + q, (clen >> (8 * q)) & 255u); }
    }
    for (uint32_t q = (uint32_t)tid; q < s_hdr[3]; q += GZ_THREADS) {      // the block header's tokens, each at its own bit offset
        const uint32_t tk = s_clt[q], pos = data0 + (uint32_t)GZ_FIXED_HDR_BITS + (tk >> 16), sh = pos & 31u;
        const uint64_t v = (uint64_t)(tk & 0xFFFu) << sh;
        atomicOr(&s_img[pos >> 5], (uint32_t)v);
        if ((uint32_t)(v >> 32)) atomicOr(&s_img[(pos >> 5) + 1], (uint32_t)(v >> 32));
    }
    __syncthreads();
    // ---- the image leaves: members are multiples of 4 bytes, so plain coalesced word stores ----
    uint32_t *dst = a.out + (s_base >> 2);
    for (uint32_t q = (uint32_t)tid; q < (member_bytes >> 2); q += GZ_THREADS) dst[q] = s_img[q];
}

uint64_t gz_chunks(uint64_t n) { return (n + GZ_CHUNK - 1) / GZ_CHUNK; }
// a member is never larger than its stored form: the text + 5 bytes of block header + 22 of framing and pad
uint64_t gz_capacity(uint64_t n) { return n + gz_chunks(n) * 32 + 64; }

// text_cap: capacity of the text buffer (the grid covers it; chunks past the real length, read from *n_dev on the device, leave at once)
void launch_gzip(hipStream_t st, const uint8_t *text, const uint64_t *n_dev, uint64_t text_cap, uint8_t *out, uint64_t out_cap, uint64_t *status, uint64_t *ticket, uint64_t *total, uint64_t *flags,
                 const uint32_t *crc_slice, const uint32_t *crc_shift)
{
    if (text_cap == 0) return;
    GzArgs a; a.text = text; a.n_dev = n_dev; a.out = reinterpret_cast<uint32_t *>(out); a.cap = out_cap; a.flags = flags; a.status = status; a.ticket = ticket; a.total = total; a.crc_slice = crc_slice; a.crc_shift = crc_shift;
    hipLaunchKernelGGL(k_gzip, dim3((uint32_t)gz_chunks(text_cap)), dim3(GZ_THREADS), 0, st, a);
}

// host: the slicing-by-4 tables of CRC-32 and the sixteen "append 2^m zero bytes" operators as 4 x 256 lookup tables each
void gz_host_tables(uint32_t *crc_slice, uint32_t *crc_shift)
{
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_slice[i] = c; }
    for (int j = 1; j < 4; ++j) for (uint32_t i = 0; i < 256; ++i) crc_slice[j * 256 + i] = (crc_slice[(j - 1) * 256 + i] >> 8) ^ crc_slice[crc_slice[(j - 1) * 256 + i] & 0xFFu];
    auto apply = [&](const uint32_t *t, uint32_t v) { return t[v & 255u] ^ t[256 + ((v >> 8) & 255u)] ^ t[512 + ((v >> 16) & 255u)] ^ t[768 + (v >> 24)]; };
    for (int j = 0; j < 4; ++j) for (uint32_t b = 0; b < 256; ++b) { const uint32_t v = b << (8 * j); crc_shift[j * 256 + b] = crc_slice[v & 0xFFu] ^ (v >> 8); }      // one zero byte
    for (int m = 1; m < 16; ++m) {
        const uint32_t *prev = crc_shift + (m - 1) * 1024; uint32_t *cur = crc_shift + m * 1024;
        for (int j = 0; j < 4; ++j) for (uint32_t b = 0; b < 256; ++b) cur[j * 256 + b] = apply(prev, apply(prev, b << (8 * j)));
    }
}

} // namespace dw
