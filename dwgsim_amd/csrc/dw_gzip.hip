// dw_gzip.hip -- gzip on the GPU for the packed FASTQ text (replaces the gzprintf / gzputc stream of the reference, src/dwgsim.c:919-981,
// files opened with gzopen at :1150-1158): the text never crosses PCIe uncompressed and the host only writes bytes.
//
// One workgroup = one 32 KiB chunk of a stream = one complete gzip member (RFC 1952), so the members of a batch are independent and their
// concatenation is a valid .gz whose decompressed bytes are exactly the text (what the reference's own test compares, testdata/test.sh:23-25):
//     header with an FNAME field of 0-3 pad characters (so that EVERY member is a multiple of 4 bytes and starts on a word of the output)
//     | one DEFLATE block with dynamic Huffman codes, literals only (RFC 1951 3.2.7) | an empty stored block (final, byte-aligns)
//     | CRC-32 and length of the chunk.
// FASTQ of simulated reads has nothing an LZ77 window could match beyond what entropy coding of its ~45 symbols gives, so the codes are Huffman
// codes of the chunk's own byte histogram, length-limited to 15 bits.
//
// A lane owns 128 consecutive bytes and keeps them in registers (8 x dwordx4) for all three passes:
//   pass 1  histogram (LDS atomics into 8 sub-histograms: the four bases would otherwise collide 64-fold) + CRC-32 of the span (slicing-by-4)
//   codes   Huffman by repeated merging of the two lightest trees -- the two minima by one block-wide reduction per merge, every lane owning
//           one symbol -- counts halved until no code is longer than 15 bits; canonical codes; the block header (code lengths) by a scan
//   pass 2  bits per span -> block scan -> decoupled look-back over the chunks for the member's byte offset
//   pass 3  every lane packs its span's codes LSB-first into an IMAGE OF THE MEMBER IN LDS (OR for the words two spans share); CRC-32 of
//           the spans joined by a tree of x^(8 L) shifts; then the image leaves with plain, coalesced word stores: no global atomics, no
//           zeroed output buffer.
#include "dw_device.hpp"
#include "dw_launch.hpp"

namespace dw {

constexpr int GZ_CHUNK = 32768, GZ_THREADS = 256, GZ_SPAN = GZ_CHUNK / GZ_THREADS;      // 128 bytes per lane
constexpr int GZ_IMG_WORDS = (GZ_CHUNK + 4096) / 4;        // the member image: a prefix code never needs more than ~8 bits per byte + ~200 bytes of tables
constexpr int GZ_FIXED_HDR_BITS = 3 + 5 + 5 + 4 + 19 * 3;  // BFINAL, BTYPE, HLIT, HDIST, HCLEN, the 19 code-length code lengths

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
DW_DEV uint32_t bit_reverse(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32u - len); }     // len 1..15

struct LdsBits {            // LSB-first bit packer into the LDS image; the first and the last word of a run may be shared with a neighbour
    uint32_t *img; uint32_t w; uint64_t acc; uint32_t nb; bool first;
    DW_DEV void init(uint32_t *image, uint32_t bitpos) { img = image; w = bitpos >> 5; nb = bitpos & 31u; acc = 0; first = true; }
    DW_DEV void put(uint32_t code, uint32_t len)
    {
        acc |= (uint64_t)code << nb; nb += len;
        if (nb >= 32) {
            if (first) { atomicOr(&img[w], (uint32_t)acc); first = false; } else img[w] = (uint32_t)acc;
            ++w; acc >>= 32; nb -= 32;
        }
    }
    DW_DEV void finish() { if (nb) atomicOr(&img[w], (uint32_t)acc); }
};

__global__ void __launch_bounds__(GZ_THREADS) k_gzip(GzArgs a)
{
    __shared__ uint32_t s_img[GZ_IMG_WORDS];
    __shared__ uint32_t s_sub[8][260];        // sub-histograms (lane & 7), then: [0] weights, [1] tree ids, [2] code lengths of the construction
    __shared__ uint32_t s_code[257];          // bit-reversed canonical code | length << 16
    __shared__ uint32_t s_tab[1024];          // CRC slicing tables, later the span CRCs ([0..255])
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_min[2][4][2];       // the two lightest trees of every wave (double-buffered by merge parity)
    __shared__ uint32_t s_blc[16], s_next[16];
    __shared__ uint32_t s_ticket; __shared__ uint64_t s_base;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = (uint32_t)atomicAdd((unsigned long long *)a.ticket, 1ull);
    for (int q = tid; q < 8 * 260; q += GZ_THREADS) (&s_sub[0][0])[q] = 0;
    for (int q = tid; q < GZ_IMG_WORDS; q += GZ_THREADS) s_img[q] = 0;
    for (int q = tid; q < 1024; q += GZ_THREADS) s_tab[q] = a.crc_slice[q];
    if (tid < 16) s_blc[tid] = 0;
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

    // ---- pass 1: histogram + CRC-32 of the span (register started from 0; the chunk's init value enters with lane 0) ----
    uint32_t crc = tid == 0 ? 0xFFFFFFFFu : 0u;
    {
        uint32_t *hist = s_sub[tid & 7];
#pragma unroll
        for (int k = 0; k < GZ_SPAN / 4; ++k) {
            const uint32_t w = d[k], base = 4u * (uint32_t)k;
            if (base + 4 <= slen) {
                atomicAdd(&hist[w & 255u], 1u); atomicAdd(&hist[(w >> 8) & 255u], 1u); atomicAdd(&hist[(w >> 16) & 255u], 1u); atomicAdd(&hist[w >> 24], 1u);
                const uint32_t x = crc ^ w;
                crc = s_tab[768 + (x & 255u)] ^ s_tab[512 + ((x >> 8) & 255u)] ^ s_tab[256 + ((x >> 16) & 255u)] ^ s_tab[x >> 24];
            } else if (base < slen) {
                for (uint32_t b = 0; base + b < slen; ++b) { const uint32_t c = (w >> (8 * b)) & 255u; atomicAdd(&hist[c], 1u); crc = s_tab[(crc ^ c) & 255u] ^ (crc >> 8); }
            }
        }
    }
    __syncthreads();

    // ---- Huffman code lengths: lane k owns symbol k (lane 0 also the end-of-block symbol 256) ----
    uint32_t *wgt = s_sub[0], *grp = s_sub[1], *len = s_sub[2];
    uint32_t h0 = 0, h1 = 0;                                      // the counts of this lane's symbol(s)
    for (int q = 0; q < 8; ++q) h0 += s_sub[q][tid];
    if (tid == 0) h1 = 1;                                         // end of block: once
    __syncthreads();                                              // (the sub-histograms become work arrays)
    for (uint32_t scale = 0;; ++scale) {
        const uint32_t w0 = h0 ? (scale ? ((h0 >> scale) | 1u) : h0) : 0u, w1 = h1;
        wgt[tid] = w0; grp[tid] = (uint32_t)tid; len[tid] = 0;
        if (tid == 0) { wgt[256] = w1; grp[256] = 256; len[256] = 0; }
        uint32_t ntrees;
        { uint32_t tot; (void)block_excl_scan((w0 ? 1u : 0u) + (tid == 0 ? 1u : 0u), s_scan, &tot); ntrees = tot; }      // (its barriers publish the arrays)
        if (ntrees == 1) { if (tid == 0) len[256] = 1; __syncthreads(); break; }     // only the end-of-block symbol: cannot happen (clen >= 1), kept complete
        for (uint32_t merge = 0; ntrees > 1; ++merge, --ntrees) {
            // keys (weight << 9 | tree id), dead trees = all ones; (a, b) = the two smallest of this lane, then of the wave, then of the block
            uint32_t ka = wgt[tid] ? ((wgt[tid] << 9) | (uint32_t)tid) : 0xFFFFFFFFu, kb = 0xFFFFFFFFu;
            if (tid == 0 && wgt[256]) { const uint32_t k2 = (wgt[256] << 9) | 256u; if (k2 < ka) { kb = ka; ka = k2; } else kb = k2; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t oa = (uint32_t)__shfl_xor((int)ka, off), ob = (uint32_t)__shfl_xor((int)kb, off);
                const uint32_t lo = ka < oa ? ka : oa, hi = ka < oa ? oa : ka, ob2 = kb < ob ? kb : ob;
                ka = lo; kb = hi < ob2 ? hi : ob2;
            }
            if (lane == 0) { s_min[merge & 1][wave][0] = ka; s_min[merge & 1][wave][1] = kb; }
            __syncthreads();
            uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t oa = s_min[merge & 1][w][0], ob = s_min[merge & 1][w][1];
                const uint32_t lo = m1 < oa ? m1 : oa, hi = m1 < oa ? oa : m1, ob2 = m2 < ob ? m2 : ob;
                m1 = lo; m2 = hi < ob2 ? hi : ob2;
            }
            const uint32_t t1 = m1 & 511u, t2 = m2 & 511u;        // tree t2 joins tree t1: every leaf of both goes one level down
            if (grp[tid] == t1 || grp[tid] == t2) { if (h0) { ++len[tid]; grp[tid] = t1; } }
            if (tid == 0 && (grp[256] == t1 || grp[256] == t2)) { ++len[256]; grp[256] = t1; }
            __syncthreads();                                      // all leaves have read the weights' owners' ids before the weights change
            if ((uint32_t)tid == (t1 & 255u) && t1 < 256u) wgt[t1] = (m1 >> 9) + (m2 >> 9);
            if (tid == 0 && t1 == 256u) wgt[256] = (m1 >> 9) + (m2 >> 9);
            if ((uint32_t)tid == (t2 & 255u) && t2 < 256u) wgt[t2] = 0;
            if (tid == 0 && t2 == 256u) wgt[256] = 0;
            __syncthreads();
        }
        uint32_t mx = len[tid]; if (tid == 0 && len[256] > mx) mx = len[256];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, off); mx = o > mx ? o : mx; }
        if (lane == 0) s_min[0][wave][0] = mx;
        __syncthreads();
        mx = s_min[0][0][0]; for (int w = 1; w < 4; ++w) mx = s_min[0][w][0] > mx ? s_min[0][w][0] : mx;
        __syncthreads();
        if (mx <= 15) break;                                      // else flatten the histogram and build again (equal weights give depth <= 9)
    }

    // ---- canonical codes (RFC 1951 3.2.2), stored bit-reversed for LSB-first packing ----
    const uint32_t my_len = len[tid], eob_len = len[256];
    if (my_len) atomicAdd(&s_blc[my_len], 1u);
    if (tid == 0) atomicAdd(&s_blc[eob_len], 1u);
    __syncthreads();
    if (tid == 0) { uint32_t code = 0; s_next[0] = 0; for (int bits = 1; bits <= 15; ++bits) { code = (code + (bits > 1 ? s_blc[bits - 1] : 0u)) << 1; s_next[bits] = code; } }
    __syncthreads();
    {
        uint32_t rank = 0;                                        // symbols below mine with my length (the end-of-block symbol is above every byte)
        if (my_len) for (int j = 0; j < tid; ++j) rank += len[j] == my_len ? 1u : 0u;
        s_code[tid] = my_len ? (bit_reverse(s_next[my_len] + rank, my_len) | (my_len << 16)) : 0u;
        if (tid == 0) { uint32_t r = 0; for (int j = 0; j < 256; ++j) r += len[j] == eob_len ? 1u : 0u; s_code[256] = bit_reverse(s_next[eob_len] + r, eob_len) | (eob_len << 16); }
    }
    // the code-length code is fixed and complete: symbols 0..12 take 4 bits (codes 0..12), 13..18 take 5 bits (codes 26..31); every literal
    // code length is sent as itself, then the single distance code length 0.  Lane k sends symbol k's length (lane 255 also symbol 256's
    // and the distance code's).
    auto cl_code = [](uint32_t sym, uint32_t &nbits) -> uint32_t { nbits = sym <= 12u ? 4u : 5u; return bit_reverse(sym <= 12u ? sym : 26u + (sym - 13u), nbits); };
    uint32_t hb0, hb1 = 0, hb2 = 0, hc0, hc1 = 0, hc2 = 0;
    hc0 = cl_code(my_len, hb0);
    if (tid == 255) { hc1 = cl_code(eob_len, hb1); hc2 = cl_code(0u, hb2); }
    uint32_t hdr_sym_bits;
    const uint32_t hdr_before = block_excl_scan(hb0 + hb1 + hb2, s_scan, &hdr_sym_bits);      // (its barriers publish s_code)

    // ---- pass 2: bits of this lane's span; scan; this member's size and its byte offset by look-back over the chunks ----
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < GZ_SPAN / 4; ++k) {
        const uint32_t w = d[k], base = 4u * (uint32_t)k;
        if (base + 4 <= slen) bits += (s_code[w & 255u] >> 16) + (s_code[(w >> 8) & 255u] >> 16) + (s_code[(w >> 16) & 255u] >> 16) + (s_code[w >> 24] >> 16);
        else for (uint32_t b = 0; base + b < slen; ++b) bits += s_code[(w >> (8 * b)) & 255u] >> 16;
    }
    uint32_t tot_bits;
    const uint32_t before = block_excl_scan(bits, s_scan, &tot_bits);
    const uint32_t eob = s_code[256];
    const uint32_t body_bits = (uint32_t)GZ_FIXED_HDR_BITS + hdr_sym_bits + tot_bits + (eob >> 16) + 3u;      // ... + the final stored block's 3 header bits
    const uint32_t body_bytes = ((body_bits + 7u) >> 3) + 4u;                                                // ... + LEN, NLEN
    const uint32_t pad = (4u - ((10u + 1u + body_bytes + 8u) & 3u)) & 3u;                                    // FNAME = pad characters + NUL
    const uint32_t hdr_bytes = 10u + pad + 1u, member_bytes = hdr_bytes + body_bytes + 8u;
    const bool fits = member_bytes <= (uint32_t)GZ_IMG_WORDS * 4u;
    if (wave == 0) { const uint64_t g = lookback_excl(a.status, t, fits ? member_bytes : 0u, 0); if (lane == 0) { s_base = g; if (c0 + clen >= n_text) *a.total = g + (fits ? member_bytes : 0u); } }
    __syncthreads();
    if (!fits || s_base + member_bytes > a.cap) { if (tid == 0) atomicOr((unsigned long long *)a.flags, 8ull); return; }      // (cannot happen for text)

    // ---- pass 3: the member image ----
    const uint32_t data0 = hdr_bytes * 8u;                        // bit position of the DEFLATE data
    {
        LdsBits bs; bs.init(s_img, data0 + (uint32_t)GZ_FIXED_HDR_BITS + hdr_before);
        bs.put(hc0, hb0);
        if (tid == 255) { bs.put(hc1, hb1); bs.put(hc2, hb2); }
        bs.finish();
    }
    {
        LdsBits bs; bs.init(s_img, data0 + (uint32_t)GZ_FIXED_HDR_BITS + hdr_sym_bits + before);
#pragma unroll
        for (int k = 0; k < GZ_SPAN / 4; ++k) {
            const uint32_t w = d[k], base = 4u * (uint32_t)k;
            if (base + 4 <= slen) {
#pragma unroll
                for (int b = 0; b < 4; ++b) { const uint32_t c = s_code[(w >> (8 * b)) & 255u]; bs.put(c & 0xFFFFu, c >> 16); }
            } else for (uint32_t b = 0; base + b < slen; ++b) { const uint32_t c = s_code[(w >> (8 * b)) & 255u]; bs.put(c & 0xFFFFu, c >> 16); }
        }
        bs.finish();
    }
    // CRC-32 of the chunk: the spans' registers joined by a tree; the right half of a node is shifted in by its true length (the last chunk of
    // a stream is short), so one rule serves every node: left' = left * x^(8 * bytes of the right half) + right
    __syncthreads();                                              // (the slicing tables are no longer needed: the slots now carry span CRCs)
    s_tab[tid] = crc;
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
    if (tid == 0) {
        const uint32_t chunk_crc = s_tab[0] ^ 0xFFFFFFFFu;
        auto put_byte = [&](uint32_t pos, uint32_t b) { atomicOr(&s_img[pos >> 2], b << (8 * (pos & 3u))); };
        // gzip header: magic, deflate, FLG = FNAME, mtime 0, xfl 0, OS 255, the pad name
        const uint8_t hd[10] = {0x1f, 0x8b, 8, 8, 0, 0, 0, 0, 0, 255};
        for (uint32_t q = 0; q < 10; ++q) put_byte(q, hd[q]);
        for (uint32_t q = 0; q < pad; ++q) put_byte(10u + q, (uint32_t)'x');
        // block header: BFINAL = 0, BTYPE = 10, HLIT = 0 (257 codes), HDIST = 0 (1 code), HCLEN = 15 (19 code-length code lengths: 4 for 0..12, 5 for 13..18)
        LdsBits bh; bh.init(s_img, data0);
        bh.put(0, 1); bh.put(2, 2); bh.put(0, 5); bh.put(0, 5); bh.put(15, 4);
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int q = 0; q < 19; ++q) bh.put(order[q] <= 12 ? 4u : 5u, 3);
        bh.finish();
        // end of block + the final empty stored block (BFINAL = 1, BTYPE = 00, pad to a byte, LEN = 0, NLEN = 0xFFFF)
        LdsBits be; be.init(s_img, data0 + (uint32_t)GZ_FIXED_HDR_BITS + hdr_sym_bits + tot_bits);
        be.put(eob & 0xFFFFu, eob >> 16); be.put(1, 1); be.put(0, 2);
        be.finish();
        const uint32_t tail = hdr_bytes + ((body_bits + 7u) >> 3);
        put_byte(tail + 2, 0xFF); put_byte(tail + 3, 0xFF);
        for (uint32_t q = 0; q < 4; ++q) { put_byte(tail + 4 + q, (chunk_crc >> (8 * q)) & 255u); put_byte(tail + 8 + q, (clen >> (8 * q)) & 255u); }
    }
    __syncthreads();
    // ---- the image leaves: members are multiples of 4 bytes, so plain coalesced word stores ----
    uint32_t *dst = a.out + (s_base >> 2);
    for (uint32_t q = (uint32_t)tid; q < (member_bytes >> 2); q += GZ_THREADS) dst[q] = s_img[q];
}

uint64_t gz_chunks(uint64_t n) { return (n + GZ_CHUNK - 1) / GZ_CHUNK; }
// An optimal prefix code never needs more than 8 bits per byte on average (the flat 8-bit code is a prefix code); + 1/8 for the rare
// length-limited case, + the per-member framing and code table
uint64_t gz_capacity(uint64_t n) { return n + n / 8 + gz_chunks(n) * 256 + 64; }

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
