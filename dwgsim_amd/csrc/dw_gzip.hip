// dw_gzip.hip -- gzip on the GPU for the packed FASTQ text (replaces the gzprintf / gzputc stream of the reference, src/dwgsim.c:919-981,
// files opened with gzopen at :1150-1158): the text never crosses PCIe uncompressed and the host only writes bytes.
//
// One workgroup = one 64 KiB chunk of a stream = one complete gzip member (RFC 1952), so the members of a batch are independent and their
// concatenation is a valid .gz whose decompressed bytes are exactly the text (what the reference's own test compares, testdata/test.sh:23-25):
//     10-byte header | one DEFLATE block with dynamic Huffman codes, literals only (RFC 1951 3.2.7) | an empty stored block (final, byte-aligns)
//     | CRC-32 and length of the chunk.
// FASTQ of simulated reads has no matches an LZ77 window could use beyond what entropy coding of the ~45 symbols gives (zlib -6 reaches the
// same ratio on it), so the codes are Huffman codes of the chunk's own byte histogram, length-limited to 15 bits.
//
//   pass 1  histogram (LDS atomics) + per-span CRC-32 (256-byte spans, table in LDS)
//   single  code lengths (Huffman, O(n^2) on <= 257 symbols, counts halved until <= 15 bits), canonical codes, block header bits
//   pass 2  bits per span -> block scan -> decoupled look-back over chunks for the member's byte offset
//   pass 3  every lane packs its span's codes LSB-first; words shared with a neighbouring span (or member) are OR-ed atomically into the
//           zeroed output, the others stored; CRC-32 of the spans combined by a tree of x^(8 L) shifts (tables from the host)
// Byte/integer work, HBM-bound by construction: the chunk is read three times from L2.
#include "dw_device.hpp"
#include "dw_launch.hpp"

namespace dw {

constexpr int GZ_CHUNK = 65536, GZ_THREADS = 256, GZ_SPAN = GZ_CHUNK / GZ_THREADS;      // 256 bytes per lane

struct GzArgs {
    const uint8_t *text; const uint64_t *n_dev;   // the stream; its length is still on the device when the kernel is enqueued
    uint8_t *out; uint64_t cap;             // zeroed output, cap bytes
    uint64_t *flags;                        // |= 8 when the output would not fit
    uint64_t *status;                       // look-back words, one per chunk (zeroed)
    uint64_t *ticket;                       // zeroed
    uint64_t *total;                        // out: compressed bytes of the stream (written by the last chunk)
    const uint32_t *crc_table;              // [256] byte-wise CRC-32 table (reflected 0xEDB88320)
    const uint32_t *crc_shift;              // [8][4][256]: multiply by x^(8 * 256 * 2^k) mod P, k = 0..7
};

DW_DEV uint32_t crc_apply_shift(const uint32_t *t, uint32_t v)
{
    return t[v & 255u] ^ t[256 + ((v >> 8) & 255u)] ^ t[512 + ((v >> 16) & 255u)] ^ t[768 + (v >> 24)];
}

struct BitSink {            // LSB-first bit packer into 32-bit words of a zeroed buffer; first and last word may be shared
    uint32_t *w; uint64_t acc; uint32_t nb; bool first;
    DW_DEV void init(uint8_t *out, uint64_t bitpos)      // out: the (4-byte aligned) output buffer, bitpos: absolute bit position in it
    {
        w = reinterpret_cast<uint32_t *>(out) + (bitpos >> 5);
        nb = (uint32_t)(bitpos & 31); acc = 0; first = true;
    }
    DW_DEV void put(uint32_t code, uint32_t len)
    {
        acc |= (uint64_t)code << nb; nb += len;
        if (nb >= 32) {
            if (first) { atomicOr(w, (uint32_t)acc); first = false; } else *w = (uint32_t)acc;
            ++w; acc >>= 32; nb -= 32;
        }
    }
    DW_DEV void finish() { if (nb) atomicOr(w, (uint32_t)acc); }
};

__global__ void __launch_bounds__(GZ_THREADS) k_gzip(GzArgs a)
{
    __shared__ uint32_t s_hist[260];
    __shared__ uint32_t s_code[257];          // bit-reversed canonical code | length << 16
    __shared__ uint32_t s_crc[256];
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_work[2 * 260];      // Huffman construction (thread 0)
    __shared__ uint32_t s_hdr[48]; __shared__ uint32_t s_hdr_bits;     // the block header as a bit string
    __shared__ uint32_t s_ticket; __shared__ uint64_t s_base;
    const int tid = (int)threadIdx.x;
    if (tid == 0) s_ticket = (uint32_t)atomicAdd((unsigned long long *)a.ticket, 1ull);
    for (int q = tid; q < 260; q += GZ_THREADS) s_hist[q] = 0;
    s_crc[tid] = a.crc_table[tid];
    __syncthreads();
    const uint32_t t = s_ticket;                                  // logical chunk: its predecessors have started
    const uint64_t n_text = *a.n_dev, c0 = (uint64_t)t * GZ_CHUNK;
    if (c0 >= n_text) return;                                     // (the grid is sized for the buffer's capacity)
    const uint32_t clen = (uint32_t)(n_text - c0 < (uint64_t)GZ_CHUNK ? n_text - c0 : (uint64_t)GZ_CHUNK);
    const uint8_t *src = a.text + c0;
    const uint32_t s0 = (uint32_t)tid * GZ_SPAN, slen = s0 >= clen ? 0u : (clen - s0 < (uint32_t)GZ_SPAN ? clen - s0 : (uint32_t)GZ_SPAN);

    // ---- pass 1: histogram + CRC-32 of this lane's span (register started from 0; the chunk's init value enters with lane 0) ----
    uint32_t crc = tid == 0 ? 0xFFFFFFFFu : 0u;
    for (uint32_t q = 0; q < slen; q += 16) {
        if (q + 16 <= slen) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + s0 + q);       // chunks and spans are 16-byte aligned in the text buffer
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; ++b) { const uint32_t c = (w[b >> 2] >> (8 * (b & 3))) & 0xFFu; atomicAdd(&s_hist[c], 1u); crc = s_crc[(crc ^ c) & 0xFFu] ^ (crc >> 8); }
        } else for (uint32_t b = q; b < slen; ++b) { const uint32_t c = src[s0 + b]; atomicAdd(&s_hist[c], 1u); crc = s_crc[(crc ^ c) & 0xFFu] ^ (crc >> 8); }
    }
    __syncthreads();

    // ---- code lengths, canonical codes and the block header: one lane (the alphabet of FASTQ text has ~45 symbols) ----
    if (tid == 0) {
        s_hist[256] = 1;                                          // end of block
        uint32_t *len = s_work, *cnt = s_work + 260;
        uint32_t scale = 0;
        for (;;) {
            // Huffman by repeated merging of the two lightest nodes; node weights in cnt[], leaf depths accumulated through parent links
            // kept compact: a leaf's length = number of merges its tree took part in
            int nsym = 0; uint16_t sym[257];
            for (int s = 0; s < 257; ++s) { len[s] = 0; if (s_hist[s]) { sym[nsym++] = (uint16_t)s; } }
            if (nsym == 1) { len[sym[0]] = 1; break; }            // (only the end-of-block symbol: an empty chunk never gets here, but keep the code complete-ish)
            // groups: each live tree is a list of leaves; represent by group id per leaf
            uint16_t grp[257]; uint32_t wgt[257]; int ngrp = nsym;
            for (int k = 0; k < nsym; ++k) { grp[k] = (uint16_t)k; const uint32_t h = s_hist[sym[k]]; wgt[k] = scale ? ((h >> scale) | 1u) : h; }
            bool alive[257]; for (int k = 0; k < nsym; ++k) alive[k] = true;
            while (ngrp > 1) {
                int m1 = -1, m2 = -1;
                for (int k = 0; k < nsym; ++k) if (alive[k]) { if (m1 < 0 || wgt[k] < wgt[m1]) { m2 = m1; m1 = k; } else if (m2 < 0 || wgt[k] < wgt[m2]) m2 = k; }
                for (int k = 0; k < nsym; ++k) if (grp[k] == m1 || grp[k] == m2) { ++len[sym[k]]; grp[k] = (uint16_t)m1; }
                wgt[m1] += wgt[m2]; alive[m2] = false; --ngrp;
            }
            uint32_t mx = 0; for (int k = 0; k < nsym; ++k) mx = len[sym[k]] > mx ? len[sym[k]] : mx;
            if (mx <= 15) break;
            ++scale;                                              // flatten the histogram and try again (terminates: equal weights give depth <= 9)
        }
        (void)cnt;
        // canonical codes (RFC 1951 3.2.2), stored bit-reversed for LSB-first packing
        uint32_t bl_count[16] = {0}, next_code[16] = {0};
        for (int s = 0; s < 257; ++s) ++bl_count[len[s]];
        bl_count[0] = 0;
        for (int bits = 1, code = 0; bits <= 15; ++bits) { code = (code + (int)bl_count[bits - 1]) << 1; next_code[bits] = (uint32_t)code; }
        for (int s = 0; s < 257; ++s) {
            const uint32_t l = len[s]; uint32_t rev = 0;
            if (l) { const uint32_t code = next_code[l]++; for (uint32_t b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b); }
            s_code[s] = rev | (l << 16);
        }
        // block header: BFINAL = 0, BTYPE = 10, HLIT = 0 (257 codes), HDIST = 0 (1 code), HCLEN = 15 (19 code-length code lengths);
        // the code-length code is fixed and complete: symbols 0..12 take 4 bits, 13..18 take 5 bits (13/16 + 6/32 = 1); every literal /
        // length code length is sent as itself (no run-length symbols), then the single distance code length 0 ("no distances")
        uint32_t nb = 0;
        for (int q = 0; q < 48; ++q) s_hdr[q] = 0;
        auto hput = [&](uint32_t v, uint32_t n) { for (uint32_t b = 0; b < n; ++b, ++nb) s_hdr[nb >> 5] |= ((v >> b) & 1u) << (nb & 31); };
        hput(0, 1); hput(2, 2); hput(0, 5); hput(0, 5); hput(15, 4);
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int q = 0; q < 19; ++q) hput(order[q] <= 12 ? 4u : 5u, 3);
        // canonical code of the code-length alphabet: lengths 4 for 0..12 -> codes 0..12; lengths 5 for 13..18 -> codes 26..31
        auto clput = [&](uint32_t sym) {
            const uint32_t l = sym <= 12 ? 4u : 5u, code = sym <= 12 ? sym : 26u + (sym - 13u);
            for (uint32_t b = 0; b < l; ++b, ++nb) s_hdr[nb >> 5] |= ((code >> (l - 1 - b)) & 1u) << (nb & 31);      // Huffman codes go MSB first
        };
        for (int s = 0; s < 257; ++s) clput(len[s]);
        clput(0);
        s_hdr_bits = nb;
    }
    __syncthreads();

    // ---- pass 2: bits of this lane's span; scan; this member's byte offset by look-back over the chunks ----
    uint32_t bits = 0;
    for (uint32_t q = 0; q < slen; q += 16) {
        if (q + 16 <= slen) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + s0 + q);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; ++b) bits += s_code[(w[b >> 2] >> (8 * (b & 3))) & 0xFFu] >> 16;
        } else for (uint32_t b = q; b < slen; ++b) bits += s_code[src[s0 + b]] >> 16;
    }
    uint32_t tot_bits;
    const uint32_t before = block_excl_scan(bits, s_scan, &tot_bits);
    const uint32_t hdr_bits = s_hdr_bits, eob = s_code[256];
    // member = 10 header bytes + ceil((block header + codes + end of block + 3 bits of the final stored block) / 8) + 4 (LEN, NLEN) + 8 (CRC, ISIZE)
    const uint32_t body_bits = hdr_bits + tot_bits + (eob >> 16) + 3;
    const uint32_t member_bytes = 10 + ((body_bits + 7) >> 3) + 4 + 8;
    if (tid < 64) { const uint64_t g = lookback_excl(a.status, t, member_bytes, 0); if (tid == 0) { s_base = g; if (c0 + clen >= n_text) *a.total = g + member_bytes; } }
    // CRC-32 of the chunk: tree of shifts over the 256 spans (full chunks); a short last chunk is summed by lane 0 alone
    __syncthreads();
    const uint64_t body0 = (s_base + 10) * 8;                     // absolute bit position of the member's DEFLATE data
    if (s_base + member_bytes > a.cap) { if (tid == 0) atomicOr((unsigned long long *)a.flags, 8ull); return; }      // (cannot happen for text: an 8-bit code is a valid prefix code)

    // ---- pass 3: pack ----
    {
        BitSink bs; bs.init(a.out, body0 + hdr_bits + before);
        for (uint32_t q = 0; q < slen; q += 16) {
            if (q + 16 <= slen) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + s0 + q);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int b = 0; b < 16; ++b) { const uint32_t c = s_code[(w[b >> 2] >> (8 * (b & 3))) & 0xFFu]; bs.put(c & 0xFFFFu, c >> 16); }
            } else for (uint32_t b = q; b < slen; ++b) { const uint32_t c = s_code[src[s0 + b]]; bs.put(c & 0xFFFFu, c >> 16); }
        }
        bs.finish();
    }
    // CRC combine
    s_crc[tid] = 0;       // (the table is no longer needed: the slots now carry span CRCs)
    __syncthreads();
    uint32_t chunk_crc = 0;
    if (clen == (uint32_t)GZ_CHUNK) {
        s_crc[tid] = crc;
        __syncthreads();
        for (int k = 0; k < 8; ++k) {              // level k joins neighbours of 256 * 2^k bytes each
            uint32_t v = 0; const bool act = (tid & ((2 << k) - 1)) == 0;
            if (act) v = crc_apply_shift(a.crc_shift + k * 1024, s_crc[tid]) ^ s_crc[tid + (1 << k)];
            __syncthreads();
            if (act) s_crc[tid] = v;
            __syncthreads();
        }
        chunk_crc = s_crc[0] ^ 0xFFFFFFFFu;
    } else if (tid == 0) {
        uint32_t r = 0xFFFFFFFFu;
        for (uint32_t b = 0; b < clen; ++b) r = a.crc_table[(r ^ src[b]) & 0xFFu] ^ (r >> 8);
        chunk_crc = r ^ 0xFFFFFFFFu;
    }
    if (tid == 0) {
        // gzip header: magic, deflate, no flags, mtime 0, xfl 0, OS 255 (bytes may share a word with the previous member's tail: OR them in)
        const uint8_t hd[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255};
        for (int q = 0; q < 10; ++q) { const uint64_t p = s_base + (uint64_t)q; atomicOr(reinterpret_cast<uint32_t *>(a.out) + (p >> 2), (uint32_t)hd[q] << (8 * (p & 3))); }
        // block header bits
        BitSink bs; bs.init(a.out, body0);
        for (uint32_t q = 0; q < hdr_bits; q += 16) { const uint32_t n = hdr_bits - q < 16 ? hdr_bits - q : 16; bs.put((s_hdr[q >> 5] >> (q & 31)) & ((1u << n) - 1u), n); }
        bs.finish();
        // end of block + final empty stored block (BFINAL = 1, BTYPE = 00, pad to a byte, LEN = 0, NLEN = 0xFFFF)
        BitSink be; be.init(a.out, body0 + hdr_bits + tot_bits);
        be.put(eob & 0xFFFFu, eob >> 16); be.put(1, 1); be.put(0, 2);
        be.finish();
        const uint64_t tail = s_base + 10 + ((body_bits + 7) >> 3);
        const uint8_t tl[12] = {0, 0, 0xFF, 0xFF, (uint8_t)chunk_crc, (uint8_t)(chunk_crc >> 8), (uint8_t)(chunk_crc >> 16), (uint8_t)(chunk_crc >> 24),
                                (uint8_t)clen, (uint8_t)(clen >> 8), (uint8_t)(clen >> 16), (uint8_t)(clen >> 24)};
        for (int q = 0; q < 12; ++q) { const uint64_t p = tail + (uint64_t)q; atomicOr(reinterpret_cast<uint32_t *>(a.out) + (p >> 2), (uint32_t)tl[q] << (8 * (p & 3))); }
    }
}

uint64_t gz_chunks(uint64_t n) { return (n + GZ_CHUNK - 1) / GZ_CHUNK; }
// An optimal prefix code never needs more than 8 bits per byte on average (the flat 8-bit code is a prefix code); + 1/8 for the rare
// length-limited case, + the per-member framing and code table
uint64_t gz_capacity(uint64_t n) { return n + n / 8 + gz_chunks(n) * 256 + 64; }

// text_cap: capacity of the text buffer (the grid covers it; chunks past the real length, read from *n_dev on the device, leave at once)
void launch_gzip(hipStream_t st, const uint8_t *text, const uint64_t *n_dev, uint64_t text_cap, uint8_t *out, uint64_t out_cap, uint64_t *status, uint64_t *ticket, uint64_t *total, uint64_t *flags,
                 const uint32_t *crc_table, const uint32_t *crc_shift)
{
    if (text_cap == 0) return;
    GzArgs a; a.text = text; a.n_dev = n_dev; a.out = out; a.cap = out_cap; a.flags = flags; a.status = status; a.ticket = ticket; a.total = total; a.crc_table = crc_table; a.crc_shift = crc_shift;
    hipLaunchKernelGGL(k_gzip, dim3((uint32_t)gz_chunks(text_cap)), dim3(GZ_THREADS), 0, st, a);
}

// host: the byte-wise CRC-32 table and the eight "append 256 * 2^k zero bytes" operators as 4 x 256 lookup tables each
void gz_host_tables(uint32_t *crc_table, uint32_t *crc_shift)
{
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[i] = c; }
    auto zeros = [&](uint32_t v, uint32_t nbytes) { for (uint32_t q = 0; q < nbytes; ++q) v = crc_table[v & 0xFFu] ^ (v >> 8); return v; };
    auto apply = [&](const uint32_t *t, uint32_t v) { return t[v & 255u] ^ t[256 + ((v >> 8) & 255u)] ^ t[512 + ((v >> 16) & 255u)] ^ t[768 + (v >> 24)]; };
    for (int j = 0; j < 4; ++j) for (uint32_t b = 0; b < 256; ++b) crc_shift[j * 256 + b] = zeros(b << (8 * j), GZ_SPAN);
    for (int k = 1; k < 8; ++k) {
        const uint32_t *prev = crc_shift + (k - 1) * 1024; uint32_t *cur = crc_shift + k * 1024;
        for (int j = 0; j < 4; ++j) for (uint32_t b = 0; b < 256; ++b) cur[j * 256 + b] = apply(prev, apply(prev, b << (8 * j)));
    }
}

} // namespace dw
