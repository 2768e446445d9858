/*
 * oracle/dwgsim_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A from-scratch, single-threaded, plain-C CPU restatement of the hot path of
 * nh13/DWGSIM (reference at /root/reference, v0.1.17-dev):
 *     src/mut.c     mutation walk, left-justification, mutations.txt / .vcf
 *     src/dwgsim.c  dwgsim_core(): contig scheduling + the per-read-pair loop
 *     src/dwgsim_opt.c  option surface, seeding, error-ramp slope
 * Every function below cites the reference file:line it restates.  Nothing here is
 * linked, imported or executed by the product (dwgsim_amd/, the C-ABI library, the
 * dwgsim-hip CLI).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker.
 *
 * Two RNG providers behind one call-site interface rng_u(domain,index,attempt,retry,slot):
 *   MODE A ("drand48"): the reference's sequential glibc drand48 stream
 *       X <- (0x5DEECE66D * X + 0xB) mod 2^48, value X * 2^-48, X0 = (seed << 16) | 0x330E
 *       (dwgsim_opt.c:385-394).  The keys are ignored; draws happen in exactly the
 *       order the reference makes them; the Box-Muller cache is global (dwgsim.c:158-159).
 *       PINNED: byte-identical to the unmodified reference (oracle/_ref/dwgsim) on the
 *       reference's own goldens (testdata/ex1.test.*) and on golden sets G1..G5
 *       (tests/golden/MANIFEST.json; tests/test_oracle_golden.py).
 *   MODE B ("philox"): Philox4x32-10 keyed by (seed, contig) with counter
 *       (index, retry, domain|attempt, block): every random decision has a fixed slot, so
 *       any read-index range is reproducible in any order.  Same code path as mode A; only
 *       the source of each uniform and the scope of the Box-Muller cache (per normal
 *       stream instead of global) differ.  This is what the HIP kernels must match
 *       bit-for-bit.
 *
 * The representation is this file's own: one byte per base per haplotype
 * (bits 0-3 base code, bits 4-5 mutation type, as the low byte of the reference's mut_t,
 * mut.h:25-30) plus a position-sorted insertion table per haplotype.  The reference's
 * short (<=26) / long insertion encodings (mut.c:282-377) carry the same sequence
 * P[0..n-1] (P[t] = draw[n-1-t]); see ins_* below.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <ctype.h>
#include <time.h>
#include <unistd.h>
#include <limits.h>
#include <stdarg.h>

/* ------------------------------------------------------------------------------------------
 * RNG providers
 * ---------------------------------------------------------------------------------------- */
enum { RNG_DRAND48 = 0, RNG_PHILOX = 1 };

/* mode-B domains (c2 = domain << 24 | attempt) */
enum {
    D_WALK = 1,        /* index = position; slots 0, 2..5 (see walk_contig); the "mutate this base?" test (mut.c:618) is a 16 + 16 bit draw of its own: */
    D_WALK_SITE = 7,   /* index = window q of 256 positions; NARROW words, word m = the m-th GAP between candidate sites inside the window (walk_site) */
    D_WALK_SITE_REF = 26,
    D_WALK_INSLEN = 2, /* index = position; slot k = k-th length-extension test */
    D_WALK_INSBASE = 3,/* index = position; slot k = k-th inserted-base draw */
    D_PAIR = 4,        /* index = ii; slot 0 rand-read test, 1 haplotype, 2 strand */
    D_PLACE = 5,       /* index = ii; slot t = position uniform of placement try t */
    D_PLACE_NORM = 6,  /* index = ii; block t = polar tries of placement try t */
    D_BASE0 = 8,       /* +j; index = ii; genomic reads: NARROW words, word m = the m-th GAP between error sites of the read end (base_error_sites); random
                          reads: 2-bit field i = base i (random_base) */
    D_QUAL0 = 10,      /* +j; index = ii; 16-BIT: the sequential stream of polar tries of the read's quality normals -- try t = the low (v1) and the high
                          (v2) half of word t & 3 of block t >> 2; every accepted try delivers two normals (v2*fac, then the cached v1*fac) */
    D_FLOW0 = 12,      /* +j; index = ii; generate_errors_flows: NARROW words, word m = the m-th GAP (quiet first draws in front of the m-th scoring
                          one) of pass 1 (+ D_FLOW_PASS2: of pass 2); every further draw of an event in its private stream + D_FLOW_EV (see flow_first) */
    D_CALIB = 14,      /* -B calibration (dwgsim_opt.c:415-457); index = read number */
    D_FLOW_PASS2 = 8,  /* added to D_FLOW0 / D_CALIB (+j) for the second pass of generate_errors_flows: domains 20-23 */
    D_SUB0 = 16,       /* +j; index = ii; NARROW: word m = substituted-base draw of error site m of the read end's chain (base_error_sites) */
    D_MUTIN = 18,      /* mutation-input files (-b): index = entry ordinal; slot 0 hom test, 1 het haplotype (mut.c:662-669) */
    D_MUTIN_BASE = 19, /* index = entry ordinal; slot j = random base j of the entry (mut.c:676, :314, :319, :354) */
    D_FLOW_REF = 32,   /* (rounds 2-5: the low halves of the per-event first uniforms; unused since the first draws are drawn as gaps) */
    D_FLOW_EV = 64,    /* added to a flow-model domain: the private stream of event h -- draw s = word s & 3 of the block (retry s >> 2, block h) */
    D_BASE_REF0 = 24   /* +j; index = ii; word m = the thinning draw of error site m on a ramp (base_error_sites) */
};

typedef struct {
    int mode;
    uint64_t x;            /* drand48 state (48 bits) */
    uint32_t k0, k1;       /* philox key: (uint32)seed, contig index */
    int iset; double gset; /* mode A: the reference's static Box-Muller cache (dwgsim.c:158-159) */
    int use_libm_log;      /* ran_normal uses libm log() instead of det_log() */
    uint64_t n_draws;      /* number of uniforms consumed (both modes) */
    uint64_t n_log_mismatch; /* mode diagnostics: det_log(x) != log(x) bitwise */
    FILE *dump;            /* --dump-draws (mode B): every uniform, as a raw double, in the order it is consumed -- what oracle/replay48.c feeds to the
                              UNMODIFIED reference as its drand48() stream (tests/test_replay_parity.py) */
} rng_t;
static inline double rng_out(rng_t *r, double u) { if (r->dump) fwrite(&u, sizeof u, 1, r->dump); return u; }

/* Philox4x32-10, Salmon et al. SC'11 (Random123).  SURVEY.md App. E.2 known answers are
 * checked in tests/test_oracle_units.py. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* the mode-B uniform: 53 bits from two words, exact in fp64 */
static inline double u53(uint32_t hi, uint32_t lo)
{
    return (double)(((uint64_t)hi << 21) | (uint64_t)(lo >> 11)) * 0x1p-53;
}

double oracle_philox_uniform(uint32_t seed, uint32_t contig, uint32_t domain, uint64_t index,
                             uint32_t attempt, uint32_t retry, uint32_t slot)
{
    uint32_t ctr[4], key[2], w[4];
    ctr[0] = (uint32_t)index;
    ctr[1] = (uint32_t)((index >> 32) & 0xFFFFu) | (retry << 16);
    ctr[2] = (domain << 24) | (attempt & 0xFFFFFFu);
    ctr[3] = slot >> 1;
    key[0] = seed; key[1] = contig;
    oracle_philox4x32_10(ctr, key, w);
    return (slot & 1) ? u53(w[2], w[3]) : u53(w[0], w[1]);
}

/* glibc drand48 */
double oracle_drand48_next(uint64_t *x)
{
    *x = (0x5DEECE66DULL * (*x) + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return (double)(*x) * 0x1p-48;
}

static void rng_seed(rng_t *r, int mode, int32_t seed)
{
    memset(r, 0, sizeof(*r));
    r->mode = mode;
    /* dwgsim_opt.c:388-393: xseed = {0x330e, seed & 0xffff, (seed >> 16) & 0xffff} */
    long sv = seed;
    r->x = ((uint64_t)((sv >> 16) & 0xffff) << 32) | ((uint64_t)(sv & 0xffff) << 16) | 0x330eULL;
    r->k0 = (uint32_t)seed;
    r->k1 = 0;
}

static inline double rng_u(rng_t *r, uint32_t dom, uint64_t idx, uint32_t att, uint32_t retry, uint32_t slot)
{
    r->n_draws++;
    if (r->mode == RNG_DRAND48) return oracle_drand48_next(&r->x);
    return rng_out(r, oracle_philox_uniform(r->k0, r->k1, dom, idx, att, retry, slot));
}

/* NARROW uniform of the per-base domains: mode B takes one 32-bit word (word = slot & 3 of block
 * slot >> 2) and returns w * 2^-32, so one Philox block serves four draws.  Mode A: next drand48. */
double oracle_philox_uniform32(uint32_t seed, uint32_t contig, uint32_t domain, uint64_t index,
                               uint32_t attempt, uint32_t retry, uint32_t slot)
{
    uint32_t ctr[4], key[2], w[4];
    ctr[0] = (uint32_t)index;
    ctr[1] = (uint32_t)((index >> 32) & 0xFFFFu) | (retry << 16);
    ctr[2] = (domain << 24) | (attempt & 0xFFFFFFu);
    ctr[3] = slot >> 2;
    key[0] = seed; key[1] = contig;
    oracle_philox4x32_10(ctr, key, w);
    return (double)w[slot & 3] * 0x1p-32;
}
static inline double rng_u32(rng_t *r, uint32_t dom, uint64_t idx, uint32_t att, uint32_t retry, uint32_t slot)
{
    r->n_draws++;
    if (r->mode == RNG_DRAND48) return oracle_drand48_next(&r->x);
    return rng_out(r, oracle_philox_uniform32(r->k0, r->k1, dom, idx, att, retry, slot));
}

/* (the per-base draws of a read end -- error test dwgsim.c:237, random-read base :1000 -- are defined behind the gap functions: base_error_sites, random_base) */
static inline uint32_t philox_halfword(rng_t *r, uint32_t dom, uint64_t idx, uint32_t att, uint32_t i)
{
    uint32_t ctr[4], key[2], w[4];
    ctr[0] = (uint32_t)idx; ctr[1] = (uint32_t)((idx >> 32) & 0xFFFFu); ctr[2] = (dom << 24) | (att & 0xFFFFFFu); ctr[3] = i >> 3;
    key[0] = r->k0; key[1] = r->k1;
    oracle_philox4x32_10(ctr, key, w);
    return (w[(i & 7) >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
}
/* a 16-bit uniform of a narrow stream: mode B takes halfword i of the stream (eight per Philox block, low half of a word first), u = h * 2^-16 */
static inline double rng_u16(rng_t *r, uint32_t dom, uint64_t idx, uint32_t att, uint32_t i)
{
    r->n_draws++;
    if (r->mode == RNG_DRAND48) return oracle_drand48_next(&r->x);
    return rng_out(r, (double)philox_halfword(r, dom, idx, att, i) * 0x1p-16);
}

/* ---- geometric gaps in integer arithmetic (round 6): shared by the site draws of the walk (walk_site below) and the first draws of the flow model
 * (flow_first) -- what a Bernoulli(e') sequence of "does this one score?" draws is in law: the quiet draws in front of each scoring one are Geometric(e'),
 * G = floor(-log2(U) / -log2(1 - e')), U = (2 w + 1) / 2^33, w one 32-bit word.  No libm: a 257-entry table of log2 made by repeated squaring, fixed
 * point, one 64 x 64 -> 128 multiplication by a reciprocal -- the kernels and the product's host code hold the same text (dw_kernels.hpp / dw_common.hpp). ---- */
static uint64_t ilog2_fixed(uint64_t y, int fb)      /* floor(log2(y) * 2^fb) for y >= 1, fb <= 56: repeated squaring in Q1.63 */
{
    int p = 63 - __builtin_clzll(y);
    uint64_t m = y << (63 - p), frac = 0;
    for (int k = 0; k < fb; ++k) {
        const unsigned __int128 sq = (unsigned __int128)m * m;       /* Q2.126 */
        if ((uint64_t)(sq >> 127)) { m = (uint64_t)(sq >> 64); frac = (frac << 1) | 1u; }
        else { m = (uint64_t)(sq >> 63); frac <<= 1; }
    }
    return ((uint64_t)p << fb) | frac;
}
static uint32_t flow_lg[257]; static int flow_lg_ready = 0;
typedef struct { uint64_t thr, R; int s; } gap_par_t;
static gap_par_t flow_gap_params(uint64_t thr)       /* -log2(1 - e') in Q8.56, normalised, and its reciprocal (dw_kernels.hpp flow_gap_params: the same text) */
{
    gap_par_t g; g.thr = thr; g.R = 0; g.s = 0;
    if (thr == 0 || thr >= 0x100000000ull) return g;
    if (!flow_lg_ready) { for (int i = 0; i < 256; ++i) flow_lg[i] = (uint32_t)ilog2_fixed(256u + (uint64_t)i, 32); flow_lg[256] = 0xFFFFFFFFu; flow_lg_ready = 1; }
    const uint64_t Lq = (32ull << 56) - ilog2_fixed(0x100000000ull - thr, 56);
    const int sh = __builtin_clzll(Lq);
    const unsigned __int128 q = ((unsigned __int128)1 << 127) / (Lq << sh);
    g.R = q >> 64 ? ~0ull : (uint64_t)q; g.s = 63 - sh;
    return g;
}
#define FLOW_NEVER 0xFFFFFFFFu
static uint32_t flow_gap(uint32_t w, const gap_par_t *g)      /* quiet first draws in front of the next scoring one */
{
    if (g->thr >= 0x100000000ull) return 0;
    const uint64_t X = ((uint64_t)w << 1) | 1u;
    const int p = 63 - __builtin_clzll(X);
    const uint64_t M = X << (63 - p);
    const uint32_t idx = (uint32_t)(M >> 55) & 0xFFu, r16 = (uint32_t)(M >> 39) & 0xFFFFu;
    const uint32_t f = flow_lg[idx] + (uint32_t)(((uint64_t)(flow_lg[idx + 1] - flow_lg[idx]) * r16) >> 16);
    const uint64_t Lu = ((uint64_t)(33 - p) << 56) - ((uint64_t)f << 24);
    const uint64_t G = (uint64_t)(((unsigned __int128)Lu * g->R) >> 64) >> g->s;
    return G > 0x3FFFFFFFull ? 0x3FFFFFFFu : (uint32_t)G;
}

/* mut.c:618 `c < 4 && drand48() < opt->mut_rate`, once per ACGT position outside a live deletion run: by far the most frequent draw of the walk -- and one
 * that says "no" 999 times in 1000.  Mode B (round 6): the candidate sites of a contig are a Bernoulli(r') process over its positions, r' = ceil(r 2^32) / 2^32;
 * restricted to a window of 256 positions it is independent of every other window, so each window [256 q, 256 q + 256) has a gap chain of its own: S_0 = G_0,
 * S_(m+1) = S_m + 1 + G_(m+1), gap m = word m & 3 of block m >> 2 of (D_WALK_SITE, index q); position 256 q + S is a candidate for every S < 256.  (Rounds 2-5:
 * one 16-bit uniform per position, a Philox block per eight positions -- 386 M blocks per 3.09 Gb genome, 2 ms of a 5.5 ms walk on every device that walks it.)
 * --dump-draws: 0 for a candidate, 1 - 2^-48 otherwise (the reference only compares the draw with the rate). */
static inline int walk_site(rng_t *r, uint32_t p, double mut_rate)
{
    static struct { int valid; uint32_t k0, k1, window, m, next; gap_par_t par; double rate; } st;
    r->n_draws++;
    if (r->mode == RNG_DRAND48) return oracle_drand48_next(&r->x) < mut_rate;
    const uint32_t q = p >> 8, in = p & 255u;
    if (!st.valid || st.k0 != r->k0 || st.k1 != r->k1 || st.window != q || st.rate != mut_rate) {
        st.valid = 1; st.k0 = r->k0; st.k1 = r->k1; st.window = q; st.rate = mut_rate; st.m = 0;
        st.par = flow_gap_params(!(mut_rate > 0) ? 0 : mut_rate >= 1.0 ? 0x100000000ull : (uint64_t)ceil(mut_rate * 4294967296.0));
        st.next = st.par.thr ? flow_gap((uint32_t)(oracle_philox_uniform32(r->k0, r->k1, D_WALK_SITE, q, 0, 0, st.m++) * 4294967296.0), &st.par) : FLOW_NEVER;
    }
    while (st.next < in) st.next += 1u + flow_gap((uint32_t)(oracle_philox_uniform32(r->k0, r->k1, D_WALK_SITE, q, 0, 0, st.m++) * 4294967296.0), &st.par);
    const int hit = st.next == in;
    rng_out(r, hit ? 0.0 : 1.0 - 0x1p-48);
    return hit;
}

/* ---- the per-base draws of read end j (round 6).
 * Sequencing errors (dwgsim.c:233-244): `drand48() < e.start + e.by * i`, once per base that is not N -- 98 % of them answer no.  Mode B draws the error SITES
 * of the read end as a gap chain over its positions 0 .. s-1 at the LARGEST rate of the ramp, thr_max = max_i ceil(e_i 2^32): gap m = word m & 3 of block
 * m >> 2 of (D_BASE0 + j, pair, attempt), sites S_0 = G_0, S_(m+1) = S_m + 1 + G_(m+1) while S_m < s.  Where the position's own threshold thr_i is below
 * thr_max (a ramp, -e 0.001-0.05) site m is THINNED: kept iff w2 * thr_max < thr_i * 2^32 with w2 = word m & 3 of block m >> 2 of (D_BASE_REF0 + j, ...) --
 * probability thr_i / thr_max, so that position i errs with probability thr_i / 2^32 as before, independently.  A site on an N base takes no error (the
 * reference makes no draw there): Bernoulli thinning again.  (Rounds 2-5: a 16 + 16 bit uniform per base, a Philox block per eight bases: 19 blocks per
 * 150-base read end where one or two do.)  --dump-draws: 0 for an error, 1 - 2^-48 otherwise.
 * Random reads (dwgsim.c:999-1001): base i = (int)(drand48() * 4) & 3 = the 2-bit field i of the same stream: bits 2 (i & 15) of word (i >> 4) & 3 of block
 * i >> 6 -- 64 bases per Philox block (rounds 2-5: eight); --dump-draws: (b + 0.5) / 4. ---- */
static uint64_t err_thr(double e) { return !(e > 0) ? 0 : e >= 1.0 ? 0x100000000ull : (uint64_t)ceil(e * 4294967296.0); }
static void base_error_sites(rng_t *r, int j, uint64_t idx, uint32_t att, int s, double e_start, double e_by, uint32_t *site)      /* site[0 .. s): m + 1 = this base errs unless it is N, as site m of the chain */
{
    memset(site, 0, sizeof(uint32_t) * (size_t)(s > 0 ? s : 0));
    if (r->mode == RNG_DRAND48 || s <= 0) return;
    uint64_t tmax = 0;
    for (int i = 0; i < s; ++i) { const uint64_t t = err_thr(e_start + e_by * i); if (t > tmax) tmax = t; }
    if (tmax == 0) return;
    const gap_par_t par = flow_gap_params(tmax);
    uint32_t m = 0;
    for (uint64_t S = flow_gap((uint32_t)(oracle_philox_uniform32(r->k0, r->k1, D_BASE0 + (uint32_t)j, idx, att, 0, 0) * 4294967296.0), &par); S < (uint64_t)s; ) {
        const uint64_t ti = err_thr(e_start + e_by * (double)(int)S);
        int keep = 1;
        if (ti < tmax) { const uint64_t w2 = (uint64_t)(oracle_philox_uniform32(r->k0, r->k1, D_BASE_REF0 + (uint32_t)j, idx, att, 0, m) * 4294967296.0); keep = w2 * tmax < (ti << 32); }
        if (keep) site[S] = m + 1u;
        ++m;
        S += 1u + flow_gap((uint32_t)(oracle_philox_uniform32(r->k0, r->k1, D_BASE0 + (uint32_t)j, idx, att, 0, m) * 4294967296.0), &par);
    }
}
/* the error test of a base that is not N, in the order the reference makes them */
static inline int base_errs(rng_t *r, const uint32_t *site, int i, double e_i)
{
    r->n_draws++;
    if (r->mode == RNG_DRAND48) return oracle_drand48_next(&r->x) < e_i;
    rng_out(r, site[i] ? 0.0 : 1.0 - 0x1p-48);
    return site[i] != 0;
}
static inline uint8_t random_base(rng_t *r, int j, uint64_t idx, uint32_t att, uint32_t i)
{
    r->n_draws++;
    if (r->mode == RNG_DRAND48) return (uint8_t)((int)(oracle_drand48_next(&r->x) * 4.0) & 3);
    uint32_t ctr[4], key[2], w[4];
    ctr[0] = (uint32_t)idx; ctr[1] = (uint32_t)((idx >> 32) & 0xFFFFu); ctr[2] = ((D_BASE0 + (uint32_t)j) << 24) | (att & 0xFFFFFFu); ctr[3] = i >> 6;
    key[0] = r->k0; key[1] = r->k1;
    oracle_philox4x32_10(ctr, key, w);
    const uint8_t b = (uint8_t)((w[(i >> 4) & 3] >> (2 * (i & 15))) & 3u);
    rng_out(r, ((double)b + 0.5) * 0.25);
    return b;
}

/* Deterministic natural log for x > 0 finite: the classic fdlibm/FreeBSD-msun e_log.c
 * algorithm (argument reduction x = 2^k (1+f), s = f/(2+f), degree-14 even polynomial in s)
 * restated with only IEEE + - * / in fp64, so gcc/x86-64 and hipcc/gfx950 (both compiled
 * with -ffp-contract=off) give identical bits.  The algorithm and its coefficients are from
 * "e_log.c (c) 1993 Sun Microsystems, Inc. -- Permission to use, copy, modify, and distribute
 * this software is freely granted, provided that this notice is preserved."  < 1 ulp. */
double oracle_det_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    uint64_t b; int32_t hx, k = 0, i, j;
    memcpy(&b, &x, 8);
    hx = (int32_t)(b >> 32);
    if (hx < 0x00100000) { /* subnormal: scale up */
        x *= 0x1p54; k -= 54;
        memcpy(&b, &x, 8); hx = (int32_t)(b >> 32);
    }
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    b = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (b & 0xFFFFFFFFu);
    memcpy(&x, &b, 8);              /* x in [sqrt(2)/2, sqrt(2)) */
    k += (i >> 20);
    double f = x - 1.0, dk = (double)k;
    double s = f / (2.0 + f);
    double z = s * s, w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    i = hx - 0x6147a; j = 0x6b851 - hx; i |= j;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* A stream of normals.  Mode A ignores it (global cache + sequential draws).  Mode B, wide (placement): polar
 * tries r = 0,1,.. of block p; the accepted try gives normal 2p (= v2*fac) and, if `cache`,
 * normal 2p+1 (= v1*fac); p advances after every accepted try.  Mode B, narrow (quality strings): the tries form ONE sequential
 * stream t = 0,1,2,.. (field p counts tries) of 16-bit uniforms: try t = halfwords 2t, 2t + 1 of the stream = the low and the high half of
 * word t & 3 of block t >> 2 -- the reference's own consumption pattern (a rejected try costs two uniforms, an accepted one delivers two
 * normals) with Philox halfwords as the uniforms: 2^32 distinct tries, far more than the truncation (int)(nrm * sigma + 0.5) can resolve. */
typedef struct { uint32_t dom; uint64_t idx; uint32_t att; uint32_t p; int cache; int has; double g; int narrow; } nstream_t;

/* dwgsim.c:156-175 ran_normal(): Marsaglia polar method */
static double ran_normal(rng_t *r, nstream_t *ns)
{
    int *iset = (r->mode == RNG_DRAND48) ? &r->iset : &ns->has;
    double *gset = (r->mode == RNG_DRAND48) ? &r->gset : &ns->g;
    if (*iset == 0) {
        double v1, v2, rsq, fac, lg;
        uint32_t retry = 0;
        do {
            if (ns->narrow) {
                v1 = 2.0 * rng_u16(r, ns->dom, ns->idx, ns->att, 2 * ns->p) - 1.0;
                v2 = 2.0 * rng_u16(r, ns->dom, ns->idx, ns->att, 2 * ns->p + 1) - 1.0;
                ns->p++;
            } else {
                v1 = 2.0 * rng_u(r, ns->dom, ns->idx, ns->att, retry, 2 * ns->p) - 1.0;
                v2 = 2.0 * rng_u(r, ns->dom, ns->idx, ns->att, retry, 2 * ns->p + 1) - 1.0;
            }
            rsq = v1 * v1 + v2 * v2;
            retry++;
        } while (rsq >= 1.0 || rsq == 0.0);
        lg = oracle_det_log(rsq);
        { double l2 = log(rsq); if (memcmp(&l2, &lg, 8) != 0) r->n_log_mismatch++; if (r->use_libm_log) lg = l2; }
        fac = sqrt(-2.0 * lg / rsq);
        *gset = v1 * fac;
        *iset = (r->mode == RNG_DRAND48) ? 1 : ns->cache;
        if (!ns->narrow) ns->p++;
        return v2 * fac;
    }
    *iset = 0;
    return *gset;
}

/* ------------------------------------------------------------------------------------------
 * Options (dwgsim_opt.h:21-60, dwgsim_opt.c)
 * ---------------------------------------------------------------------------------------- */
enum { ILLUMINA = 0, SOLID = 1, IONTORRENT = 2 };
typedef struct { double start, by, end; } erate_t;
typedef struct {
    erate_t e[2];
    int is_inner, dist; double std_dev;
    int64_t N; double C;
    int length[2];
    double mut_rate, mut_freq, indel_frac, indel_extend; int indel_min;
    double rand_read; int max_n, data_type, strandedness, read_one_strand;
    int8_t *flow_order; int flow_order_len, use_base_error, is_hap;
    int32_t seed;
    char *fixed_quality; double quality_std;
    char *read_prefix; int reads_output_type, output_type, amplicons;
    /* oracle-only switches */
    char *fn_regions;               /* -x */
    char *fn_muts; int muts_type;   /* -m (1, txt) / -b (0, bed) / -v (2, vcf): mut_input.h:29-33 */
    int rng_mode, use_libm_log, null_fastq, verbose;
    const char *dump_path;            /* --dump-draws FILE (mode B only) */
    int64_t emit_first, emit_count;   /* --emit-range first:count (mode B only): emit only these read indices of every contig */
    /* Whole-genome checks without walking the whole genome (mode B only: every decision has its own RNG slot, so nothing else carries over):
     * --as-contig K,TOT,AFTER,NSIM  the FASTA holds ONE contig of a larger genome: it is contig number K (RNG key), the genome's total length is
     *                               TOT, AFTER contigs follow it, NSIM pairs were simulated before it (dwgsim.c:519-537, :582-590)
     * --range-rand-base R           with --emit-range: start the pair loop AT the window, with R random reads emitted before it (dwgsim.c:1042,1096) */
    int64_t as_contig, as_tot, as_after, as_nsim, range_rand_base;
} opt_t;

static void opt_defaults(opt_t *o) /* dwgsim_opt.c:40-80 */
{
    memset(o, 0, sizeof(*o));
    o->e[0].start = o->e[0].end = o->e[1].start = o->e[1].end = 0.02;
    o->dist = 500; o->std_dev = 50; o->N = -1; o->C = 100;
    o->length[0] = o->length[1] = 70;
    o->mut_rate = 0.001; o->mut_freq = 0.5; o->indel_frac = 0.1; o->indel_extend = 0.3; o->indel_min = 1;
    o->rand_read = 0.05; o->seed = -1; o->quality_std = 2.0; o->muts_type = -1;
}

static uint8_t nt4(int ch) /* dwgsim.c:56-73 nst_nt4_table */
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-': return 5;
    default: return 4;
    }
}

/* ------------------------------------------------------------------------------------------
 * FASTA (mut.c:49-87 seq_read_fasta): name = first token after '>', sequence keeps
 * isalpha / '-' / '.'.
 * ---------------------------------------------------------------------------------------- */
typedef struct { int64_t l, m; uint8_t *s; } seq_t;

static int64_t fasta_next(FILE *fp, seq_t *seq, char *name)
{
    int c = 0; char *p = name;
    while (!feof(fp) && fgetc(fp) != '>');
    if (feof(fp)) return -1;
    while (!feof(fp) && (c = fgetc(fp)) != ' ' && c != '\t' && c != '\n')
        if (c != '\r') *p++ = (char)c;
    *p = 0;
    if (c != '\n') while (!feof(fp) && fgetc(fp) != '\n');
    int64_t l = 0;
    while (!feof(fp) && (c = fgetc(fp)) != '>') {
        if (isalpha(c) || c == '-' || c == '.') {
            if (l + 1 >= seq->m) { seq->m = seq->m ? seq->m * 2 : 1 << 16; seq->s = realloc(seq->s, (size_t)seq->m); }
            seq->s[l++] = (uint8_t)c;
        }
    }
    if (c == '>') ungetc(c, fp);
    if (!seq->s) { seq->m = 16; seq->s = malloc(16); }
    seq->s[l] = 0; seq->l = l;
    return l;
}

/* ------------------------------------------------------------------------------------------
 * Mutated haplotypes
 * ---------------------------------------------------------------------------------------- */
#define T_NONE 0x00
#define T_INS  0x10
#define T_SUB  0x20
#define T_DEL  0x30
#define TMASK  0x30
#define BT_MASK 0x3f   /* mut_and_type_mask, mut.c:103 */

typedef struct { int64_t pos; uint32_t n; uint8_t *b; } ins_t;   /* b[t] = P[t], printed order */
typedef struct { int64_t l; uint8_t *c; ins_t *ins; int n_ins, m_ins; } hap_t;

static void hap_free(hap_t *h) { for (int i = 0; i < h->n_ins; ++i) free(h->ins[i].b); free(h->ins); free(h->c); memset(h, 0, sizeof(*h)); }

static ins_t *ins_find(hap_t *h, int64_t pos)
{
    int lo = 0, hi = h->n_ins - 1;
    while (lo <= hi) { int mid = (lo + hi) >> 1; if (h->ins[mid].pos == pos) return &h->ins[mid]; if (h->ins[mid].pos < pos) lo = mid + 1; else hi = mid - 1; }
    fprintf(stderr, "oracle: insertion lookup failed at %lld\n", (long long)pos); exit(2);
}

static void ins_push(hap_t *h, int64_t pos, uint32_t n, const uint8_t *P)
{
    if (h->n_ins == h->m_ins) { h->m_ins = h->m_ins ? h->m_ins * 2 : 16; h->ins = realloc(h->ins, sizeof(ins_t) * (size_t)h->m_ins); }
    ins_t *e = &h->ins[h->n_ins++];
    e->pos = pos; e->n = n; e->b = malloc(n ? n : 1); memcpy(e->b, P, n);
}

/* mut.c:282-377 mut_add_ins(), random-bases branch (bases == NULL, num_ins == 0, hap < 0) */
static void walk_add_ins(const opt_t *o, rng_t *r, hap_t *h0, hap_t *h1, int64_t i, uint8_t c)
{
    uint64_t num = 0; uint32_t k = 0; int hap;
    do { num++; } while (num < UINT32_MAX && (num < (uint64_t)o->indel_min || rng_u(r, D_WALK_INSLEN, (uint64_t)i, 0, 0, k++) < o->indel_extend));
    if (o->is_hap || rng_u(r, D_WALK, (uint64_t)i, 0, 0, 4) < 0.333333) hap = 3;
    else if (rng_u(r, D_WALK, (uint64_t)i, 0, 0, 5) < 0.5) hap = 1;
    else hap = 2;
    uint8_t *P = malloc((size_t)num);
    for (uint64_t j = 0; j < num; ++j) /* draw j lands at printed index num-1-j (mut.c:313-315 / :347-365 with :249-279) */
        P[num - 1 - j] = (uint8_t)(uint64_t)(rng_u(r, D_WALK_INSBASE, (uint64_t)i, 0, 0, (uint32_t)j) * 4.0);
    if (hap & 1) { h0->c[i] = T_INS | c; ins_push(h0, i, (uint32_t)num, P); }
    if (hap & 2) { h1->c[i] = T_INS | c; ins_push(h1, i, (uint32_t)num, P); }
    free(P);
}

/* mut.c:606-643 mut_diref(), random branch.  Mode-B slots at position i (domain D_WALK):
 * 0 deletion-extend test, 1 mutate test, 2 substitution-vs-indel, 3 substituted base / deletion-vs-insertion,
 * 4 hom test, 5 het haplotype. */
static void walk_contig(const opt_t *o, rng_t *r, const seq_t *seq, hap_t *h0, hap_t *h1)
{
    int deleting = 0; int64_t dlen = 0;
    hap_t *h[2] = { h0, h1 };
    for (int x = 0; x < 2; ++x) { h[x]->l = seq->l; h[x]->c = calloc((size_t)seq->l + 1, 1); h[x]->ins = NULL; h[x]->n_ins = h[x]->m_ins = 0; }
    for (int64_t i = 0; i < seq->l; ++i) {
        uint8_t c = nt4(seq->s[i]);
        h0->c[i] = h1->c[i] = c;
        if (deleting) {
            if (dlen < o->indel_min || rng_u(r, D_WALK, (uint64_t)i, 0, 0, 0) < o->indel_extend) {
                if (deleting & 1) h0->c[i] |= T_DEL | c;
                if (deleting & 2) h1->c[i] |= T_DEL | c;
                dlen++;
                continue;
            }
            deleting = 0; dlen = 0;
        }
        if (c < 4 && walk_site(r, (uint32_t)i, o->mut_rate)) {
            if (rng_u(r, D_WALK, (uint64_t)i, 0, 0, 2) >= o->indel_frac) { /* substitution */
                double rr = rng_u(r, D_WALK, (uint64_t)i, 0, 0, 3);
                uint8_t c2 = (uint8_t)((c + (uint64_t)(rr * 3.0 + 1)) & 3);
                if (o->is_hap || rng_u(r, D_WALK, (uint64_t)i, 0, 0, 4) < 0.333333) h0->c[i] = h1->c[i] = T_SUB | c2;
                else h[rng_u(r, D_WALK, (uint64_t)i, 0, 0, 5) < 0.5 ? 0 : 1]->c[i] = T_SUB | c2;
            } else if (rng_u(r, D_WALK, (uint64_t)i, 0, 0, 3) < 0.5) { /* deletion */
                if (o->is_hap || rng_u(r, D_WALK, (uint64_t)i, 0, 0, 4) < 0.3333333) { h0->c[i] = h1->c[i] = T_DEL | c; deleting = 3; }
                else { deleting = rng_u(r, D_WALK, (uint64_t)i, 0, 0, 5) < 0.5 ? 1 : 2; h[deleting - 1]->c[i] = T_DEL | c; }
                dlen = 1;
            } else walk_add_ins(o, r, h0, h1, i, c);
        }
    }
}

/* mut.c:379-425 mut_debug(): the consistency asserts the reference runs over both haplotypes before AND after the left-justification
 * (mut.c:753, :757).  They are live in the reference build (no -DNDEBUG): a mutation input that makes one fail -- e.g. `-m` with a
 * homozygous SNP whose alt base equals the reference base -- ends the reference with SIGABRT (rc 134) before anything is written.
 * Restated on the byte cells; the vacuous asserts (n1 == n2 with both read from hap1, n <= UINT32_MAX, ...) are omitted.  Returns the
 * text of the first failing assert (NULL if none) and its 0-based position. */
static const char *mut_debug(const seq_t *seq, hap_t *h0, hap_t *h1, int64_t *at)
{
    for (int64_t i = 0; i < seq->l; ++i) {
        const uint8_t c0 = nt4(seq->s[i]), c1 = h0->c[i], c2 = h1->c[i];
        if (c0 >= 4) continue;
        if ((c1 & TMASK) == T_NONE && (c2 & TMASK) == T_NONE) continue;
        *at = i;
        if ((c1 & BT_MASK) == (c2 & BT_MASK)) {                       /* hom, mut.c:391-405 */
            if ((c1 & TMASK) == T_SUB) { if ((c0 & 3) == (c1 & 3)) return "(c[0]&0x3) != (c[1]&0x3)"; }
            else if ((c1 & TMASK) == T_INS) {
                uint32_t n = 0; for (int q = 0; q < h0->n_ins; ++q) if (h0->ins[q].pos == i) n = h0->ins[q].n;
                if (n == 0) return "n1 > 0";
            }
        } else {                                                      /* het, mut.c:406-422 */
            if ((c1 & TMASK) == T_SUB || (c2 & TMASK) == T_SUB) {
                if ((c1 & 3) == (c2 & 3)) return "(c[1]&0x3) != (c[2]&0x3)";
                if (!((c0 & 3) == (c1 & 3) || (c0 & 3) == (c2 & 3))) return "(c[0]&0x3) == (c[1]&0x3) || (c[0]&0x3) == (c[2]&0x3)";
            } else if ((c1 & TMASK) == T_DEL || (c2 & TMASK) == T_DEL) { }
            else if ((c1 & TMASK) == T_INS) {
                uint32_t n = 0; for (int q = 0; q < h0->n_ins; ++q) if (h0->ins[q].pos == i) n = h0->ins[q].n;
                if (n == 0) return "n > 0";
            } else if ((c2 & TMASK) == T_INS) {
                uint32_t n = 0; for (int q = 0; q < h1->n_ins; ++q) if (h1->ins[q].pos == i) n = h1->ins[q].n;
                if (n == 0) return "n > 0";
            }
        }
    }
    return NULL;
}
static void mut_debug_or_abort(const char *name, const seq_t *seq, hap_t *h0, hap_t *h1)
{
    int64_t at = 0;
    const char *what = mut_debug(seq, h0, h1, &at);
    if (!what) return;
    fprintf(stderr, "dwgsim: src/mut.c: mut_debug: Assertion `%s' failed. [%s:%lld]\n", what, name, (long long)at + 1);
    fflush(NULL);
    abort();                                                          /* as assert() does: SIGABRT, shell rc 134 */
}

/* ------------------------------------------------------------------------------------------
 * Mutation-input files: mut_txt.c:40-133 (-m), mut_bed.c:37-137 (-b), mut_vcf.c:42-280 (-v) and their
 * application in mut_diref, mut.c:644-745.  Entries keep file order; `contig` is the FASTA ordinal.
 * ---------------------------------------------------------------------------------------- */
typedef struct { char **name; int64_t *len; int n; } ctab_t;
typedef struct { uint32_t contig, pos; uint8_t type, is_hap; char *bases; } min_txt_t;      /* mut_txt.h:23-29 */
typedef struct { uint32_t contig, start, end; uint8_t type; char *bases; } min_bed_t;        /* mut_bed.h */
typedef struct { int type; min_txt_t *t; int nt; min_bed_t *b; int nb; } mutin_t;

static char iupac_and_base_to_mut(char iupac, char base) /* dwgsim.c:202-213 */
{
    static const char *codes = "XACMGRSVTWYHKDBN";
    int b = nt4(base);
    for (int i = 0; i < 4; ++i) if (codes[1 << (b & 3) | 1 << (i & 3)] == iupac) return "ACGTN"[i];
    return 'X';
}
static int get_muttype(char *str) /* dwgsim.c:183-200 */
{
    for (char *q = str; *q; ++q) *q = (char)tolower((unsigned char)*q);
    if (!strcmp("snp", str) || !strcmp("substitute", str) || !strcmp("sub", str) || !strcmp("s", str)) return T_SUB;
    if (!strcmp("insertion", str) || !strcmp("insert", str) || !strcmp("ins", str) || !strcmp("i", str)) return T_INS;
    if (!strcmp("deletion", str) || !strcmp("delet", str) || !strcmp("del", str) || !strcmp("d", str)) return T_DEL;
    return -1;
}
static void mi_push_txt(mutin_t *m, uint32_t contig, uint32_t pos, int type, uint32_t is_hap, const char *bases)
{
    m->t = realloc(m->t, sizeof(min_txt_t) * (size_t)(m->nt + 1));
    min_txt_t *e = &m->t[m->nt++];
    e->contig = contig; e->pos = pos; e->type = (uint8_t)type; e->is_hap = (uint8_t)is_hap; e->bases = bases ? strdup(bases) : NULL;
}
static void parse_txt(FILE *fp, const ctab_t *c, mutin_t *m) /* mut_txt.c:40-133 */
{
    char name[1024], mut[1024], ref; uint32_t pos, prev_pos = 0, is_hap; int i = 0;
    (void)prev_pos;
    while (0 < fscanf(fp, "%1023s\t%u\t%c\t%1023s\t%d", name, &pos, &ref, mut, &is_hap)) {
        while (i < c->n && 0 != strcmp(name, c->name[i])) { i++; prev_pos = 0; }
        if (c->n == i) { fprintf(stderr, "Error: mutation contig not found or out of order [%s]\n", name); exit(1); }
        else if (pos <= 0 || c->len[i] < pos) { fprintf(stderr, "Error: start out of range [%s,%u]\n", name, pos); exit(1); }
        else if (pos < prev_pos) { fprintf(stderr, "Error: out of order [%s,%u]\n", name, pos); exit(1); }   /* prev_pos is never advanced in the reference */
        int type;
        if ('-' == ref && '-' != mut[0]) type = T_INS;
        else if ('-' != ref && '-' == mut[0]) type = T_DEL;
        else if ('-' != ref && '-' != mut[0]) {
            type = T_SUB;
            if (is_hap < 3) {
                if (nt4(mut[0]) < 4) { fprintf(stderr, "Error: heterozygous bases must be in IUPAC form\n"); exit(1); }
                mut[0] = iupac_and_base_to_mut(mut[0], ref);
                if ('X' == mut[0]) { fprintf(stderr, "Error: out of range\n"); exit(1); }
                mut[1] = '\0';
            }
        } else { fprintf(stderr, "Error: out of range\n"); exit(1); }
        mi_push_txt(m, (uint32_t)i, pos, type, is_hap, mut);
    }
}
static void parse_bed(FILE *fp, const ctab_t *c, mutin_t *m) /* mut_bed.c:37-137 */
{
    char name[1024], type[1024], bases[1024]; uint32_t start, end, prev_contig = 0, max_end = 0; int i = 0;
    while (0 < fscanf(fp, "%1023s\t%u\t%u\t%1023s\t%1023s", name, &start, &end, bases, type)) {
        while (i < c->n && 0 != strcmp(name, c->name[i])) i++;
        if (c->n == i) { fprintf(stderr, "Error: contig not found [%s]\n", name); exit(1); }
        else if (c->len[i] <= start) { fprintf(stderr, "Error: start out of range [%s,%u]\n", name, start); exit(1); }
        else if (c->len[i] < end) { fprintf(stderr, "Error: end out of range [%s,%u]\n", name, end); exit(1); }
        else if (end <= start) { fprintf(stderr, "Error: end <= start [%s,%u,%u]\n", name, start, end); exit(1); }
        else if (0 != strcmp("*", bases) && (end - start) != strlen(bases)) { fprintf(stderr, "Error: bases did not match start and end [%s,%u,%u,%s]\n", name, start, end, bases); exit(1); }
        else if (prev_contig == (uint32_t)i && start + 1 <= max_end) { fprintf(stderr, "Warning: overlapping entries, ignoring entry [%s\t%u\t%u\t%s\t%s]\n", name, start, end, bases, type); continue; }
        if (prev_contig != (uint32_t)i || max_end < end) { prev_contig = (uint32_t)i; max_end = end; }
        int ty = get_muttype(type);
        if (ty == T_INS && 26 < end - start) { fprintf(stderr, "Error: insertion of length %d exceeded the maximum supported length of %d\n", end - start, 26); exit(1); }
        if (ty < 0) { fprintf(stderr, "Error: mutation type unrecognized [%s]\n", type); exit(1); }
        m->b = realloc(m->b, sizeof(min_bed_t) * (size_t)(m->nb + 1));
        min_bed_t *e = &m->b[m->nb++];
        e->contig = (uint32_t)i; e->start = start; e->end = end; e->type = (uint8_t)ty; e->bases = strdup(bases);
    }
}
static void parse_vcf(FILE *fp, const ctab_t *c, mutin_t *m) /* mut_vcf.c:42-280, line by line */
{
    static int warned = 0;
    char *line = NULL; size_t cap = 0; ssize_t got;
    /* the reference splits at '\n' or '\r' inside a 1 MiB window; whole lines are equivalent for lines shorter than that */
    char *all = NULL; size_t n_all = 0, m_all = 0; int ch;
    while ((ch = fgetc(fp)) != EOF) { if (n_all + 1 >= m_all) { m_all = m_all ? m_all * 2 : 1 << 16; all = realloc(all, m_all); } all[n_all++] = (char)ch; }
    (void)line; (void)cap; (void)got;
    int i = 0; uint32_t prev_pos = 0;
    size_t s = 0;
    while (s < n_all) {
        size_t n = s;
        while (n < n_all && all[n] != '\n' && all[n] != '\r') n++;
        if (s == n) { s++; continue; }
        if (all[s] == '#') { s = n; continue; }
        char name[1024], id[1024], ref[1024], alt[1025]; uint32_t pos = 0, is_hap;
        char save = n < n_all ? all[n] : 0; if (n < n_all) all[n] = 0;
        if (EOF == sscanf(all + s, "%1023s\t%u\t%1023s\t%1023s\t%1024s", name, &pos, id, ref, alt)) { fprintf(stderr, "Error: VCF parsing error\n"); exit(1); }
        if (n < n_all) all[n] = save;
        is_hap = 4;
        size_t q = s;
        while (q + 4 < n) {
            if (('\t' == all[q] || ';' == all[q]) && 'p' == all[q + 1] && 'l' == all[q + 2] && '=' == all[q + 3]) {
                q += 4;
                switch (all[q]) { case '1': is_hap = 1; break; case '2': is_hap = 2; break; case '3': is_hap = 3; break;
                default: fprintf(stderr, "Error: Could not determine the strand of the mutation from the 'pl' tag.\n"); exit(1); }
                break;
            }
            q++;
        }
        if (4 == is_hap && 0 == warned) { fprintf(stderr, "Warning: strand of the mutation not found; please use the 'pl' tag.\n"); warned = 1; is_hap = 3; }
        while (i < c->n && 0 != strcmp(name, c->name[i])) { i++; prev_pos = 0; }
        if (c->n == i) { fprintf(stderr, "Error: contig not found [%s]\n", name); exit(1); }
        else if (pos <= 0 || c->len[i] < pos) { fprintf(stderr, "Error: start out of range [%s,%u]\n", name, pos); exit(1); }
        else if (pos < prev_pos) { fprintf(stderr, "Error: out of order [%s,%u]\n", name, pos); exit(1); }
        int ref_l = (int)strlen(ref), alt_l = (int)strlen(alt), j;
        if (1 == ref_l && ref[0] == '.') { ref[0] = 0; ref_l = 0; }
        if (1 == alt_l && alt[0] == '.') { alt[0] = 0; alt_l = 0; }
        if (0 == alt_l && 0 == ref_l) { fprintf(stderr, "Error: empty alleles\n"); exit(1); }
        for (j = 0; j < alt_l; ++j) if (',' == alt[j]) { fprintf(stderr, "Error: multiple alleles are not supported\n"); exit(1); }
        for (j = 0; j < ref_l; ++j) { ref[j] = "ACGTNN"[nt4(ref[j])]; if ('N' == ref[j]) { fprintf(stderr, "Error: non-ACGT base found\n"); exit(1); } }
        for (j = 0; j < alt_l; ++j) { alt[j] = "ACGTNN"[nt4(alt[j])]; if ('N' == alt[j]) { fprintf(stderr, "Error: non-ACGT base found\n"); exit(1); } }
        if (ref_l == alt_l) {
            for (j = 0; j < ref_l; ++j) { char b2[2] = { alt[j], 0 }; mi_push_txt(m, (uint32_t)i, pos + (uint32_t)j, T_SUB, is_hap, b2); }
        } else if (ref_l < alt_l) {
            for (j = 0; j < ref_l; ++j, ++pos) if (ref[j] != alt[j]) break;
            mi_push_txt(m, (uint32_t)i, pos, T_INS, is_hap, alt + j);
        } else {
            for (j = 0; j < alt_l; ++j, ++pos) if (ref[j] != alt[j]) break;
            if (j == ref_l) { fprintf(stderr, "Error: no deleted bases\n"); exit(1); }
            for (; j < ref_l; ++j, ++pos) mi_push_txt(m, (uint32_t)i, pos, T_DEL, is_hap, NULL);
        }
        prev_pos = pos;
        s = n;
    }
    free(all);
}

/* mut.c:282-377 mut_add_ins() with given bases (hap >= 0) or a given random length (bed "*") */
static void input_add_ins(rng_t *r, hap_t *h0, hap_t *h1, int64_t i, uint8_t c, int hap, const char *bases, uint32_t num, uint64_t entry)
{
    if (bases) num = (uint32_t)strlen(bases);
    uint8_t *P = malloc(num ? num : 1);
    if (!bases) for (uint32_t j = 0; j < num; ++j) P[num - 1 - j] = (uint8_t)(uint64_t)(rng_u(r, D_MUTIN_BASE, entry, 0, 0, j) * 4.0);
    else for (int64_t j = (int64_t)num - 1; j >= 0; --j) {     /* last base first: the order in which N / unknown bases consume draws (mut.c:317-321, :347-355) */
        int b = nt4(bases[j]);
        if (b >= 4) b = (int)(rng_u(r, D_MUTIN_BASE, entry, 0, 0, (uint32_t)j) * 4.0);
        P[j] = (uint8_t)b;
    }
    hap_t *h[2] = { h0, h1 };
    for (int x = 0; x < 2; ++x) if (hap & (1 << x)) {
        /* a second insertion at the same cell replaces the first one's payload, as overwriting the reference's cell does */
        int found = 0;
        for (int k = h[x]->n_ins - 1; k >= 0 && h[x]->ins[k].pos >= i; --k) if (h[x]->ins[k].pos == i) { free(h[x]->ins[k].b); h[x]->ins[k].b = malloc(num ? num : 1); memcpy(h[x]->ins[k].b, P, num); h[x]->ins[k].n = num; found = 1; break; }
        if (!found) {   /* keep the table sorted by position: input files need not be sorted within a contig for -b */
            ins_push(h[x], i, num, P);
            for (int k = h[x]->n_ins - 1; k > 0 && h[x]->ins[k - 1].pos > h[x]->ins[k].pos; --k) { ins_t t = h[x]->ins[k]; h[x]->ins[k] = h[x]->ins[k - 1]; h[x]->ins[k - 1] = t; }
        }
        h[x]->c[i] = T_INS | c;
    }
    free(P);
}

/* mut.c:644-745: the haplotypes start as the reference, then the file's entries for this contig are applied in file order */
static void apply_mutation_input(const opt_t *o, rng_t *r, const seq_t *seq, hap_t *h0, hap_t *h1, uint32_t contig_i, const mutin_t *m)
{
    hap_t *h[2] = { h0, h1 };
    for (int x = 0; x < 2; ++x) { h[x]->l = seq->l; h[x]->c = calloc((size_t)seq->l + 1, 1); h[x]->ins = NULL; h[x]->n_ins = h[x]->m_ins = 0; }
    for (int64_t i = 0; i < seq->l; ++i) h0->c[i] = h1->c[i] = nt4(seq->s[i]);
    if (m->type == 0) {
        for (int k = 0; k < m->nb; ++k) {
            const min_bed_t *e = &m->b[k];
            if (e->contig == contig_i) {
                int has_bases = strcmp("*", e->bases) != 0, is_hom = 0, hap, which = 0;
                if (o->is_hap || rng_u(r, D_MUTIN, (uint64_t)k, 0, 0, 0) < 0.333333) { is_hom = 1; hap = 3; }
                else { which = rng_u(r, D_MUTIN, (uint64_t)k, 0, 0, 1) < 0.5 ? 0 : 1; hap = 1 << which; }
                if (e->type == T_SUB) {
                    for (uint32_t j = e->start; j < e->end; ++j) {
                        uint8_t c = nt4(seq->s[j]);
                        if (!has_bases) { double rr = rng_u(r, D_MUTIN_BASE, (uint64_t)k, 0, 0, j - e->start); c = (uint8_t)((c + (uint64_t)(rr * 3.0 + 1)) & 3); }
                        else c = nt4(e->bases[j - e->start]);
                        if (is_hom) h0->c[j] = h1->c[j] = T_SUB | c; else h[which]->c[j] = T_SUB | c;
                    }
                } else if (e->type == T_DEL) {
                    for (uint32_t j = e->start; j < e->end; ++j) {
                        uint8_t c = nt4(seq->s[j]);
                        if (is_hom) h0->c[j] = h1->c[j] = T_DEL | c; else h[which]->c[j] = T_DEL | c;
                    }
                } else {
                    uint8_t c = nt4(seq->s[e->start]);
                    if (!has_bases) input_add_ins(r, h0, h1, e->start, c, hap, NULL, e->end - e->start, (uint64_t)k);
                    else input_add_ins(r, h0, h1, e->start, c, hap, e->bases, 0, (uint64_t)k);
                }
            } else if (contig_i < e->contig) break;
        }
    } else {
        for (int k = 0; k < m->nt; ++k) {
            const min_txt_t *e = &m->t[k];
            if (e->contig != contig_i) continue;
            const int64_t p = (int64_t)e->pos - 1;
            uint8_t c = nt4(seq->s[p]);
            if (e->type == T_DEL) { if (e->is_hap & 1) h0->c[p] |= T_DEL | c; if (e->is_hap & 2) h1->c[p] |= T_DEL | c; }
            else if (e->type == T_SUB) { if (e->is_hap & 1) h0->c[p] = T_SUB | nt4(e->bases[0]); if (e->is_hap & 2) h1->c[p] = T_SUB | nt4(e->bases[0]); }
            else input_add_ins(r, h0, h1, p, c, e->is_hap, e->bases, 0, (uint64_t)k);
        }
    }
}

/* mut.c:427-478 mut_left_justify_ins(): while the cell to the left is unmutated and its base
 * (low 2 bits -- N aliases to A) equals the LAST inserted base, rotate the
 * insertion one base to the left. */
static void justify_ins(hap_t *h, int64_t i)
{
    ins_t *e = ins_find(h, i);
    int64_t j = i;
    while (j > 0 && (h->c[j - 1] & TMASK) == T_NONE && e->b[e->n - 1] == (h->c[j - 1] & 3)) {
        memmove(e->b + 1, e->b, e->n - 1);
        e->b[0] = h->c[j - 1] & 3;
        h->c[j] = h->c[j] & 3;
        j--;
    }
    h->c[j] = T_INS | (h->c[j] & 3);
    e->pos = j;
}

/* one haplotype's part of a deletion shift, mut.c:515-516 / :547 / :569 */
static inline void del_swap(hap_t *h, int64_t j, int64_t dl)
{
    uint8_t t = h->c[j]; h->c[j] = h->c[j + dl]; h->c[j + dl] = (uint8_t)((t | TMASK) ^ TMASK);
}
static inline int64_t del_run(const hap_t *h, int64_t i)
{
    int64_t j, dl = 1;
    for (j = i + 1; j < h->l && (h->c[j] & TMASK) == T_DEL; ++j) dl++;
    return dl;
}

/* mut.c:481-589 mut_left_justify() */
static void left_justify(const seq_t *seq, hap_t *h0, hap_t *h1)
{
    int prev_del[2] = { 0, 0 };
    hap_t *h[2] = { h0, h1 };
    for (int64_t i = 0; i < seq->l; ++i) {
        uint8_t r0 = nt4(seq->s[i]), c1 = h0->c[i], c2 = h1->c[i];
        if (r0 >= 4) continue;
        if ((c1 & TMASK) == T_NONE && (c2 & TMASK) == T_NONE) { prev_del[0] = prev_del[1] = 0; continue; }
        if ((c1 & BT_MASK) == (c2 & BT_MASK)) { /* hom */
            if ((c1 & TMASK) == T_SUB) { prev_del[0] = prev_del[1] = 0; }
            else if ((c1 & TMASK) == T_DEL) {
                if (prev_del[0] == 1 || prev_del[1] == 1) continue;
                prev_del[0] = prev_del[1] = 1;
                int64_t dl = del_run(h0, i);
                if (seq->l <= i + dl) continue;
                if (i > 0) for (int64_t j = i - 1;; --j) {
                    uint8_t a = h0->c[j], b = h1->c[j];
                    if ((a & TMASK) != T_INS && (b & TMASK) != T_INS && (a & TMASK) != T_DEL && (b & TMASK) != T_DEL
                        && (a & 3) == (h0->c[j + dl] & 3) && (b & 3) == (h1->c[j + dl] & 3)) {
                        del_swap(h0, j, dl); del_swap(h1, j, dl);
                    } else break;
                    if (j == 0) break;
                }
            } else { /* insertion */
                prev_del[0] = prev_del[1] = 0;
                justify_ins(h0, i); justify_ins(h1, i);
            }
        } else { /* het */
            if ((c1 & TMASK) == T_SUB || (c2 & TMASK) == T_SUB) { prev_del[0] = prev_del[1] = 0; }
            else if ((c1 & TMASK) == T_DEL || (c2 & TMASK) == T_DEL) {
                int x = ((c1 & TMASK) == T_DEL) ? 0 : 1;
                if (prev_del[x] == 1) continue;
                prev_del[x] = 1;
                int64_t dl = del_run(h[x], i);
                if (seq->l <= i + dl) continue;
                if (i > 0) for (int64_t j = i - 1;; --j) {
                    uint8_t a = h[x]->c[j];
                    if ((a & TMASK) == T_NONE && (a & 3) == (h[x]->c[j + dl] & 3)) del_swap(h[x], j, dl);
                    else break;
                    if (j == 0) break;
                }
            } else if ((c1 & TMASK) == T_INS) { prev_del[0] = prev_del[1] = 0; justify_ins(h0, i); }
            else { prev_del[0] = prev_del[1] = 0; justify_ins(h1, i); }
        }
    }
}

/* growable text sink */
typedef struct { char *p; size_t n, m; FILE *fp; int discard; uint64_t total; } sink_t;
static void sink_flush(sink_t *s) { if (s->fp && s->n) fwrite(s->p, 1, s->n, s->fp); s->n = 0; }
static inline void sink_putc(sink_t *s, char c)
{
    s->total++;
    if (s->discard) return;
    if (s->n == s->m) { if (s->fp && s->m >= (1u << 20)) sink_flush(s); else { s->m = s->m ? s->m * 2 : 1 << 16; s->p = realloc(s->p, s->m); } }
    s->p[s->n++] = c;
}
static void sink_puts(sink_t *s, const char *z) { while (*z) sink_putc(s, *z++); }
static void sink_printf(sink_t *s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void sink_printf(sink_t *s, const char *fmt, ...)
{
    char buf[4096]; va_list ap; va_start(ap, fmt); int n = vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (n >= (int)sizeof buf) { char *b = malloc((size_t)n + 1); va_start(ap, fmt); vsnprintf(b, (size_t)n + 1, fmt, ap); va_end(ap); sink_puts(s, b); free(b); }
    else sink_puts(s, buf);
}

static void print_ins(sink_t *s, hap_t *h, int64_t i) /* mut.c:249-279 */
{
    ins_t *e = ins_find(h, i);
    for (uint32_t t = 0; t < e->n; ++t) sink_putc(s, "ACGTN"[e->b[t] & 3]);
}

/* mut.c:781-893 mut_print() */
static void print_mutations(const char *name, const seq_t *seq, hap_t *h0, hap_t *h1, sink_t *txt, sink_t *vcf)
{
    int prev[2] = { 0, 0 };
    for (int64_t i = 0; i < seq->l; ++i) {
        uint8_t r0 = nt4(seq->s[i]), c1 = h0->c[i], c2 = h1->c[i];
        if (r0 < 4 && ((c1 & TMASK) != T_NONE || (c2 & TMASK) != T_NONE)) {
            sink_printf(txt, "%s\t%lld\t", name, (long long)i + 1);
            int hom = (c1 & BT_MASK) == (c2 & BT_MASK);
            if ((hom && (c1 & TMASK) == T_SUB) || (!hom && ((c1 & TMASK) == T_SUB || (c2 & TMASK) == T_SUB))) {
                if (hom) {
                    sink_printf(txt, "%c\t%c\t3\n", "ACGTN"[r0], "ACGTN"[c1 & 0xf]);
                    sink_printf(vcf, "%s\t%lld\t.\t%c\t%c\t100\tPASS\tAF=1.0;pl=3;mt=SUBSTITUTE\n", name, (long long)i + 1, "ACGTN"[r0], "ACGTN"[c1 & 0xf]);
                } else {
                    int hap = ((c1 & TMASK) == T_SUB) ? 1 : 2;
                    sink_printf(txt, "%c\t%c\t%d\n", "ACGTN"[r0], "XACMGRSVTWYHKDBN"[1 << (c1 & 3) | 1 << (c2 & 3)], hap);
                    sink_printf(vcf, "%s\t%lld\t.\t%c\t%c\t100\tPASS\tAF=0.5;pl=%d;mt=SUBSTITUTE\n", name, (long long)i + 1, "ACGTN"[r0], "ACGTN"[(hap == 1 ? c1 : c2) & 0xf], hap);
                }
            } else if ((hom && (c1 & TMASK) == T_DEL) || (!hom && ((c1 & TMASK) == T_DEL || (c2 & TMASK) == T_DEL))) {
                /* hom: pl 3; het: haplotype 1 takes precedence (mut.c:833 before :853) */
                int pl = hom ? 3 : (((c1 & TMASK) == T_DEL) ? 1 : 2);
                sink_printf(txt, "%c\t-\t%d\n", "ACGTN"[r0], pl);
                int open = hom ? (prev[0] == 0 || prev[1] == 0) : (prev[pl - 1] == 0);
                if (open) { /* one VCF record per run, anchored on the previous reference base (POS = i, 1-based i) */
                    sink_printf(vcf, "%s\t%lld\t.\t", name, (long long)i);
                    if (i > 0) sink_putc(vcf, "ACGTN"[nt4(seq->s[i - 1])]);
                    uint8_t a0 = r0, a1 = c1, a2 = c2;
                    for (int64_t j = i; j < seq->l; ++j) {
                        int h2 = (a1 & BT_MASK) == (a2 & BT_MASK);
                        uint8_t td = (pl == 2) ? a2 : a1;
                        if (!(h2 == hom && (td & TMASK) == T_DEL)) break;
                        sink_putc(vcf, "ACGTN"[a0]);
                        if (j + 1 < seq->l) { a0 = nt4(seq->s[j + 1]); a1 = h0->c[j + 1]; a2 = h1->c[j + 1]; }
                    }
                    if (i > 0) sink_printf(vcf, "\t%c", "ACGTN"[nt4(seq->s[i - 1])]); else sink_puts(vcf, "\t.");
                    sink_printf(vcf, "\t100\tPASS\tAF=%s;pl=%d;mt=DELETE\n", hom ? "1.0" : "0.5", pl);
                }
            } else { /* insertion */
                int pl = hom ? 3 : (((c1 & TMASK) == T_INS) ? 1 : 2);
                hap_t *hh = (pl == 2) ? h1 : h0;
                if (!hom && (c1 & TMASK) != T_INS && (c2 & TMASK) != T_INS) { fprintf(stderr, "oracle: unreachable mutation state\n"); exit(2); }
                sink_puts(txt, "-\t"); print_ins(txt, hh, i); sink_printf(txt, "\t%d\n", pl);
                sink_printf(vcf, "%s\t%lld\t.\t%c\t%c", name, (long long)i + 1, "ACGTN"[r0], "ACGTN"[r0]);
                print_ins(vcf, hh, i);
                sink_printf(vcf, "\t100\tPASS\tAF=%s;pl=%d;mt=INSERT\n", hom ? "1.0" : "0.5", pl);
            }
        }
        prev[0] = (c1 & TMASK) != T_NONE; prev[1] = (c2 & TMASK) != T_NONE;
    }
}

/* ------------------------------------------------------------------------------------------
 * Read extraction: dwgsim.c:75-153 __gen_read
 * ---------------------------------------------------------------------------------------- */
typedef struct { int ext_coor, n_sub, n_indel, n_sub_first, n_indel_first; } readinfo_t;

static void gen_read(hap_t *h, int64_t seq_l, int64_t start, int step, int s, int strand, uint8_t *out, readinfo_t *ri)
{
    int k = 0; int64_t i;
    ri->ext_coor = -10;
    for (i = start; i >= 0 && i < seq_l && k < s; i += step) {
        uint8_t c = h->c[i], mt = c & TMASK;
        if (ri->ext_coor < 0) {
            if (mt != T_NONE && mt != T_SUB) continue;
            ri->ext_coor = (int)i;
            if (strand == 1) ri->ext_coor -= s - 1;
        }
        if (mt == T_DEL) {
            ++ri->n_indel;
            if (strand == 1) ri->ext_coor--;
            if (k == 0) ri->n_indel_first++;
        } else if (mt == T_NONE || mt == T_SUB) {
            out[k++] = c & 0xf;
            if (mt == T_SUB) { ++ri->n_sub; if (k == 0) ri->n_sub_first++; }
        } else {
            ins_t *e = ins_find(h, i);
            uint32_t n = e->n;
            ++ri->n_indel; ri->n_indel_first++;
            if (strand == 0) {
                if (k < s) out[k++] = c & 0xf;
                for (uint32_t t = 0; t < n && k < s; ++t) out[k++] = e->b[t] & 3;
            } else {
                while (n > 0 && k < s) { ri->ext_coor++; out[k++] = e->b[n - 1] & 3; --n; }
                if (k < s) out[k++] = c & 0xf;
            }
        }
    }
    if (k != s) ri->ext_coor = -10;
    if (strand == 1) for (k = 0; k < s; ++k) out[k] = out[k] < 4 ? 3 - out[k] : 4;
}

/* ------------------------------------------------------------------------------------------
 * Ion Torrent flow-space errors: dwgsim.c:246-417 generate_errors_flows
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint8_t *seq, *mask; int mem; } flowbuf_t;
static void flow_grow(flowbuf_t *b, int need) /* dwgsim.c:296-311: grow while mem <= need (allocation only, no effect on results) */
{
    int grew = 0;
    while (b->mem <= need) { b->mem <<= 1; grew = 1; }
    if (grew) { b->seq = realloc(b->seq, (size_t)b->mem); b->mask = realloc(b->mask, (size_t)b->mem); }
}
static void flow_alloc(flowbuf_t *b, int len, int F)
{
    b->mem = (len + 2 > F + 2) ? len + 2 : F + 2;
    b->seq = calloc((size_t)b->mem, 1); b->mask = calloc((size_t)b->mem, 1);
}
/* mode B draws of the flow model (round 6).  An "event" is a homopolymer start in pass 1 or an empty flow in pass 2; the reference gives each a
 * FIRST uniform, `drand48() < e` (dwgsim.c:296, :373), and 99 % of the events draw nothing else.  Rounds 2-5 gave every event its own 16 + 16 bit
 * uniform: one Philox block per eight events, 170 blocks per 400-base read, and the kernels spent most of the flow model drawing and comparing
 * numbers that say "no".  The first draws of a pass are a Bernoulli(e') sequence, e' = thr / 2^32 with thr = ceil(e 2^32) as everywhere (u < e <=>
 * w < thr); such a sequence is, in law, its GAPS: the number of quiet draws in front of each scoring one is Geometric(e'), independent.  So mode B
 * draws the gaps, by inversion: gap m of the pass comes from word m & 3 of block m >> 2 of the domain (D_FLOW0 + j, + D_FLOW_PASS2),
 *     G = floor(-log2(U) / -log2(1 - e')),  U = (2 w + 1) / 2^33,
 * in INTEGER arithmetic only (flow_gap below: a 257-entry table of log2 made by repeated squaring, fixed point, one 64 x 64 -> 128 multiplication), so
 * that gcc and the kernels agree bit for bit.  The ordinals of the scoring first draws are S_0 = G_0, S_(m+1) = S_m + 1 + G_(m+1); the k-th first draw
 * of the pass (k = 0, 1, ...: homopolymer starts examined so far in pass 1, empty flows examined so far in pass 2) scores iff k is one of them.
 * Every further draw of an event (more errors, insert or delete, the dot-fill flow) is word s & 3 of the block (retry s >> 2, block h) of domain +
 * D_FLOW_EV, s = 0, 1, ..., with h = the event's position in the evolving read (pass 1) / the number of empty flows examined before it (pass 2), as before.
 * --dump-draws (replay through the unmodified reference): a first draw is handed over as 0 (scores) or 1 - 2^-48 (quiet) -- the reference only
 * compares it with e. */
typedef struct { int init; uint32_t m, next; gap_par_t par; } gapstate_t;
/* does first draw number `ordinal` (0, 1, 2, ... in the order the reference makes them) of the pass score? */
static inline int flow_first(rng_t *r, uint32_t fdom, uint64_t idx, uint32_t att, uint32_t ordinal, double e, gapstate_t *gs)
{
    r->n_draws++;
    if (r->mode == RNG_DRAND48) return oracle_drand48_next(&r->x) < e;
    if (!gs->init) {
        gs->init = 1; gs->m = 0;
        gs->par = flow_gap_params(!(e > 0) ? 0 : e >= 1.0 ? 0x100000000ull : (uint64_t)ceil(e * 4294967296.0));
        gs->next = FLOW_NEVER;
    }
    if (gs->par.thr && gs->m == 0) { gs->next = flow_gap((uint32_t)(oracle_philox_uniform32(r->k0, r->k1, fdom, idx, att, 0, 0) * 4294967296.0), &gs->par); gs->m = 1; }
    const int hit = ordinal == gs->next;
    if (hit) { const uint32_t m = gs->m++; gs->next = ordinal + 1u + flow_gap((uint32_t)(oracle_philox_uniform32(r->k0, r->k1, fdom, idx, att, 0, m) * 4294967296.0), &gs->par); }
    rng_out(r, hit ? 0.0 : 1.0 - 0x1p-48);
    return hit;
}
static inline double flow_u(rng_t *r, uint32_t fdom, uint64_t idx, uint32_t att, uint32_t evt, uint32_t *es)
{
    const uint32_t s = (*es)++;
    return rng_u32(r, fdom + D_FLOW_EV, idx, att, s >> 2, (evt << 2) | (s & 3));
}
#define FLOW_U() flow_u(r, fdom, idx, att, evt, &es)
static int flow_errors(const opt_t *o, rng_t *r, uint32_t dom, uint64_t idx, uint32_t att, uint32_t *slot,
                       flowbuf_t *b, int len, int strand, double e, int *n_err_out)
{
    int i, j, k, hp_l, flow_i, n_err, F = o->flow_order_len;
    uint8_t prev_c, c;
    uint32_t fdom = dom, evt = 0, es = 0, g = 0, k1 = 0; (void)slot;
    gapstate_t gs1, gs2; memset(&gs1, 0, sizeof gs1); memset(&gs2, 0, sizeof gs2);
    for (i = 0; i < len; ++i) if (b->seq[i] >= 4) b->seq[i] = 0;
    if (strand == 1) for (i = 0; i < len >> 1; ++i) { c = b->seq[i]; b->seq[i] = b->seq[len - i - 1]; b->seq[len - i - 1] = c; }
    for (i = 0; i < F; ++i) {
        c = (4 <= b->seq[0]) ? 0 : b->seq[0]; /* NB: reads seq[0] even when len == 0 (dwgsim.c:268-274) */
        if (c == o->flow_order[i]) break;
        b->mask[i] = 0;
    }
    if (F == i) { fprintf(stderr, "Error: first base not found in flow order\n"); return -1; }
    flow_i = i; prev_c = 4;
    for (i = 0; i < len; ++i) {
        c = (4 <= b->seq[i]) ? 0 : b->seq[i];
        while (c != o->flow_order[flow_i]) { b->mask[flow_i] = 0; flow_i = (flow_i + 1) % F; }
        if (prev_c != c) {
            b->mask[flow_i] = 0;
            evt = (uint32_t)i; es = 0;
            n_err = 0;
            if (flow_first(r, fdom, idx, att, k1++, e, &gs1)) { n_err = 1; while (FLOW_U() < e) n_err++; } /* while(drand48() < e) n_err++ (dwgsim.c:296) */
            if (0 < n_err) {
                if (FLOW_U() < 0.5) { /* insert */
                    flow_grow(b, len + n_err);
                    for (j = len - 1; i <= j; --j) b->seq[j + n_err] = b->seq[j];
                    for (j = i; j < i + n_err; ++j) b->seq[j] = c;
                    len += n_err;
                } else { /* delete */
                    int next_c = 4;
                    for (j = i, hp_l = 0; j < len; ++j, ++hp_l) { next_c = (4 <= b->seq[j]) ? 0 : b->seq[j]; if (c != next_c) break; }
                    n_err = (hp_l < n_err) ? hp_l : n_err;
                    for (j = i; j < len - n_err; ++j) b->seq[j] = b->seq[j + n_err];
                    len -= n_err;
                    b->mask[flow_i] = 1;
                    if (n_err == hp_l && (0 == i || prev_c == next_c)) { /* dot-fill */
                        j = 0;
                        while (next_c != o->flow_order[(flow_i + j) % F]) j++;
                        if (j == 0) { fprintf(stderr, "oracle: assert(0 < j) (dwgsim.c:349) fails: the whole read was one deleted homopolymer\n"); exit(134); } /* the reference abort()s here */
                        k = (int)(FLOW_U() * j);
                        for (j = len - 1; i <= j; --j) b->seq[j + 1] = b->seq[j];
                        b->seq[i] = (uint8_t)o->flow_order[(flow_i + k) % F];
                        len++;
                    }
                }
                *n_err_out += n_err;
            }
            prev_c = c;
        }
    }
    fdom = dom + D_FLOW_PASS2;
    for (i = 0; i < len; ++i) { /* second pass: empty flows (flow_i continues) */
        c = (4 <= b->seq[i]) ? 0 : b->seq[i];
        while (c != o->flow_order[flow_i]) {
            evt = g++; es = 0;
            n_err = 0;
            if (flow_first(r, fdom, idx, att, evt, e, &gs2)) { n_err = 1; while (FLOW_U() < e) n_err++; } /* dwgsim.c:373 */
            if (0 == b->mask[flow_i] && 0 < n_err) {
                flow_grow(b, len + n_err);
                for (j = len - 1; i <= j; --j) b->seq[j + n_err] = b->seq[j];
                for (j = i; j < i + n_err; ++j) b->seq[j] = (uint8_t)o->flow_order[flow_i];
                len += n_err;
                *n_err_out += n_err;
            }
            flow_i = (flow_i + 1) % F;
        }
    }
    if (strand == 1) for (i = 0; i < len >> 1; ++i) { c = b->seq[i]; b->seq[i] = b->seq[len - i - 1]; b->seq[len - i - 1] = c; }
    return len;
}

/* ------------------------------------------------------------------------------------------
 * Option parsing: dwgsim_opt.c:204-472 (mutation-input / region options are accepted by the
 * reference but are outside this oracle's scope: SURVEY 8(f))
 * ---------------------------------------------------------------------------------------- */
static void get_error_rate(const char *str, erate_t *e) /* dwgsim_opt.c:162-179 */
{
    size_t i, n = strlen(str);
    e->start = atof(str);
    for (i = 0; i < n; ++i) if (str[i] == ',' || str[i] == '-') break;
    if (n > 0 && i < n - 1) e->end = atof(str + i + 1); else e->end = e->start;
}
static int is_int(const char *a, int neg_ok) /* dwgsim_opt.c:181-192 */
{
    size_t n = strlen(a);
    if (n == 0) return 0;
    if ('+' != a[0] && (neg_ok == 0 || '-' != a[0]) && !isdigit((unsigned char)a[0])) return 0;
    for (size_t i = 1; i < n; ++i) if (!isdigit((unsigned char)a[i])) return 0;
    return 1;
}
static int xatoi(const char *a, char flag, int neg_ok)
{
    if (!is_int(a, neg_ok)) { fprintf(stderr, "Error: command line option -%c is not a number [%s]\n", flag, a); exit(1); }
    return atoi(a);
}
#define CHECK(v, lo, hi, nm) do { if ((v) < (lo) || (hi) < (v)) { fprintf(stderr, "Error: command line option %s was out of range\n", nm); return 0; } } while (0)

static int calibrate_flow_error(opt_t *o, rng_t *r);

static int opt_parse(opt_t *o, rng_t *r, int argc, char **argv, int *first_arg)
{
    int c, muts_flags = 0;
    optind = 1;
    while ((c = getopt(argc, argv, "id:s:N:C:1:2:e:E:r:F:R:X:I:c:S:A:n:y:BHf:z:M:m:b:v:x:P:q:Q:o:ah")) >= 0) {
        switch (c) {
        case 'i': o->is_inner = 1; break;
        case 'd': o->dist = xatoi(optarg, 'd', 0); break;
        case 's': o->std_dev = atof(optarg); break;
        case 'N': o->N = xatoi(optarg, 'N', 1); o->C = -1; break;
        case 'C': o->C = atof(optarg); o->N = -1; break;
        case '1': o->length[0] = xatoi(optarg, '1', 0); break;
        case '2': o->length[1] = xatoi(optarg, '2', 0); break;
        case 'e': get_error_rate(optarg, &o->e[0]); break;
        case 'E': get_error_rate(optarg, &o->e[1]); break;
        case 'r': o->mut_rate = atof(optarg); break;
        case 'F': o->mut_freq = atof(optarg); break;
        case 'R': o->indel_frac = atof(optarg); break;
        case 'X': o->indel_extend = atof(optarg); break;
        case 'I': o->indel_min = xatoi(optarg, 'I', 0); break;
        case 'c': o->data_type = xatoi(optarg, 'c', 0); break;
        case 'S': o->strandedness = xatoi(optarg, 'S', 0); break;
        case 'A': o->read_one_strand = xatoi(optarg, 'A', 0); break;
        case 'n': o->max_n = xatoi(optarg, 'n', 0); break;
        case 'y': o->rand_read = atof(optarg); break;
        case 'f': free(o->flow_order); o->flow_order = (int8_t *)strdup(optarg); break;
        case 'B': o->use_base_error = 1; break;
        case 'H': o->is_hap = 1; break;
        case 'h': return 0;
        case 'z': o->seed = xatoi(optarg, 'z', 1); break;
        case 'M': o->output_type = xatoi(optarg, 'M', 0); break;
        case 'm': free(o->fn_muts); o->fn_muts = strdup(optarg); o->muts_type = 1; muts_flags |= 1; break;
        case 'b': free(o->fn_muts); o->fn_muts = strdup(optarg); o->muts_type = 0; muts_flags |= 2; break;
        case 'v': free(o->fn_muts); o->fn_muts = strdup(optarg); o->muts_type = 2; muts_flags |= 4; break;
        case 'x': free(o->fn_regions); o->fn_regions = strdup(optarg); break;
        case 'P': free(o->read_prefix); o->read_prefix = strdup(optarg); break;
        case 'q': free(o->fixed_quality); o->fixed_quality = strdup(optarg); break;
        case 'Q': o->quality_std = atof(optarg); break;
        case 'o': o->reads_output_type = atoi(optarg); break;
        case 'a': o->amplicons = 1; break;
        default: fprintf(stderr, "Unrecognized option: -%c\n", c); return 0;
        }
    }
    if (argc - optind < 2) return 0;
    *first_arg = optind;
    CHECK(o->dist, 0, INT32_MAX, "-d");
    CHECK(o->std_dev, 0, INT32_MAX, "-s");
    if (o->N < 0 && o->C < 0) { fprintf(stderr, "Must use one of -N or -C"); return 0; }
    else if (0 < o->N && 0 < o->C) { fprintf(stderr, "Cannot use both -N or -C"); return 0; }
    else if (0 < o->N) { CHECK(o->N, 1, INT32_MAX, "-N"); CHECK(o->C, INT32_MIN, -1, "-C"); }
    else { CHECK(o->N, INT32_MIN, -1, "-N"); CHECK(o->C, 0, INT32_MAX, "-C"); }
    CHECK(o->length[0], 1, INT32_MAX, "-1");
    CHECK(o->length[1], 0, INT32_MAX, "-2");
    for (int i = 0; i < 2; ++i) {
        if (o->e[i].start < 0.0 || 1.0 < o->e[i].start) { fprintf(stderr, "End %s: the start error is out of range (-e)\n", i ? "two" : "one"); return 0; }
        if (o->e[i].end < 0.0 || 1.0 < o->e[i].end) { fprintf(stderr, "End %s: the end error is out of range (-e)\n", i ? "two" : "one"); return 0; }
        if (IONTORRENT == o->data_type && o->e[i].end != o->e[i].start) { fprintf(stderr, "End %s: a uniform error rate must be given for Ion Torrent data\n", i ? "two" : "one"); return 0; }
    }
    CHECK(o->mut_rate, 0, 1.0, "-r"); CHECK(o->indel_frac, 0, 1.0, "-R"); CHECK(o->indel_extend, 0, 1.0, "-X");
    CHECK(o->indel_min, 1, INT32_MAX, "-I"); CHECK(o->data_type, 0, 2, "-c"); CHECK(o->strandedness, 0, 2, "-S");
    CHECK(o->read_one_strand, 0, 2, "-A"); CHECK(o->max_n, 0, INT32_MAX, "-n"); CHECK(o->rand_read, 0, 1.0, "-y");
    if (IONTORRENT == o->data_type && NULL == o->flow_order) { fprintf(stderr, "Error: command line option -f is required\n"); return 0; }
    if (o->fixed_quality && strlen(o->fixed_quality) != 1) { fprintf(stderr, "Error: command line option -q requires one character\n"); return 0; }
    CHECK(o->quality_std, 0, INT32_MAX, "-Q");
    CHECK(o->reads_output_type, 0, 2, "-o");
    if (muts_flags != 0 && muts_flags != 1 && muts_flags != 2 && muts_flags != 4) { fprintf(stderr, "Error: -m/-b/-v cannot be used together\n"); return 0; }

    rng_seed(r, o->rng_mode, (-1 == o->seed) ? (int32_t)time(0) : o->seed);
    r->use_libm_log = o->use_libm_log;
    if (o->dump_path) {
        if (r->mode != RNG_PHILOX) { fprintf(stderr, "oracle: --dump-draws needs --rng philox\n"); return 0; }
        r->dump = fopen(o->dump_path, "wb");
        if (!r->dump) { fprintf(stderr, "oracle: cannot write %s\n", o->dump_path); return 0; }
    }

    if (IONTORRENT == o->data_type) { /* dwgsim_opt.c:396-413 */
        o->flow_order_len = (int)strlen((char *)o->flow_order);
        for (int i = 0; i < o->flow_order_len; ++i) o->flow_order[i] = (int8_t)nt4(o->flow_order[i]);
    }
    if (IONTORRENT == o->data_type && o->use_base_error) { if (!calibrate_flow_error(o, r)) return 0; }
    else { /* dwgsim_opt.c:459-460 (NaN for -2 0: harmless) */
        o->e[0].by = (o->e[0].end - o->e[0].start) / o->length[0];
        o->e[1].by = (o->e[1].end - o->e[1].start) / o->length[1];
    }
    CHECK(o->output_type, 0, 2, "-M");
    if (o->amplicons == 1 && o->fn_regions) { fprintf(stderr, "Error: cannot use a regions BED file (-x) when simulating amplicons (-a)\n"); return 0; }
    return 1;
}

/* dwgsim_opt.c:415-457: rescale the per-flow error so the per-base rate matches -e */
static int calibrate_flow_error(opt_t *o, rng_t *r)
{
    double sf = 0.0;
    for (int i = 0; i < 2; ++i) {
        if (o->length[i] <= 0) continue;
        if (0 < i && o->length[i] == o->length[1 - i]) { o->e[i] = o->e[1 - i]; continue; }
        flowbuf_t b; flow_alloc(&b, o->length[i], o->flow_order_len);
        int n_err = 0, counts = 0; /* int32 accumulators as in the reference */
        for (int j = 0; j < 1000000; ++j) {
            uint32_t slot = 0;
            /* mode B: bases = narrow words of attempt 0, the flow model's sequential stream = attempt 1; per-read mask (see core()) */
            for (int k = 0; k < o->length[i]; ++k) b.seq[k] = (uint8_t)((int)(rng_u32(r, D_CALIB + i, (uint64_t)j, 0, 0, (uint32_t)k) * 4.0) & 3);
            if (r->mode == RNG_PHILOX) memset(b.mask, 0, (size_t)b.mem);
            int cur = 0;
            int s = flow_errors(o, r, D_CALIB + i, (uint64_t)j, 1, &slot, &b, o->length[i], 0, o->e[i].start, &cur);
            n_err += cur; counts += s;
        }
        sf = o->e[i].start / (n_err / (1.0 * counts));
        o->e[i].start = o->e[i].end *= sf;
        o->e[i].by = (o->e[i].end - o->e[i].start) / o->length[i];
        free(b.seq); free(b.mask);
        fprintf(stderr, "[oracle] end %d flow-error scaling factor %.5lf\n", i + 1, sf);
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * dwgsim_core: dwgsim.c:419-1121
 * ---------------------------------------------------------------------------------------- */
typedef struct { sink_t bfast, bwa1, bwa2, txt, vcf; int has_bfast, has_bwa, has_mut; } outs_t;

/* quality string for read end j, dwgsim.c:899-918 (and :1002-1021) */
static void make_quals(const opt_t *o, rng_t *r, int j, uint64_t ii, uint32_t att, int len, char *q)
{
    nstream_t ns = { D_QUAL0 + (uint32_t)j, ii, att, 0, 1, 0, 0.0, 1 };
    int i;
    if (o->fixed_quality) { for (i = 0; i < len; ++i) q[i] = o->fixed_quality[0]; }
    else for (i = 0; i < len; ++i) {
        double ei = o->e[j].start + o->e[j].by * i;
        if (ei > 0) q[i] = (char)((int)(-10.0 * log(ei) / log(10.0) + 0.499) + '!');
        else q[i] = 40 + '!';
        if (0 < o->quality_std) q[i] = (char)(q[i] + (int)((ran_normal(r, &ns) * o->quality_std) + 0.5));
        if (q[i] < '!') q[i] = '!';
        if (40 + '!' < q[i]) q[i] = 40 + '!';
    }
    q[i] = 0;
}

static void put_name(sink_t *s, const opt_t *o, const char *name, unsigned p0, unsigned p1, unsigned s0, unsigned s1,
                     unsigned r0, unsigned r1, const int a[6], unsigned long long ii, int suffix)
{
    sink_printf(s, "@%s%s%s_%u_%u_%1u_%1u_%1u_%1u_%d:%d:%d_%d:%d:%d_%llx", o->read_prefix ? o->read_prefix : "", o->read_prefix ? "_" : "",
                name, p0, p1, s0, s1, r0, r1, a[0], a[1], a[2], a[3], a[4], a[5], ii);
    if (suffix) sink_printf(s, "/%d", suffix);
    sink_putc(s, '\n');
}

/* dwgsim.c:919-981 / :1033-1094: emit read end j to the bwa file j and to bfast */
static void emit_read(const opt_t *o, outs_t *out, int j, const char *name, unsigned p0, unsigned p1, unsigned s0, unsigned s1,
                      unsigned r0, unsigned r1, const int cnt[6], const int cnt_solid_bwa[6], unsigned long long ii,
                      const uint8_t *seq, int len, const char *q)
{
    int i;
    if (out->has_bwa) {
        sink_t *f = j ? &out->bwa2 : &out->bwa1;
        if (o->data_type != SOLID) {
            put_name(f, o, name, p0, p1, s0, s1, r0, r1, cnt, ii, j + 1);
            for (i = 0; i < len; ++i) sink_putc(f, "ACGTN"[seq[i]]);
            sink_puts(f, "\n+\n"); sink_puts(f, q); sink_putc(f, '\n');
        } else {
            put_name(f, o, name, p0, p1, s0, s1, r0, r1, cnt_solid_bwa, ii, 2 - j);
            for (i = 1; i < len; ++i) sink_putc(f, "ACGTN"[seq[i]]);
            sink_puts(f, "\n+\n");
            for (i = 1; i < len; ++i) sink_putc(f, q[i]);
            sink_putc(f, '\n');
        }
    }
    if (out->has_bfast) {
        sink_t *f = &out->bfast;
        put_name(f, o, name, p0, p1, s0, s1, r0, r1, cnt, ii, 0);
        if (o->data_type != SOLID) {
            for (i = 0; i < len; ++i) sink_putc(f, "ACGTN"[seq[i]]);
            sink_puts(f, "\n+\n"); sink_puts(f, q); sink_putc(f, '\n');
        } else {
            sink_putc(f, 'A');
            for (i = 0; i < len; ++i) sink_putc(f, "01234"[seq[i]]);
            sink_puts(f, "\n+\n");
            for (i = 0; i < len; ++i) sink_putc(f, q[i]);
            sink_putc(f, '\n');
        }
    }
}

static void to_colors(uint8_t *seq, int len) /* dwgsim.c:845-858, __gf_add dwgsim.h:6 */
{
    int c1 = 0;
    for (int i = 0; i < len; ++i) { int c2 = seq[i]; seq[i] = (uint8_t)((c1 >= 4 || c2 >= 4) ? 4 : (c1 ^ c2)); c1 = c2; }
}

/* regions_bed.c:38-125 regions_bed_init(): sorted BED, overlapping / touching intervals of a contig are merged */
typedef struct { uint32_t *contig, *start, *end; int n; } regions_t;
static void parse_regions(FILE *fp, const ctab_t *c, regions_t *r)
{
    char name[1024]; uint32_t start, end; int i = 0, b; int32_t prev_contig = -1, prev_start = -1, prev_end = -1;
    while (0 < fscanf(fp, "%1023s\t%u\t%u", name, &start, &end)) {
        while (i < c->n && 0 != strcmp(name, c->name[i])) i++;
        if (c->n == i) { fprintf(stderr, "Error: contig not found [%s].  Are you sure your BED is coordinate sorted?\n", name); exit(1); }
        else if (c->len[i] < start) { fprintf(stderr, "Error: start out of range [%s,%u]\n", name, start); exit(1); }
        else if (c->len[i] < end) { fprintf(stderr, "Error: end out of range [%s,%u]\n", name, end); exit(1); }
        else if (end < start) { fprintf(stderr, "Error: end < start [%s,%u,%u]\n", name, start, end); exit(1); }
        else if (prev_contig == i && start < (uint32_t)prev_start) { fprintf(stderr, "Error: the input was not sorted [%s,%u,%u,%u]\n", name, start, end, end - start); exit(1); }
        if (prev_contig == i && start <= (uint32_t)prev_end && (uint32_t)prev_start <= start) {
            if ((uint32_t)prev_end < end) { r->end[r->n - 1] = end; prev_end = (int32_t)end; }
        } else {
            prev_contig = i; prev_start = (int32_t)start; prev_end = (int32_t)end;
            r->contig = realloc(r->contig, sizeof(uint32_t) * (size_t)(r->n + 1)); r->start = realloc(r->start, sizeof(uint32_t) * (size_t)(r->n + 1)); r->end = realloc(r->end, sizeof(uint32_t) * (size_t)(r->n + 1));
            r->contig[r->n] = (uint32_t)i; r->start[r->n] = start; r->end[r->n] = end; r->n++;
        }
        while (EOF != (b = fgetc(fp))) if ('\n' == b || '\r' == b) break;
    }
}
static int regions_query(const regions_t *r, uint32_t contig, uint32_t start, uint32_t end) /* regions_bed.c:130-156 */
{
    int low = 0, high = r->n - 1;
    while (low <= high) {
        int mid = low + (high - low) / 2;
        if (contig < r->contig[mid] || (contig == r->contig[mid] && start < r->start[mid])) high = mid - 1;
        else if (r->contig[mid] < contig || (r->contig[mid] == contig && r->end[mid] < end)) low = mid + 1;
        else if (r->contig[mid] == contig && r->start[mid] <= start && end <= r->end[mid]) return 1;
        else break;
    }
    return 0;
}

typedef struct { uint64_t n_pairs_total, n_rand_total, n_attempt_fail; } stats_t;

static int core(opt_t *o, rng_t *r, const char *fn_fa, outs_t *out, stats_t *st)
{
    FILE *fp = fopen(fn_fa, "r");
    if (!fp) { fprintf(stderr, "[oracle] fail to open file '%s'. Abort!\n", fn_fa); return 1; }
    char fn_fai[4096]; snprintf(fn_fai, sizeof fn_fai, "%s.fai", fn_fa);
    FILE *fai = fopen(fn_fai, "r");
    seq_t seq = { 0, 0, NULL };
    char name[4096];
    uint64_t tot_len = 0, ctr = 0, rand_ii = 0; int n_ref = 0; int64_t l, n_sim = 0;
    ctab_t ct = { NULL, NULL, 0 }; mutin_t mi; memset(&mi, 0, sizeof mi); mi.type = o->muts_type;
    regions_t rg = { NULL, NULL, NULL, 0 }; int have_rg = o->fn_regions != NULL;
    int lmax = o->length[0] > o->length[1] ? o->length[0] : o->length[1];
    flowbuf_t tb[2];
    for (int j = 0; j < 2; ++j) flow_alloc(&tb[j], lmax, o->flow_order_len);
    int qcap = lmax; char *qstr = calloc((size_t)qcap + 1, 1);
    int site_cap = lmax > 0 ? lmax : 1; uint32_t *err_site = calloc((size_t)site_cap, sizeof(uint32_t));      /* error sites of a read end (base_error_sites) */
    int size[2] = { o->length[0], o->length[1] };

    /* pass 1: contig lengths and the VCF header, dwgsim.c:465-492 */
    if (out->has_mut) sink_puts(&out->vcf, "##fileformat=VCFv4.1\n");
    if (fai) {
        int ll, d0, d1, d2;
        while (0 < fscanf(fai, "%s\t%d\t%d\t%d\t%d", name, &ll, &d0, &d1, &d2)) {
            tot_len += (uint64_t)ll; ++n_ref;
            ct.name = realloc(ct.name, sizeof(char *) * (size_t)(ct.n + 1)); ct.len = realloc(ct.len, sizeof(int64_t) * (size_t)(ct.n + 1)); ct.name[ct.n] = strdup(name); ct.len[ct.n++] = ll;
            if (out->has_mut) sink_printf(&out->vcf, "##contig=<ID=%s,length=%d>\n", name, ll);
        }
        fclose(fai);
    } else {
        while ((l = fasta_next(fp, &seq, name)) >= 0) {
            tot_len += (uint64_t)l; ++n_ref;
            ct.name = realloc(ct.name, sizeof(char *) * (size_t)(ct.n + 1)); ct.len = realloc(ct.len, sizeof(int64_t) * (size_t)(ct.n + 1)); ct.name[ct.n] = strdup(name); ct.len[ct.n++] = l;
            if (out->has_mut) sink_printf(&out->vcf, "##contig=<ID=%s,length=%d>\n", name, (int)l);
        }
    }
    rewind(fp);
    if (out->has_mut) sink_puts(&out->vcf,
        "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n"
        "##INFO=<ID=pl,Number=1,Type=Integer,Description=\"Phasing: 1 - HET contig 1, #2 - HET contig #2, 3 - HOM both contigs\">\n"
        "##INFO=<ID=mt,Number=1,Type=String,Description=\"Variant Type: SUBSTITUTE/INSERT/DELETE\">\n"
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");

    if (o->muts_type >= 0) {   /* dwgsim.c:494-497 */
        FILE *fm = fopen(o->fn_muts, "r");
        if (!fm) { fprintf(stderr, "[oracle] fail to open file '%s'. Abort!\n", o->fn_muts); return 1; }
        if (o->muts_type == 1) parse_txt(fm, &ct, &mi); else if (o->muts_type == 0) parse_bed(fm, &ct, &mi); else parse_vcf(fm, &ct, &mi);
        fclose(fm);
    }
    if (have_rg) {   /* dwgsim.c:499-506 */
        FILE *fr = fopen(o->fn_regions, "r");
        if (!fr) { fprintf(stderr, "[oracle] fail to open file '%s'. Abort!\n", o->fn_regions); return 1; }
        parse_regions(fr, &ct, &rg); fclose(fr);
        tot_len = 0;
        for (int i = 0; i < rg.n; ++i) tot_len += rg.end[i] - rg.start[i];
    }
    uint32_t contig_i = 0;
    if (o->as_contig >= 0) {
        if (r->mode != RNG_PHILOX) { fprintf(stderr, "oracle: --as-contig needs --rng philox\n"); return 1; }
        contig_i = (uint32_t)o->as_contig; tot_len = (uint64_t)o->as_tot; n_ref += (int)o->as_after; n_sim = o->as_nsim;
    }
    if (o->range_rand_base >= 0) {
        if (r->mode != RNG_PHILOX || o->emit_count < 0) { fprintf(stderr, "oracle: --range-rand-base needs --rng philox and --emit-range\n"); return 1; }
        rand_ii = (uint64_t)o->range_rand_base;
    }
    while ((l = fasta_next(fp, &seq, name)) >= 0) {
        int64_t n_pairs = 0;
        n_ref--;
        if (o->output_type != 2) {
            if (0 == n_ref && o->C < 0) n_pairs = o->N - n_sim;               /* dwgsim.c:535-537 (NB: no region bookkeeping on this path) */
            else if (have_rg) {                                                  /* :539-581 */
                int64_t m = 0;
                for (int i = 0; i < rg.n; ++i) if (contig_i == rg.contig[i]) m += rg.end[i] - rg.start[i];
                if (0 == m) { contig_i++; continue; }                            /* #0 */
                l = m;
                int64_t num_n = 0;
                for (int i = 0; i < rg.n; ++i) if (contig_i == rg.contig[i])
                    for (int64_t q = rg.start[i]; q <= (int64_t)rg.end[i]; ++q) {   /* 1-based inclusive in the reference (App. B.10); s[-1] is out of bounds there: counted as non-ACGT */
                        int ch = q >= 1 ? seq.s[q - 1] : 'N';
                        if (nt4(ch) >= 4) num_n++;
                    }
                if (0.95 < num_n / (double)l) { contig_i++; continue; }           /* #1 */
            }
            if (0 == n_ref && o->C < 0) { }
            else if (0 < o->N) {                                                 /* :582-586 */
                n_pairs = (int64_t)(uint64_t)((long double)l / tot_len * o->N + 0.5);
                if (o->N - n_sim < n_pairs) n_pairs = o->N - n_sim;
            } else                                                               /* :589 */
                n_pairs = (int64_t)(uint64_t)(l * o->C / ((long double)(size[0] + size[1])) / (1.0 - o->rand_read) + 0.5);
            /* skip rules #2-#5, :595-623 */
            if (o->amplicons == 1) { if (l < lmax) { contig_i++; continue; } }
            else if (0 < o->length[1] && l < o->dist + 3 * o->std_dev) { contig_i++; continue; }
            else if (l < o->length[0] || (0 < o->length[1] && l < o->length[1])) { contig_i++; continue; }
            else if (n_pairs < 0) continue;
        }
        hap_t hap[2]; memset(hap, 0, sizeof hap);
        r->k1 = contig_i;
        if (o->muts_type >= 0) apply_mutation_input(o, r, &seq, &hap[0], &hap[1], contig_i, &mi);   /* mut.c:644-745 */
        else walk_contig(o, r, &seq, &hap[0], &hap[1]);      /* dwgsim.c:628-629 -> mut.c:591 */
        mut_debug_or_abort(name, &seq, &hap[0], &hap[1]);    /* mut.c:753 */
        left_justify(&seq, &hap[0], &hap[1]);                /* mut.c:756 */
        mut_debug_or_abort(name, &seq, &hap[0], &hap[1]);    /* mut.c:757 */
        if (out->has_mut) print_mutations(name, &seq, &hap[0], &hap[1], &out->txt, &out->vcf);

        if (o->output_type != 2) {
            int num_failed = 0; uint32_t att = 0;
            uint64_t ii_first = 0, ii_end = (uint64_t)n_pairs;
            if (o->range_rand_base >= 0) {      /* jump to the window */
                ii_first = (uint64_t)o->emit_first < ii_end ? (uint64_t)o->emit_first : ii_end;
                if ((uint64_t)(o->emit_first + o->emit_count) < ii_end) ii_end = (uint64_t)(o->emit_first + o->emit_count);
                n_sim += (int64_t)ii_first;
            }
            for (uint64_t ii = ii_first; ii != ii_end; ++ii, ++ctr) {
                int s[2] = { size[0], size[1] }, strand[2] = { 0, 0 }, d = 0, pos = 0;
                readinfo_t ri[2]; memset(ri, 0, sizeof ri);
                int n_err[2] = { 0, 0 }, n_err_first[2] = { 0, 0 };
                if (o->rand_read < rng_u(r, D_PAIR, ii, att, 0, 0)) {
                    if (o->amplicons == 1) { pos = 0; d = (int)seq.l; }
                    else {
                        uint32_t t = 0;
                        do { /* dwgsim.c:655-675 */
                            if (0 < s[1]) {
                                nstream_t ns = { D_PLACE_NORM, ii, att, t, 0, 0, 0.0, 0 };
                                double ran = ran_normal(r, &ns);
                                ran = ran * o->std_dev + o->dist;
                                d = (int)(ran + 0.5);
                                int min_dist = s[0] + s[1];
                                if (d < min_dist) d = min_dist;
                                if (d > l) d = (int)l;
                            } else d = 0;
                            int64_t range = (int64_t)l - d + 1;
                            pos = (int)(range * rng_u(r, D_PLACE, ii, att, 0, t));
                            if (have_rg) for (int i = 0; i < rg.n; ++i) if (contig_i == rg.contig[i]) {   /* dwgsim.c:696-707 */
                                int j = (int)(rg.end[i] - rg.start[i]);
                                if (pos < j) { pos = (int)rg.start[i] + pos - 1; break; }
                                else pos -= j;
                            }
                            t++;
                        } while (pos < 0 || pos >= seq.l || pos + d - 1 >= seq.l
                                 || (0 < s[1] && 0 == o->is_inner && ((0 < s[0] && d <= s[1]) || (d <= s[0] && 0 < s[1])))
                                 || (have_rg && 0 == regions_query(&rg, contig_i, (uint32_t)pos, (uint32_t)(pos + d))));
                    }
                    hap_t *cur = &hap[rng_u(r, D_PAIR, ii, att, 0, 1) < o->mut_freq ? 0 : 1];  /* :716 */
                    switch (o->read_one_strand) {                                               /* :722-727 */
                    case 0: strand[0] = (rng_u(r, D_PAIR, ii, att, 0, 2) < 0.5) ? 1 : 0; break;
                    case 1: strand[0] = 0; break;
                    default: strand[0] = 1; break;
                    }
                    switch (o->strandedness) {                                                  /* :730-742 */
                    case 0: strand[1] = (o->data_type == ILLUMINA) ? 1 - strand[0] : strand[0]; break;
                    case 1: strand[1] = strand[0]; break;
                    default: strand[1] = 1 - strand[0]; break;
                    }
                    /* read geometry, dwgsim.c:745-821 */
                    int64_t sl = seq.l, st0, st1 = 0; int step0, step1 = 1;
                    if (0 < s[1]) {
                        int64_t far_outer = pos + (int64_t)d - 1;
                        if (strand[0] == strand[1]) {
                            if (0 == strand[0]) { st0 = o->amplicons ? sl - 1 : (o->is_inner ? (int64_t)pos + s[1] + d - 1 : (int64_t)pos + d - s[0]); step0 = 1; st1 = pos; step1 = 1; }
                            else { st0 = (int64_t)pos + s[0] - 1; step0 = -1; st1 = o->amplicons ? sl - 1 : (o->is_inner ? (int64_t)pos + s[0] + d + s[1] - 1 : far_outer); step1 = -1; }
                        } else {
                            if (0 == strand[0]) { st0 = pos; step0 = 1; st1 = o->amplicons ? sl - 1 : (o->is_inner ? (int64_t)pos + s[0] + d + s[1] - 1 : far_outer); step1 = -1; }
                            else { st0 = o->amplicons ? sl - 1 : (o->is_inner ? (int64_t)pos + s[1] + d + s[0] - 1 : far_outer); step0 = -1; st1 = pos; step1 = 1; }
                        }
                        gen_read(cur, sl, st0, step0, s[0], strand[0], tb[0].seq, &ri[0]);
                        gen_read(cur, sl, st1, step1, s[1], strand[1], tb[1].seq, &ri[1]);
                    } else {
                        if (0 == strand[0]) { st0 = pos; step0 = 1; }
                        else if (o->amplicons == 1) { st0 = sl - 1; step0 = -1; }
                        else { st0 = (int64_t)pos + s[0] - 1; step0 = -1; }
                        gen_read(cur, sl, st0, step0, s[0], strand[0], tb[0].seq, &ri[0]);
                        ri[1].ext_coor = 0; /* dwgsim.c:643 ext_coor[2]={0,0}: never touched for single-end */
                    }
                    int num_n[2] = { 0, 0 };
                    for (int j = 0; j < 2; ++j) for (int i = 0; i < s[j]; ++i) if (tb[j].seq[i] == 4) num_n[j]++;
                    if (ri[0].ext_coor < 0 || ri[1].ext_coor < 0 || o->max_n < num_n[0] || o->max_n < num_n[1]) { /* :833-842 */
                        --ii; --ctr; num_failed++; att++; st->n_attempt_fail++;
                        if (num_failed > 10000) { fprintf(stderr, "\r[dwgsim_core] failed to generate a read after %d trials\n", num_failed); return 1; }
                        continue;
                    }
                    num_failed = 0;
                    if (o->emit_count >= 0 && (r->mode != RNG_PHILOX || (int64_t)ii < o->emit_first || (int64_t)ii >= o->emit_first + o->emit_count)) {
                        if (r->mode != RNG_PHILOX) { fprintf(stderr, "oracle: --emit-range needs --rng philox\n"); return 1; }
                        att = 0; n_sim++; continue;     /* outside the window: nothing is emitted, no later state depends on it */
                    }
                    if (SOLID == o->data_type) for (int j = 0; j < 2; ++j) if (0 < s[j]) to_colors(tb[j].seq, s[j]);
                    if (IONTORRENT == o->data_type) { /* :861-864 */
                        for (int j = 0; j < 2; ++j) {
                            uint32_t slot = 0;
                            /* the reference's flow mask persists between reads (dwgsim.c:430, :450-451); a read visits every flow in
                             * pass 1, so nothing stale survives -- mode B makes that explicit (per-read mask) to stay order-independent */
                            if (r->mode == RNG_PHILOX) memset(tb[j].mask, 0, (size_t)tb[j].mem);
                            s[j] = flow_errors(o, r, D_FLOW0 + (uint32_t)j, ii, att, &slot, &tb[j], s[j], strand[j], o->e[j].start, &n_err[j]);
                        }
                    } else for (int j = 0; j < 2; ++j) if (0 < s[j]) { /* :233-244, :866-881 */
                        int i = strand[j] ? s[j] - 1 : 0, step = strand[j] ? -1 : 1;
                        if (site_cap < s[j]) { site_cap = s[j]; err_site = realloc(err_site, sizeof(uint32_t) * (size_t)site_cap); }
                        base_error_sites(r, j, ii, att, s[j], o->e[j].start, o->e[j].by, err_site);
                        for (; 0 <= i && i < s[j]; i += step) {
                            uint8_t c = tb[j].seq[i];
                            if (c >= 4) c = 4;
                            else if (base_errs(r, err_site, i, o->e[j].start + o->e[j].by * i)) {
                                c = (uint8_t)((c + (uint64_t)(rng_u32(r, D_SUB0 + (uint32_t)j, ii, att, 0, r->mode == RNG_PHILOX ? err_site[i] - 1u : (uint32_t)i) * 3.0 + 1)) & 3);      /* mode B: the substitution draw of error SITE m (word m & 3 of block m >> 2): one Philox block per four errors, drawn with the chain */
                                ++n_err[j];
                                if (0 == i) ++n_err_first[j];
                            }
                            tb[j].seq[i] = c;
                        }
                    }
                    int cnt[6] = { n_err[0], ri[0].n_sub, ri[0].n_indel, n_err[1], ri[1].n_sub, ri[1].n_indel };
                    int cnt2[6] = { n_err[0] - n_err_first[0], ri[0].n_sub - ri[0].n_sub_first, ri[0].n_indel - ri[0].n_indel_first,
                                    n_err[1] - n_err_first[1], ri[1].n_sub - ri[1].n_sub_first, ri[1].n_indel - ri[1].n_indel_first };
                    for (int j = 0; j < 2; ++j) { /* :885-981 */
                        if (s[j] <= 0) continue;
                        if (qcap < s[j]) { qcap = s[j]; qstr = realloc(qstr, (size_t)qcap + 1); }
                        make_quals(o, r, j, ii, att, s[j], qstr);
                        emit_read(o, out, j, name, (unsigned)(ri[0].ext_coor + 1), (unsigned)(ri[1].ext_coor + 1), (unsigned)strand[0], (unsigned)strand[1],
                                  0, 0, cnt, cnt2, ii, tb[j].seq, s[j], qstr);
                    }
                } else { /* random read, dwgsim.c:983-1097 */
                    static const int zero6[6] = { 0, 0, 0, 0, 0, 0 };
                    const int in_win = o->emit_count < 0 || ((int64_t)ii >= o->emit_first && (int64_t)ii < o->emit_first + o->emit_count);
                    for (int j = 0; j < 2 && in_win; ++j) {
                        if (s[j] <= 0) continue;
                        for (int i = 0; i < s[j]; ++i) tb[j].seq[i] = random_base(r, j, ii, att, (uint32_t)i);
                        make_quals(o, r, j, ii, att, s[j], qstr);
                        if (SOLID == o->data_type) to_colors(tb[j].seq, s[j]);
                        emit_read(o, out, j, "rand", 0, 0, 0, 0, 1, 1, zero6, zero6, rand_ii, tb[j].seq, s[j], qstr);
                    }
                    rand_ii++;
                }
                att = 0;
                n_sim++;
            }
        }
        hap_free(&hap[0]); hap_free(&hap[1]);
        contig_i++;
    }
    st->n_pairs_total = ctr; st->n_rand_total = rand_ii;
    fclose(fp);
    free(seq.s); free(qstr);
    for (int j = 0; j < 2; ++j) { free(tb[j].seq); free(tb[j].mask); }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Entry point: dwgsim_oracle [--rng drand48|philox] [--log det|libm] [--null-fastq] [--verbose]
 *                            <dwgsim options> <in.ref.fa> <out.prefix>
 * Writes <prefix>.mutations.{txt,vcf} and UNCOMPRESSED <prefix>.{bfast,bwa.read1,bwa.read2}.fastq
 * (the reference's own test compares decompressed bytes, testdata/test.sh:21-26).
 * ---------------------------------------------------------------------------------------- */
static FILE *open_out(const char *prefix, const char *suffix)
{
    char fn[4096]; snprintf(fn, sizeof fn, "%s.%s", prefix, suffix);
    FILE *f = fopen(fn, "w");
    if (!f) { fprintf(stderr, "[oracle] fail to open file '%s'. Abort!\n", fn); exit(1); }
    return f;
}

int oracle_main(int argc, char **argv)
{
    opt_t o; rng_t r; opt_defaults(&o); o.emit_first = 0; o.emit_count = -1; o.as_contig = -1; o.as_tot = o.as_after = o.as_nsim = 0; o.range_rand_base = -1;
    /* strip the oracle's own long options */
    char **av = malloc(sizeof(char *) * (size_t)(argc + 1)); int ac = 0;
    for (int i = 0; i < argc; ++i) {
        if (!strcmp(argv[i], "--rng") && i + 1 < argc) { ++i; o.rng_mode = !strcmp(argv[i], "philox") ? RNG_PHILOX : RNG_DRAND48; }
        else if (!strcmp(argv[i], "--log") && i + 1 < argc) { ++i; o.use_libm_log = !strcmp(argv[i], "libm"); }
        else if (!strcmp(argv[i], "--null-fastq")) o.null_fastq = 1;
        else if (!strcmp(argv[i], "--dump-draws") && i + 1 < argc) { ++i; o.dump_path = argv[i]; }
        else if (!strcmp(argv[i], "--verbose")) o.verbose = 1;
        else if (!strcmp(argv[i], "--emit-range") && i + 1 < argc) { ++i; long long a = 0, b = 0; sscanf(argv[i], "%lld:%lld", &a, &b); o.emit_first = a; o.emit_count = b; }
        else if (!strcmp(argv[i], "--as-contig") && i + 1 < argc) { ++i; long long a = 0, b = 0, c = 0, d = 0; sscanf(argv[i], "%lld,%lld,%lld,%lld", &a, &b, &c, &d); o.as_contig = a; o.as_tot = b; o.as_after = c; o.as_nsim = d; }
        else if (!strcmp(argv[i], "--range-rand-base") && i + 1 < argc) { ++i; o.range_rand_base = atoll(argv[i]); }
        else av[ac++] = argv[i];
    }
    av[ac] = NULL;
    int first = 0;
    if (!opt_parse(&o, &r, ac, av, &first)) { fprintf(stderr, "usage: dwgsim_oracle [--rng drand48|philox] [--log det|libm] [--null-fastq] [--dump-draws FILE] [dwgsim options] <in.ref.fa> <out.prefix>\n"); free(av); return 1; }
    const char *fa = av[first], *prefix = av[first + 1];
    outs_t out; memset(&out, 0, sizeof out);
    out.has_mut = o.output_type != 1;
    if (o.output_type != 2) { out.has_bfast = o.reads_output_type != 1; out.has_bwa = o.reads_output_type != 2; }
    if (out.has_mut) { out.txt.fp = open_out(prefix, "mutations.txt"); out.vcf.fp = open_out(prefix, "mutations.vcf"); }
    /* NB: the reference with -M 1 dereferences a NULL fp_vcf; the oracle simply writes no mutation files */
    if (o.null_fastq) { out.bfast.discard = out.bwa1.discard = out.bwa2.discard = 1; }
    else {
        if (out.has_bfast) out.bfast.fp = open_out(prefix, "bfast.fastq");
        if (out.has_bwa) { out.bwa1.fp = open_out(prefix, "bwa.read1.fastq"); out.bwa2.fp = open_out(prefix, "bwa.read2.fastq"); }
    }
    stats_t st; memset(&st, 0, sizeof st);
    int rc = core(&o, &r, fa, &out, &st);
    sink_t *all[5] = { &out.bfast, &out.bwa1, &out.bwa2, &out.txt, &out.vcf };
    for (int i = 0; i < 5; ++i) { sink_flush(all[i]); if (all[i]->fp) fclose(all[i]->fp); free(all[i]->p); }
    if (o.verbose)
        fprintf(stderr, "[oracle] rng=%s pairs=%llu random=%llu failed_attempts=%llu uniforms=%llu log_mismatch=%llu fastq_bytes=%llu\n",
                o.rng_mode ? "philox" : "drand48", (unsigned long long)st.n_pairs_total, (unsigned long long)st.n_rand_total,
                (unsigned long long)st.n_attempt_fail, (unsigned long long)r.n_draws, (unsigned long long)r.n_log_mismatch,
                (unsigned long long)(out.bfast.total + out.bwa1.total + out.bwa2.total));
    if (r.dump) fclose(r.dump);
    free(o.flow_order); free(o.read_prefix); free(o.fixed_quality); free(av);
    return rc;
}

#ifdef ORACLE_MAIN
int main(int argc, char **argv) { return oracle_main(argc, argv); }
#endif
