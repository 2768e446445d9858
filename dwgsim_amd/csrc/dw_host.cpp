// dw_host.cpp -- C-ABI (include/dwgsim_hip.h) over the HIP kernels of dw_walk.hip / dw_simulate.hip.
//
// Host responsibilities, mirroring what dwgsim_core() does around its two hot loops:
//   * option defaults / checks / error-ramp tables      (dwgsim_opt.c:40-80, :307-371, :459-460)
//   * contig scheduling arithmetic                       (dwgsim.c:535-537, :582-590, :595-618)
//   * device residency of contigs and mutated haplotypes (replaces seq_t / mutseq_t, mut.h:12-47)
//   * mutations.txt / .vcf text from the sparse list of mutated cells (mut.c:781-893)
// There is no CPU implementation of the hot path here: without a HIP device create() fails.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <stdarg.h>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/dwgsim_hip.h"
#include "dw_kernels.hpp"
#include "dw_launch.hpp"
#include "dw_mutin.hpp"

using namespace dw;

namespace {

struct Contig {
    std::string name;
    std::vector<uint8_t> ascii;        // host copy (reference bases for the txt/vcf writer)
    int64_t l = 0;
    uint32_t contig_index = 0;
    bool alive = false, mutated = false;
    uint8_t *d_ref = nullptr, *d_cells[2] = {nullptr, nullptr}, *d_view[2] = {nullptr, nullptr};      // reference codes, byte cells, 4-bit read views
    int32_t *d_ins_pos[2] = {nullptr, nullptr};
    uint32_t *d_ins_len[2] = {nullptr, nullptr}, *d_ins_off[2] = {nullptr, nullptr};
    uint8_t *d_ins_bases[2] = {nullptr, nullptr};
    uint32_t n_ins[2] = {0, 0}, n_ins_bases[2] = {0, 0};
    size_t cap_ins[2] = {0, 0}, cap_bases[2] = {0, 0};
    uint8_t *d_name_fixed = nullptr; int32_t name_fixed_len = 0;
    uint32_t n_cand = 0;
    uint16_t *d_summ[2] = {nullptr, nullptr}; bool summ_valid = false;      // haplotype summaries for count_random (built on demand)
    int64_t l_place = 0;                // fragment-placement length (region length with -x)
    int32_t *d_reg = nullptr; int32_t n_reg = 0;   // -x: [start[0..n), end[0..n)] of this contig
};

struct DevBuf {                     // grow-only device buffer
    void *p = nullptr; size_t cap = 0;
};

constexpr int N_COUNTERS = 32;      // u64 words of a counter block (SimArgs::counters)

struct Slot {                       // one of the two batches a context can have in flight
    uint64_t *d_counters = nullptr, *h_counters = nullptr;      // device block + pinned mirror
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_done = nullptr, ev_fetched = nullptr;
    bool pending = false, empty = true, fetch_in_flight = false;
    uint64_t n_pairs = 0, out_bytes[3] = {0, 0, 0}, gz_bytes[3] = {0, 0, 0};
    DevBuf gz_out[3], gz_status;        // GPU gzip: the members of each stream, look-back words
};

} // namespace

struct dwgsim_hip_ctx {
    dwgsim_hip_params_t prm;
    std::string read_prefix;
    std::vector<uint8_t> flow;            // Ion Torrent flow order as base codes
    uint8_t *d_flow = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    double e_by[2] = {0, 0};
    uint64_t *d_thr[2] = {nullptr, nullptr};
    uint32_t *d_thr32[2] = {nullptr, nullptr}; int e_full = 0;
    uint32_t *d_qbase[2] = {nullptr, nullptr}; int32_t qb_words = 1;
    uint8_t *d_rand_fixed = nullptr; int32_t rand_fixed_len = 0;
    std::vector<Contig> contigs;
    // simulate() working set
    DevBuf meta, fail_summ, block_rand, status_all, out[2][3], scratch_mask, scratch_cnt;
    DevBuf w_cand, w_ev, w_flags, w_lo, w_sufmin, w_bound;     // mutation-walk scratch (grow-only)
    bool seq_justify = false;
    MutInput mutin; bool has_mutin = false;                             // -m / -b / -v
    Regions regions; bool has_regions = false;                           // -x
    DevBuf w_ppos, w_pcells, flow_scratch;
    uint64_t *d_counters = nullptr;          // N_COUNTERS x u64: walk / calibrate / count_random (synchronous calls)
    uint64_t *h_counters = nullptr;          // pinned mirror
    Slot slot[2];                            // simulate(): two batches in flight (kernels of one overlap the copy-out of the other)
    hipStream_t copy_stream = nullptr;       // device -> host copies of finished text
    uint64_t *d_chain = nullptr;             // [0] random reads emitted before the next batch, [1] the abort rule's carry: handed from batch to batch on the device
    int chain_contig = -1; uint64_t chain_next_ii = 0;      // which (contig, read index) the carry continues
    bool has_carry_override = false; uint64_t carry_override = 0;
    int64_t walk_cap = -1; bool phases = false; int writer = -1, force_threads = 0;      // dwgsim_hip_debug_option
    bool gzip_on = false; uint32_t *d_crc_table = nullptr, *d_crc_shift = nullptr;      // dwgsim_hip_set_gzip
    void *h_stage = nullptr; size_t h_stage_cap = 0;   // pinned staging for fetch
    std::string txt, vcf;
};

namespace {

#define HIPC(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char b_[512]; snprintf(b_, sizeof b_, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); (ctx)->err = b_; return DWGSIM_HIP_ERR_DEVICE; } } while (0)

int ensure(dwgsim_hip_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) HIPC(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    HIPC(c, hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

uint8_t nt4(int ch)      // dwgsim.c:56-73
{
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

// Parameters of the lazy quality normals (error budget: dw_simulate.hip quality_try_lazy).  Polar radii with L' = -log2(r) < lmin give
// |nrm| < sqrt(2 ln 2 lmin): lmin is the largest value (<= 2^-4) for which that keeps |nrm * sigma| below 0.45, so the offset is 0 there
// without further work; when sigma is too large for that (lmin would fall below 2^-10) the exact path handles those radii instead.
void lazy_quality_params(double sigma, float *k, float *eps, float *lmin, int32_t *near1_zero)
{
    const double two_ln2 = 2.0 * log(2.0);
    double lm = 0.98 * (0.45 / sigma) * (0.45 / sigma) / two_ln2;
    if (!(lm < 0x1p-4)) lm = 0x1p-4;
    *near1_zero = lm >= 0x1p-10 ? 1 : 0;
    if (lm < 0x1p-10) lm = 0x1p-10;
    *lmin = (float)lm;
    lm = (double)*lmin * (1.0 - 0x1p-20);         // (what the budget may assume after the rounding to float)
    *k = (float)(sqrt(two_ln2) * 0x1p-31 * sigma);
    *eps = (float)(1.5 * sigma * (3.4e-7 / sqrt(lm) + 7.2e-6) + 0x1p-18);       // (+inf for an absurd -Q: every value then takes the exact path)
}

WalkParams walk_params(const dwgsim_hip_ctx *c)
{
    WalkParams w; w.mut_rate = c->prm.mut_rate; w.indel_frac = c->prm.indel_frac; w.indel_extend = c->prm.indel_extend;
    w.indel_min = c->prm.indel_min; w.is_hap = c->prm.is_hap; w.seed = (uint32_t)c->prm.seed;
    w.mut_thr = !(c->prm.mut_rate > 0) ? 0 : c->prm.mut_rate >= 1.0 ? 0x100000000ull : (uint64_t)ceil(c->prm.mut_rate * 4294967296.0);   // exact scaling by 2^32
    return w;
}

ContigDev contig_dev(const Contig &k)
{
    ContigDev d;
    for (int h = 0; h < 2; ++h) {
        d.hap[h].cells = k.d_cells[h]; d.hap[h].view = k.d_view[h]; d.hap[h].ins_pos = k.d_ins_pos[h]; d.hap[h].ins_len = k.d_ins_len[h];
        d.hap[h].ins_off = k.d_ins_off[h]; d.hap[h].ins_bases = k.d_ins_bases[h]; d.hap[h].n_ins = k.n_ins[h];
    }
    d.ref = k.d_ref; d.l = k.l; d.contig_index = k.contig_index;
    d.tot4 = nullptr; d.cap_bases[0] = d.cap_bases[1] = 0;
    return d;
}

void free_contig(Contig &k)
{
    hipFree(k.d_ref);
    for (int h = 0; h < 2; ++h) { hipFree(k.d_cells[h]); hipFree(k.d_view[h]); hipFree(k.d_ins_pos[h]); hipFree(k.d_ins_len[h]); hipFree(k.d_ins_off[h]); hipFree(k.d_ins_bases[h]); }
    hipFree(k.d_name_fixed); hipFree(k.d_reg); hipFree(k.d_summ[0]); hipFree(k.d_summ[1]);
    k = Contig();
}

} // namespace

extern "C" {

void dwgsim_hip_params_default(dwgsim_hip_params_t *p)
{
    memset(p, 0, sizeof(*p));
    p->e_start[0] = p->e_end[0] = p->e_start[1] = p->e_end[1] = 0.02;
    p->dist = 500; p->std_dev = 50; p->N = -1; p->C = 100;
    p->length[0] = p->length[1] = 70;
    p->mut_rate = 0.001; p->mut_freq = 0.5; p->indel_frac = 0.1; p->indel_extend = 0.3; p->indel_min = 1;
    p->rand_read = 0.05; p->seed = -1; p->fixed_quality = -1; p->quality_std = 2.0;
}

#define CHK(v, lo, hi, nm) do { if ((v) < (lo) || (hi) < (v)) { if (msg) snprintf(msg, cap, "Error: command line option %s was out of range\n", nm); return DWGSIM_HIP_ERR_ARG; } } while (0)

int dwgsim_hip_params_check(const dwgsim_hip_params_t *p, char *msg, size_t cap)
{
    if (msg && cap) msg[0] = 0;
    CHK(p->is_inner, 0, 1, "-i"); CHK(p->dist, 0, INT32_MAX, "-d"); CHK(p->std_dev, 0, INT32_MAX, "-s");
    if (p->N < 0 && p->C < 0) { if (msg) snprintf(msg, cap, "Must use one of -N or -C"); return DWGSIM_HIP_ERR_ARG; }
    else if (0 < p->N && 0 < p->C) { if (msg) snprintf(msg, cap, "Cannot use both -N or -C"); return DWGSIM_HIP_ERR_ARG; }
    else if (0 < p->N) { CHK(p->N, 1, INT32_MAX, "-N"); CHK(p->C, INT32_MIN, -1, "-C"); }
    else { CHK(p->N, INT32_MIN, -1, "-N"); CHK(p->C, 0, INT32_MAX, "-C"); }
    CHK(p->length[0], 1, INT32_MAX, "-1"); CHK(p->length[1], 0, INT32_MAX, "-2");
    for (int i = 0; i < 2; ++i) {
        if (p->e_start[i] < 0.0 || 1.0 < p->e_start[i]) { if (msg) snprintf(msg, cap, "End %s: the start error is out of range (-e)\n", i ? "two" : "one"); return DWGSIM_HIP_ERR_ARG; }
        if (p->e_end[i] < 0.0 || 1.0 < p->e_end[i]) { if (msg) snprintf(msg, cap, "End %s: the end error is out of range (-e)\n", i ? "two" : "one"); return DWGSIM_HIP_ERR_ARG; }
    }
    CHK(p->mut_rate, 0, 1.0, "-r"); CHK(p->indel_frac, 0, 1.0, "-R"); CHK(p->indel_extend, 0, 1.0, "-X");
    CHK(p->indel_min, 1, INT32_MAX, "-I"); CHK(p->data_type, 0, 2, "-c"); CHK(p->strandedness, 0, 2, "-S");
    CHK(p->read_one_strand, 0, 2, "-A"); CHK(p->max_n, 0, INT32_MAX, "-n"); CHK(p->rand_read, 0, 1.0, "-y");
    CHK(p->use_base_error, 0, 1, "-B"); CHK(p->is_hap, 0, 1, "-H");
    CHK(p->quality_std, 0, INT32_MAX, "-Q"); CHK(p->reads_output_type, 0, 2, "-o"); CHK(p->output_type, 0, 2, "-M"); CHK(p->amplicons, 0, 1, "-a");
    if (p->data_type == 2 && !p->flow_order) { if (msg) snprintf(msg, cap, "Error: command line option -f is required\n"); return DWGSIM_HIP_ERR_ARG; }
    if (p->data_type == 2) {       // dwgsim_opt.c:338-343, :396-413
        for (int i = 0; i < 2; ++i) if (p->e_end[i] != p->e_start[i]) { if (msg) snprintf(msg, cap, "End %s: a uniform error rate must be given for Ion Torrent data\n", i ? "two" : "one"); return DWGSIM_HIP_ERR_ARG; }
        const size_t F = strlen(p->flow_order);
        bool has[4] = {false, false, false, false};
        for (size_t i = 0; i < F; ++i) { const uint8_t c = nt4((unsigned char)p->flow_order[i]); if (c < 4) has[c] = true; }
        bool only_acgt = true; for (size_t i = 0; i < F; ++i) if (nt4((unsigned char)p->flow_order[i]) >= 4) only_acgt = false;
        if (F == 0 || F > 64 || !only_acgt || !(has[0] && has[1] && has[2] && has[3])) { if (msg) snprintf(msg, cap, "dwgsim-hip: the flow order (-f) must hold 1..64 flows and contain all of A, C, G, T\n"); return DWGSIM_HIP_ERR_UNSUP; }
    }
    if (p->seed < 0) { if (msg) snprintf(msg, cap, "dwgsim-hip: the seed must be resolved (>= 0) before the context is created\n"); return DWGSIM_HIP_ERR_ARG; }
    return DWGSIM_HIP_OK;
}

int64_t dwgsim_hip_pairs_for_contig(const dwgsim_hip_params_t *p, int64_t l, uint64_t tot_len, int is_last_contig, int64_t n_sim_so_far)
{
    int64_t n_pairs = 0;
    const int size0 = p->length[0], size1 = p->length[1];
    if (is_last_contig && p->C < 0) n_pairs = p->N - n_sim_so_far;                                  // dwgsim.c:535-537
    else if (0 < p->N) {                                                                            // :582-586
        n_pairs = (int64_t)(uint64_t)((long double)l / tot_len * p->N + 0.5);
        if (p->N - n_sim_so_far < n_pairs) n_pairs = p->N - n_sim_so_far;
    } else n_pairs = (int64_t)(uint64_t)(l * p->C / ((long double)(size0 + size1)) / (1.0 - p->rand_read) + 0.5);   // :589
    const int max_len = size0 > size1 ? size0 : size1;
    if (p->amplicons == 1) { if (l < max_len) return -2; }                                          // #2 :596-603
    else if (0 < size1 && l < p->dist + 3 * p->std_dev) return -3;                                 // #3 :605-611
    else if (l < size0 || (0 < size1 && l < size1)) return -4;                                     // #4 :612-618
    return n_pairs < 0 ? -5 : n_pairs;
}

int dwgsim_hip_device_info(int device, char *name, size_t cap, int *n_cu, size_t *hbm_bytes)
{
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return DWGSIM_HIP_ERR_DEVICE;
    if (name && cap) snprintf(name, cap, "%s (%s)", pr.name, pr.gcnArchName);
    if (n_cu) *n_cu = pr.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = pr.totalGlobalMem;
    return DWGSIM_HIP_ABI_VERSION;
}

int dwgsim_hip_failseg_join(uint64_t acc[4], const uint64_t next[4])      // the same monoid as failseg_join in dw_simulate.hip
{
    const uint64_t aP = acc[0], aS = acc[1], aR = acc[2], aB = acc[3], bP = next[0], bS = next[1], bR = next[2], bB = next[3];
    acc[3] = (aB | bB | ((aR && aS + bP > (uint64_t)MAX_ATTEMPTS) ? 1u : 0u)) ? 1 : 0;
    acc[0] = aR ? aP : aP + bP;
    acc[1] = bR ? bS : aS + bS;
    acc[2] = (aR | bR) ? 1 : 0;
    return (acc[3] || acc[0] > (uint64_t)MAX_ATTEMPTS || acc[1] > (uint64_t)MAX_ATTEMPTS) ? 1 : 0;
}

void dwgsim_hip_shard_range(uint64_t n_pairs, int rank, int world, uint64_t *first, uint64_t *n)
{
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    const uint64_t base = n_pairs / (uint64_t)world, rem = n_pairs % (uint64_t)world, r = (uint64_t)rank;
    if (first) *first = r * base + (r < rem ? r : rem);
    if (n) *n = base + (r < rem ? 1 : 0);
}

// Ion Torrent: room for a read after the flow model.  Every empty flow (about three per base) inserts Geometric(e) bases, inserted bases
// are examined again: the mean growth is ~3 e / (1 - e) per base; four times that plus slack keeps overflow (reported as an error, never
// written out of bounds) out of reach for realistic error rates and far away even for e = 0.3.
static int flow_read_capacity(int len, double e)
{
    const double ec = !(e > 0) ? 0 : e > 0.9 ? 0.9 : e;       // (NaN, e.g. -B with -e 0: no flow errors at all)
    return len + 64 + (int)(len * 12.0 * ec / (1.0 - ec));
}

static int set_err(int *err, int v) { if (err) *err = v; return v; }

dwgsim_hip_ctx_t *dwgsim_hip_create(const dwgsim_hip_params_t *p, int device, int *err)
{
    char msg[512];
    int rc = dwgsim_hip_params_check(p, msg, sizeof msg);
    if (rc != DWGSIM_HIP_OK) { fprintf(stderr, "%s", msg); set_err(err, rc); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
        fprintf(stderr, "dwgsim-hip: no usable HIP device (requested %d of %d); the hot path has no CPU fallback\n", device, ndev);
        set_err(err, DWGSIM_HIP_ERR_DEVICE); return nullptr;
    }
    dwgsim_hip_ctx *c = new dwgsim_hip_ctx();
    c->prm = *p;
    if (p->read_prefix) c->read_prefix = p->read_prefix;
    if (p->data_type == 2) for (const char *q = p->flow_order; *q; ++q) c->flow.push_back(nt4((unsigned char)*q));
    c->prm.read_prefix = nullptr; c->prm.flow_order = nullptr;
    c->device = device;
    auto fail = [&](const char *what) { fprintf(stderr, "dwgsim-hip: %s: %s\n", what, c->err.c_str()); set_err(err, DWGSIM_HIP_ERR_DEVICE); dwgsim_hip_destroy(c); return (dwgsim_hip_ctx *)nullptr; };
    auto init = [&]() -> int {
        HIPC(c, hipSetDevice(device));
        HIPC(c, hipStreamCreate(&c->stream));
        HIPC(c, hipStreamCreate(&c->copy_stream));
        HIPC(c, hipMalloc((void **)&c->d_counters, N_COUNTERS * sizeof(uint64_t)));
        HIPC(c, hipHostMalloc((void **)&c->h_counters, N_COUNTERS * sizeof(uint64_t), hipHostMallocDefault));
        HIPC(c, hipMalloc((void **)&c->d_chain, 4 * sizeof(uint64_t)));
        HIPC(c, hipMemset(c->d_chain, 0, 4 * sizeof(uint64_t)));
        for (Slot &sl : c->slot) {
            HIPC(c, hipMalloc((void **)&sl.d_counters, N_COUNTERS * sizeof(uint64_t)));
            HIPC(c, hipHostMalloc((void **)&sl.h_counters, N_COUNTERS * sizeof(uint64_t), hipHostMallocDefault));
            HIPC(c, hipEventCreate(&sl.ev_k0)); HIPC(c, hipEventCreate(&sl.ev_k1)); HIPC(c, hipEventCreate(&sl.ev_done)); HIPC(c, hipEventCreate(&sl.ev_fetched));
        }
        { std::vector<uint8_t> fl(64, 4); for (size_t i = 0; i < c->flow.size() && i < 64; ++i) fl[i] = c->flow[i];
          HIPC(c, hipMalloc((void **)&c->d_flow, 64)); HIPC(c, hipMemcpy(c->d_flow, fl.data(), 64, hipMemcpyHostToDevice)); }
        // -B (dwgsim_opt.c:415-457): rescale the flow error so that the per-base error rate of 10^6 random reads matches -e
        if (c->prm.data_type == 2 && c->prm.use_base_error) {
            double sf = 0.0;
            for (int i = 0; i < 2; ++i) {
                const int len = c->prm.length[i];
                if (len <= 0) continue;
                fprintf(stderr, "[dwgsim_core] Updating error rate for end %d\n", i + 1);
                if (0 < i && len == c->prm.length[1 - i]) {
                    c->prm.e_start[i] = c->prm.e_start[1 - i]; c->prm.e_end[i] = c->prm.e_end[1 - i];
                    fprintf(stderr, "[dwgsim_core] Using scaling factor from previous end\n[dwgsim_core] Updated with scaling factor %.5lf\n", sf);
                    continue;
                }
                const double e = c->prm.e_start[i];
                CalibArgs ca;
                ca.seed = (uint32_t)c->prm.seed; ca.end = i; ca.len = len; ca.n_reads = 1000000;       // ERROR_RATE_NUM_RANDOM_READS, dwgsim_opt.h:5
                ca.thr = !(e > 0) ? 0 : e >= 1.0 ? 0x100000000ull : (uint64_t)ceil(e * 4294967296.0);
                ca.flow = c->d_flow; ca.flow_len = (int32_t)c->flow.size();
                ca.cap = flow_read_capacity(len, e); ca.lds_words = (ca.cap + 7) / 8;
                const size_t nblk = (size_t)((ca.n_reads + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK);
                if (ensure(c, c->flow_scratch, (size_t)flow_words_per_lane(ca.lds_words, ca.cap) * PAIRS_PER_BLOCK * nblk * sizeof(uint32_t))) return -1;
                ca.scratch = (uint32_t *)c->flow_scratch.p; ca.counters = c->d_counters;
                HIPC(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(uint64_t), c->stream));
                launch_calibrate(c->stream, ca);
                HIPC(c, hipMemcpyAsync(c->h_counters, c->d_counters, N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
                HIPC(c, hipStreamSynchronize(c->stream));
                if (c->h_counters[2]) { c->err = "-B calibration: a read outgrew its flow-space buffer"; return -1; }
                const int32_t n_err = (int32_t)c->h_counters[8], counts = (int32_t)c->h_counters[9];       // int32 accumulators as in the reference
                sf = e / (n_err / (1.0 * counts));
                c->prm.e_end[i] *= sf; c->prm.e_start[i] = c->prm.e_end[i];
                fprintf(stderr, "[dwgsim_core] Updated with scaling factor %.5lf!\n", sf);
            }
        }
        // per-position error thresholds and base qualities (dwgsim_opt.c:459-460, dwgsim.c:237, :906-910)
        for (int j = 0; j < 2; ++j) {
            const int n = c->prm.length[j];
            if (n <= 0) continue;
            c->e_by[j] = (c->prm.e_end[j] - c->prm.e_start[j]) / n;
            std::vector<uint64_t> thr((size_t)n); std::vector<int8_t> qb((size_t)n);
            for (int i = 0; i < n; ++i) {
                const double ei = c->prm.e_start[j] + c->e_by[j] * i;
                thr[(size_t)i] = !(ei > 0) ? 0 : ei >= 1.0 ? 0x100000000ull : (uint64_t)ceil(ei * 4294967296.0);   // u = w * 2^-32 < ei  <=>  w < ceil(ei * 2^32) (exact: scaling by 2^32 is exact)
                char q;
                if (ei > 0) q = (char)((int)(-10.0 * log(ei) / log(10.0) + 0.499) + '!'); else q = 40 + '!';
                qb[(size_t)i] = (int8_t)q;
            }
            HIPC(c, hipMalloc((void **)&c->d_thr[j], sizeof(uint64_t) * (size_t)n));
            {   // packed table: n entries, then the last one repeated (>= 4 copies), the same word count for both read ends
                const int lmax = c->prm.length[0] > c->prm.length[1] ? c->prm.length[0] : c->prm.length[1];
                c->qb_words = (lmax + 4 + 3) / 4 + 1;
                std::vector<int8_t> padded((size_t)c->qb_words * 4, qb[(size_t)n - 1]);
                memcpy(padded.data(), qb.data(), (size_t)n);
                HIPC(c, hipMalloc((void **)&c->d_qbase[j], padded.size()));
                HIPC(c, hipMemcpy(c->d_qbase[j], padded.data(), padded.size(), hipMemcpyHostToDevice));
            }
            HIPC(c, hipMemcpy(c->d_thr[j], thr.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice));
            std::vector<uint32_t> t32(((size_t)n + 7) / 8 * 8, 0u);
            for (int i = 0; i < n; ++i) {
                if (thr[(size_t)i] >= 0x100000000ull) { t32[(size_t)i] = 0xFFFFFFFFu; c->e_full = 1; }
                else t32[(size_t)i] = (uint32_t)thr[(size_t)i];
            }
            if (c->e_full) for (int i = 0; i < n; ++i) if (thr[(size_t)i] == 0xFFFFFFFFull) { c->err = "an error rate within 2^-32 of (but not equal to) 1 next to one equal to 1 is not representable"; return -1; }
            HIPC(c, hipMalloc((void **)&c->d_thr32[j], sizeof(uint32_t) * t32.size()));
            HIPC(c, hipMemcpy(c->d_thr32[j], t32.data(), sizeof(uint32_t) * t32.size(), hipMemcpyHostToDevice));
        }
        // device copy: '@' + "[prefix_]rand", zero padded to >= 256 + 16 bytes (the kernel stages 256 bytes in LDS)
        std::string rf = c->read_prefix.empty() ? std::string("rand") : c->read_prefix + "_rand";
        c->rand_fixed_len = (int32_t)rf.size();
        std::string rbuf = "@" + rf; rbuf.resize(rbuf.size() < 256 ? 272 : rbuf.size() + 16, '\0');
        HIPC(c, hipMalloc((void **)&c->d_rand_fixed, rbuf.size()));
        HIPC(c, hipMemcpy(c->d_rand_fixed, rbuf.data(), rbuf.size(), hipMemcpyHostToDevice));
        return 0;
    };
    if (init() != 0) return fail("context initialisation failed");
    set_err(err, DWGSIM_HIP_OK);
    return c;
}

// Not part of the drop-in ABI (include/dwgsim_hip.h): a device self-test used by tests/test_gpu_parity.py.  Compares the
// range-restricted fp64 division / sqrt / log of the quality path with the compiler's general forms on n operand sets;
// out[0..2] = bitwise differences (division, sqrt, log), out[3] = comparisons made.
extern "C" int dwgsim_hip_selftest_fp64(int device, uint32_t seed, uint64_t n, uint64_t *out)
{
    if (!out || hipSetDevice(device) != hipSuccess) return DWGSIM_HIP_ERR_DEVICE;
    uint64_t *d = nullptr;
    if (hipMalloc((void **)&d, 4 * sizeof(uint64_t)) != hipSuccess) return DWGSIM_HIP_ERR_NOMEM;
    hipMemset(d, 0, 4 * sizeof(uint64_t));
    launch_selftest_fp64(nullptr, seed, n, d);
    const hipError_t e = hipMemcpy(out, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? DWGSIM_HIP_OK : DWGSIM_HIP_ERR_DEVICE;
}

// Not part of the drop-in ABI either: self-test of the lazy quality normals (dw_simulate.hip k_selftest_lazy).  out[0..4] = counters of the
// comparison with the exact form on n random blocks at quality_std = sigma, out[5] = max |estimate - exact| / eps, out[6..8] = worst error of
// v_log_f32 / v_rcp_f32 / v_sqrt_f32 over EVERY float of their operand ranges, in units of the bounds the error budget assumes (doubles).
extern "C" int dwgsim_hip_selftest_lazy(int device, uint32_t seed, uint64_t n, double sigma, int exhaustive, uint64_t *out)
{
    if (!out || hipSetDevice(device) != hipSuccess) return DWGSIM_HIP_ERR_DEVICE;
    uint64_t *d = nullptr;
    if (hipMalloc((void **)&d, 12 * sizeof(uint64_t)) != hipSuccess) return DWGSIM_HIP_ERR_NOMEM;
    hipMemset(d, 0, 12 * sizeof(uint64_t));
    float qk, qeps, qlmin; int32_t qnear1;
    lazy_quality_params(sigma, &qk, &qeps, &qlmin, &qnear1);
    launch_selftest_lazy(nullptr, 0, seed, n, sigma, qk, qeps, qlmin, qnear1, d);
    if (exhaustive) {
        launch_selftest_lazy(nullptr, 1, 0, 0x3F800000u - 0x20800000u, 0, 0, 0, 0, 0, d);
        launch_selftest_lazy(nullptr, 2, 0, 0x3F800000u - 0x20800000u, 0, 0, 0, 0, 0, d);
        launch_selftest_lazy(nullptr, 3, 0, 0x62800000u - 0x3A000000u, 0, 0, 0, 0, 0, d);
    }
    const hipError_t e = hipMemcpy(out, d, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? DWGSIM_HIP_OK : DWGSIM_HIP_ERR_DEVICE;
}

void dwgsim_hip_destroy(dwgsim_hip_ctx_t *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    for (auto &k : c->contigs) if (k.alive) free_contig(k);
    for (int j = 0; j < 2; ++j) { hipFree(c->d_thr[j]); hipFree(c->d_thr32[j]); hipFree(c->d_qbase[j]); }
    hipFree(c->status_all.p);
    hipFree(c->w_ppos.p); hipFree(c->w_pcells.p); hipFree(c->flow_scratch.p); hipFree(c->w_cand.p); hipFree(c->w_ev.p); hipFree(c->w_flags.p); hipFree(c->w_lo.p); hipFree(c->w_sufmin.p); hipFree(c->w_bound.p);
    hipFree(c->d_rand_fixed); hipFree(c->meta.p); hipFree(c->fail_summ.p); hipFree(c->block_rand.p); hipFree(c->scratch_mask.p); hipFree(c->scratch_cnt.p);
    for (int s = 0; s < 2; ++s) for (int t = 0; t < 3; ++t) hipFree(c->out[s][t].p);
    hipFree(c->d_counters); hipFree(c->d_flow); hipFree(c->d_chain); hipFree(c->d_crc_table); hipFree(c->d_crc_shift);
    if (c->h_counters) hipHostFree(c->h_counters);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (Slot &sl : c->slot) {
        hipFree(sl.d_counters); hipFree(sl.gz_status.p);
        for (int t = 0; t < 3; ++t) hipFree(sl.gz_out[t].p);
        if (sl.h_counters) hipHostFree(sl.h_counters);
        for (hipEvent_t e : {sl.ev_k0, sl.ev_k1, sl.ev_done, sl.ev_fetched}) if (e) hipEventDestroy(e);
    }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *dwgsim_hip_last_error(const dwgsim_hip_ctx_t *c) { return c ? c->err.c_str() : "no context"; }

int dwgsim_hip_add_contig(dwgsim_hip_ctx_t *c, const char *name, const uint8_t *ascii, int64_t len, uint32_t contig_index)
{
    if (!c || !name || (!ascii && len > 0) || len < 0 || len > INT32_MAX) { if (c) c->err = "bad contig arguments"; return DWGSIM_HIP_ERR_ARG; }
    HIPC(c, hipSetDevice(c->device));
    int id = -1;
    for (size_t i = 0; i < c->contigs.size(); ++i) if (!c->contigs[i].alive) { id = (int)i; break; }
    if (id < 0) { c->contigs.emplace_back(); id = (int)c->contigs.size() - 1; }
    Contig &k = c->contigs[(size_t)id];
    k.name = name; k.l = len; k.contig_index = contig_index; k.ascii.assign(ascii, ascii + len); k.alive = true; k.mutated = false;
    const size_t padded = (size_t)((len + 15) & ~(int64_t)15) + CELL_PAD;
    uint8_t *d_ascii = nullptr;
    auto fill = [&]() -> int {
        HIPC(c, hipMalloc((void **)&d_ascii, padded));
        HIPC(c, hipMalloc((void **)&k.d_ref, padded));
        for (int h = 0; h < 2; ++h) { HIPC(c, hipMalloc((void **)&k.d_cells[h], padded)); HIPC(c, hipMalloc((void **)&k.d_view[h], padded / 2 + 32)); }
        HIPC(c, hipMemcpyAsync(d_ascii, ascii, (size_t)len, hipMemcpyHostToDevice, c->stream));
        HIPC(c, hipMemsetAsync(k.d_ref, 4, padded, c->stream));
        for (int h = 0; h < 2; ++h) HIPC(c, hipMemsetAsync(k.d_cells[h], 4, padded, c->stream));
        if (len > 0) launch_pack(c->stream, d_ascii, k.d_ref, k.d_cells[0], k.d_cells[1], len);
        HIPC(c, hipGetLastError());
        k.l_place = len;
        if (c->has_regions) {
            std::vector<int32_t> st, en; int64_t tot = 0;
            for (size_t q = 0; q < c->regions.contig.size(); ++q) if (c->regions.contig[q] == contig_index) { st.push_back((int32_t)c->regions.start[q]); en.push_back((int32_t)c->regions.end[q]); tot += c->regions.end[q] - c->regions.start[q]; }
            k.n_reg = (int32_t)st.size(); k.l_place = tot;
            HIPC(c, hipMalloc((void **)&k.d_reg, sizeof(int32_t) * (2 * st.size() + 2)));
            if (!st.empty()) {
                HIPC(c, hipMemcpy(k.d_reg, st.data(), sizeof(int32_t) * st.size(), hipMemcpyHostToDevice));
                HIPC(c, hipMemcpy(k.d_reg + st.size(), en.data(), sizeof(int32_t) * en.size(), hipMemcpyHostToDevice));
            }
        }
        std::string nf = c->read_prefix.empty() ? k.name : c->read_prefix + "_" + k.name;
        k.name_fixed_len = (int32_t)nf.size();
        std::string nbuf = "@" + nf; nbuf.resize(nbuf.size() < 256 ? 272 : nbuf.size() + 16, '\0');
        HIPC(c, hipMalloc((void **)&k.d_name_fixed, nbuf.size()));
        HIPC(c, hipMemcpyAsync(k.d_name_fixed, nbuf.data(), nbuf.size(), hipMemcpyHostToDevice, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        return 0;
    };
    const int rc = fill();
    (void)hipFree(d_ascii);
    if (rc != 0) { (void)hipStreamSynchronize(c->stream); free_contig(k); return rc; }      // no half-built contig stays behind a failed call
    if (c->chain_contig == id) c->chain_contig = -1;      // a recycled handle does not continue its predecessor's failure counter
    return id;
}

int dwgsim_hip_drop_contig(dwgsim_hip_ctx_t *c, int contig)
{
    if (!c || contig < 0 || (size_t)contig >= c->contigs.size() || !c->contigs[(size_t)contig].alive) return DWGSIM_HIP_ERR_ARG;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->copy_stream);
    free_contig(c->contigs[(size_t)contig]);
    if (c->chain_contig == contig) c->chain_contig = -1;
    return DWGSIM_HIP_OK;
}

static Contig *get_contig(dwgsim_hip_ctx_t *c, int contig)
{
    if (!c || contig < 0 || (size_t)contig >= c->contigs.size() || !c->contigs[(size_t)contig].alive) { if (c) c->err = "unknown contig handle"; return nullptr; }
    return &c->contigs[(size_t)contig];
}

int dwgsim_hip_set_regions(dwgsim_hip_ctx_t *c, const char *path, const char *const *names, const int64_t *lens, int n_contigs, uint64_t *total_len)
{
    if (!c || !path || n_contigs < 0 || (n_contigs && (!names || !lens))) { if (c) c->err = "bad regions arguments"; return DWGSIM_HIP_ERR_ARG; }
    if (c->prm.amplicons) { c->err = "Error: cannot use a regions BED file (-x) when simulating amplicons (-a)\n"; return DWGSIM_HIP_ERR_ARG; }
    std::vector<ContigName> tab;
    for (int i = 0; i < n_contigs; ++i) tab.push_back(ContigName{names[i], lens[i]});
    std::string err;
    if (!parse_regions(path, tab, c->regions, err)) { c->err = err; c->has_regions = false; return DWGSIM_HIP_ERR_ARG; }
    c->has_regions = true;
    uint64_t tot = 0;
    for (size_t q = 0; q < c->regions.start.size(); ++q) tot += c->regions.end[q] - c->regions.start[q];
    if (total_len) *total_len = tot;
    return DWGSIM_HIP_OK;
}

int64_t dwgsim_hip_contig_region_length(dwgsim_hip_ctx_t *c, uint32_t contig_index, const uint8_t *ascii, int64_t len)
{
    if (!c || !c->has_regions) return len;
    int64_t m = 0, num_n = 0;
    for (size_t q = 0; q < c->regions.contig.size(); ++q) if (c->regions.contig[q] == contig_index) {
        m += c->regions.end[q] - c->regions.start[q];
        // the reference walks start..end INCLUSIVE with 1-based indexing (dwgsim.c:558-559); its read of
        // s[-1] for start == 0 is out of bounds there and is counted as non-ACGT here
        for (int64_t p = c->regions.start[q]; p <= (int64_t)c->regions.end[q]; ++p) { const int ch = (p >= 1 && p - 1 < len) ? ascii[p - 1] : 'N'; if (nt4(ch) >= 4) ++num_n; }
    }
    if (m == 0) return -10;
    if (0.95 < num_n / (double)m) return -11;
    return m;
}

int dwgsim_hip_contig_set_placement_length(dwgsim_hip_ctx_t *c, int contig, int64_t l)
{
    Contig *kp = (c && contig >= 0 && (size_t)contig < c->contigs.size() && c->contigs[(size_t)contig].alive) ? &c->contigs[(size_t)contig] : nullptr;
    if (!kp || l < 0) { if (c) c->err = "bad placement-length arguments"; return DWGSIM_HIP_ERR_ARG; }
    kp->l_place = l;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_set_mutation_input(dwgsim_hip_ctx_t *c, int type, const char *path, const char *const *names, const int64_t *lens, int n_contigs)
{
    if (!c || !path || type < 0 || type > 2 || n_contigs < 0 || (n_contigs && (!names || !lens))) { if (c) c->err = "bad mutation-input arguments"; return DWGSIM_HIP_ERR_ARG; }
    std::vector<ContigName> tab;
    for (int i = 0; i < n_contigs; ++i) tab.push_back(ContigName{names[i], lens[i]});
    std::string err;
    if (!parse_mutation_input(type, path, tab, c->mutin, err)) { c->err = err; c->has_mutin = false; return DWGSIM_HIP_ERR_ARG; }
    c->has_mutin = true;
    return DWGSIM_HIP_OK;
}

// mut_debug (mut.c:379-425): where the reference's asserts end the run (SIGABRT) the call returns DWGSIM_HIP_ERR_FAILED with the assert's text
static int mut_debug_verdict(dwgsim_hip_ctx_t *c, const Contig &k, uint64_t v)
{
    if (v == ~0ull) return DWGSIM_HIP_OK;
    static const char *what[4] = {"", "(c[0]&0x3) != (c[1]&0x3)", "(c[1]&0x3) != (c[2]&0x3)", "(c[0]&0x3) == (c[1]&0x3) || (c[0]&0x3) == (c[2]&0x3)"};
    char b[256]; snprintf(b, sizeof b, "dwgsim: src/mut.c: mut_debug: Assertion `%s' failed. [%s:%lld]\n", what[v & 3], k.name.c_str(), (long long)(v >> 8) + 1);
    c->err = b;
    return DWGSIM_HIP_ERR_FAILED;
}

int dwgsim_hip_mutate_contig(dwgsim_hip_ctx_t *c, int contig)
{
    Contig *kp = get_contig(c, contig);
    if (!kp) return DWGSIM_HIP_ERR_ARG;
    Contig &k = *kp;
    HIPC(c, hipSetDevice(c->device));
    const WalkParams wp = walk_params(c);
    const int64_t l = k.l;
    const bool again = k.mutated;      // walked before: the cells start again from the resident packed reference
    if (again) for (int h = 0; h < 2; ++h) k.n_ins[h] = k.n_ins_bases[h] = 0;
    if (again && c->has_mutin) {
        const size_t padded = (size_t)((l + 15) & ~(int64_t)15) + CELL_PAD;
        for (int h = 0; h < 2; ++h) HIPC(c, hipMemcpyAsync(k.d_cells[h], k.d_ref, padded, hipMemcpyDeviceToDevice, c->stream));
    }
    k.mutated = true; k.n_cand = 0; k.summ_valid = false;
    const size_t padded_cells = (size_t)((l + 15) & ~(int64_t)15) + CELL_PAD;
    if (l == 0) return DWGSIM_HIP_OK;
    if (c->has_mutin) {      // file-driven mutations (mut.c:644-745): host resolves the entries, the GPU scatters and left-justifies
        ResolvedContig rc;
        resolve_mutation_input(c->mutin, k.contig_index, k.ascii.data(), l, (uint32_t)c->prm.seed, c->prm.is_hap != 0, rc);
        const uint32_t np = (uint32_t)rc.pos.size();
        std::vector<Event> evs;
        for (uint32_t q = 0; q < np; ++q) if (rc.cells[q] & 0x3030) { Event e; e.pos = rc.pos[q]; e.type = 4; e.hap = 3; e.base = 0; e.live = 1; e.len = 1; evs.push_back(e); }
        const uint32_t nev = (uint32_t)evs.size();
        k.n_cand = nev;
        if (np == 0) {      // nothing listed for this contig: both haplotypes are the reference
            launch_make_view(c->stream, k.d_cells[0], k.d_cells[1], (int64_t)padded_cells, k.d_view[0], k.d_view[1]);
            HIPC(c, hipGetLastError()); HIPC(c, hipStreamSynchronize(c->stream));
            return DWGSIM_HIP_OK;
        }
        if (ensure(c, c->w_ppos, sizeof(int32_t) * np) || ensure(c, c->w_pcells, sizeof(uint16_t) * np) || ensure(c, c->w_ev, sizeof(Event) * (nev ? nev : 1)) ||
            ensure(c, c->w_lo, sizeof(int32_t) * (nev ? nev : 1)) || ensure(c, c->w_sufmin, sizeof(int32_t) * ((nev ? nev : 1) + 64)) || ensure(c, c->w_bound, nev ? nev : 1)) return DWGSIM_HIP_ERR_DEVICE;      // (+ 64: segment minima of k_sufmin)
        HIPC(c, hipMemcpyAsync(c->w_ppos.p, rc.pos.data(), sizeof(int32_t) * np, hipMemcpyHostToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->w_pcells.p, rc.cells.data(), sizeof(uint16_t) * np, hipMemcpyHostToDevice, c->stream));
        if (nev) HIPC(c, hipMemcpyAsync(c->w_ev.p, evs.data(), sizeof(Event) * nev, hipMemcpyHostToDevice, c->stream));
        for (int h = 0; h < 2; ++h) {
            const size_t n = rc.ins[h].size();
            std::vector<int32_t> ip(n); std::vector<uint32_t> il(n), io(n); std::vector<uint8_t> ib;
            for (size_t q = 0; q < n; ++q) { ip[q] = rc.ins[h][q].pos; il[q] = (uint32_t)rc.ins[h][q].bases.size(); io[q] = (uint32_t)ib.size(); ib.insert(ib.end(), rc.ins[h][q].bases.begin(), rc.ins[h][q].bases.end()); }
            k.n_ins[h] = (uint32_t)n; k.n_ins_bases[h] = (uint32_t)ib.size();
            const size_t nn = n ? n : 1, nb = ib.size() ? ib.size() : 1;
            if (nn > k.cap_ins[h]) {
                hipFree(k.d_ins_pos[h]); hipFree(k.d_ins_len[h]); hipFree(k.d_ins_off[h]);
                k.cap_ins[h] = nn + nn / 4 + 64;
                HIPC(c, hipMalloc((void **)&k.d_ins_pos[h], sizeof(int32_t) * k.cap_ins[h]));
                HIPC(c, hipMalloc((void **)&k.d_ins_len[h], sizeof(uint32_t) * k.cap_ins[h]));
                HIPC(c, hipMalloc((void **)&k.d_ins_off[h], sizeof(uint32_t) * k.cap_ins[h]));
            }
            if (nb > k.cap_bases[h]) { hipFree(k.d_ins_bases[h]); k.cap_bases[h] = nb + nb / 4 + 256; HIPC(c, hipMalloc((void **)&k.d_ins_bases[h], k.cap_bases[h] + 16)); }
            if (n) {
                HIPC(c, hipMemcpy(k.d_ins_pos[h], ip.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
                HIPC(c, hipMemcpy(k.d_ins_len[h], il.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
                HIPC(c, hipMemcpy(k.d_ins_off[h], io.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
                HIPC(c, hipMemcpy(k.d_ins_bases[h], ib.data(), ib.size(), hipMemcpyHostToDevice));
            }
        }
        launch_apply_patches(c->stream, (const int32_t *)c->w_ppos.p, (const uint16_t *)c->w_pcells.p, np, k.d_cells[0], k.d_cells[1]);
        const ContigDev cd = contig_dev(k);
        HIPC(c, hipMemsetAsync(&c->d_counters[12], 0xff, 2 * sizeof(uint64_t), c->stream));
        launch_mut_debug(c->stream, k.d_ref, k.d_cells[0], k.d_cells[1], l, &c->d_counters[12]);      // mut.c:753
        if (nev) {
            if (c->seq_justify) launch_justify_seq(c->stream, (const Event *)c->w_ev.p, Count{nullptr, nev}, cd);
            else launch_justify(c->stream, (const Event *)c->w_ev.p, Count{nullptr, nev}, cd, (int32_t *)c->w_lo.p, (int32_t *)c->w_sufmin.p, (uint8_t *)c->w_bound.p);
        }
        launch_mut_debug(c->stream, k.d_ref, k.d_cells[0], k.d_cells[1], l, &c->d_counters[13]);      // mut.c:757
        launch_make_view(c->stream, k.d_cells[0], k.d_cells[1], (int64_t)padded_cells, k.d_view[0], k.d_view[1]);
        HIPC(c, hipGetLastError());
        HIPC(c, hipMemcpyAsync(&c->h_counters[12], &c->d_counters[12], 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        if (const int rc = mut_debug_verdict(c, k, c->h_counters[12])) return rc;
        return mut_debug_verdict(c, k, c->h_counters[13]);
    }
    const uint32_t nblk = (uint32_t)((l + SCAN_POS_PER_BLOCK - 1) / SCAN_POS_PER_BLOCK);
    if (ensure(c, c->scratch_mask, (size_t)nblk * SCAN_THREADS * sizeof(uint16_t))) return DWGSIM_HIP_ERR_DEVICE;
    if (ensure(c, c->scratch_cnt, (size_t)nblk * sizeof(uint32_t))) return DWGSIM_HIP_ERR_DEVICE;
    uint16_t *d_mask = (uint16_t *)c->scratch_mask.p; uint32_t *d_cnt = (uint32_t *)c->scratch_cnt.p;
    // The whole walk is enqueued without an intermediate host read-back: buffers are sized for a capacity (candidate sites are a
    // Binomial(l, mut_rate) draw: mean + 8 sigma), the kernels take their element counts from device memory, and the one
    // synchronisation at the end also tells whether a capacity was exceeded -- then the walk is simply run again with exact sizes.
    const double mean = (double)l * c->prm.mut_rate;
    uint32_t cap = (uint32_t)std::min<double>((double)l, mean + 8.0 * sqrt(mean + 1.0) + 256.0);
    size_t cap_bases = (size_t)cap * 8 + 4096;
    if (c->walk_cap >= 0) { cap = (uint32_t)c->walk_cap; cap_bases = 1; }      // dwgsim_hip_debug_option("walk_cap"): start too small, exercise the re-run
    for (int attempt = 0; ; ++attempt) {
        if (attempt > 0) {      // start again from the resident packed reference
            const size_t padded = (size_t)((l + 15) & ~(int64_t)15) + CELL_PAD;
            for (int h = 0; h < 2; ++h) HIPC(c, hipMemcpyAsync(k.d_cells[h], k.d_ref, padded, hipMemcpyDeviceToDevice, c->stream));
        }
        const size_t ncap = cap ? cap : 1;
        if (ensure(c, c->w_cand, sizeof(int32_t) * ncap) || ensure(c, c->w_ev, sizeof(Event) * ncap) ||
            ensure(c, c->w_flags, sizeof(uint4) * (ncap + 64)) ||      // (+ 64 rows / entries: the segment totals of k_scan4, the segment minima of k_sufmin)
            ensure(c, c->w_lo, sizeof(int32_t) * ncap) || ensure(c, c->w_sufmin, sizeof(int32_t) * (ncap + 64)) ||
            ensure(c, c->w_bound, ncap)) return DWGSIM_HIP_ERR_DEVICE;
        for (int h = 0; h < 2; ++h) {      // insertion tables: at most one entry per candidate; the base pools are checked on the device
            if (ncap > k.cap_ins[h]) {
                hipFree(k.d_ins_pos[h]); hipFree(k.d_ins_len[h]); hipFree(k.d_ins_off[h]);
                k.cap_ins[h] = ncap + ncap / 4 + 64;
                HIPC(c, hipMalloc((void **)&k.d_ins_pos[h], sizeof(int32_t) * k.cap_ins[h]));
                HIPC(c, hipMalloc((void **)&k.d_ins_len[h], sizeof(uint32_t) * k.cap_ins[h]));
                HIPC(c, hipMalloc((void **)&k.d_ins_off[h], sizeof(uint32_t) * k.cap_ins[h]));
            }
            if (cap_bases > k.cap_bases[h]) {
                hipFree(k.d_ins_bases[h]);
                k.cap_bases[h] = cap_bases + cap_bases / 4 + 256;
                HIPC(c, hipMalloc((void **)&k.d_ins_bases[h], k.cap_bases[h] + 16));
            }
        }
        int32_t *d_cand = (int32_t *)c->w_cand.p; Event *d_ev = (Event *)c->w_ev.p; uint4 *d_flags = (uint4 *)c->w_flags.p;
        uint32_t *d_small = reinterpret_cast<uint32_t *>(&c->d_counters[8]);   // [0] max_del, [1..4] tot4: eight words in counters[8..11], so that one copy brings counters[7..11] back
        const Count nc{&c->d_counters[7], cap};
        HIPC(c, hipMemsetAsync(&c->d_counters[7], 0, 5 * sizeof(uint64_t), c->stream));
        // K1: candidate sites -> ordered list
        const bool reset = again && attempt == 0;      // (a capacity re-run has just copied the cells back; a first walk finds them fresh from k_pack)
        launch_site_scan(c->stream, k.d_ref, l, wp, k.contig_index, d_mask, d_cnt, reset ? k.d_cells[0] : nullptr, reset ? k.d_cells[1] : nullptr);
        launch_scan_excl(c->stream, d_cnt, nblk, &c->d_counters[7]);
        launch_compact(c->stream, d_mask, d_cnt, d_cand, l, cap);
        // K2: events, liveness, insertion-table allocation
        launch_events(c->stream, d_cand, nc, k.d_ref, l, wp, k.contig_index, d_ev, &d_small[0]);
        launch_resolve(c->stream, d_ev, nc, &d_small[0], d_flags, &d_small[1]);
        // K3 + K4
        ContigDev cd = contig_dev(k);
        cd.tot4 = &d_small[1]; cd.cap_bases[0] = (uint32_t)std::min<size_t>(k.cap_bases[0], 0xFFFFFFFFu); cd.cap_bases[1] = (uint32_t)std::min<size_t>(k.cap_bases[1], 0xFFFFFFFFu);
        launch_apply(c->stream, d_ev, nc, d_flags, cd, wp);
        // mut_debug (mut.c:753, :757) cannot fire on randomly drawn mutations and is not run here: a substitution always changes the base
        // ((c + 1..3) & 3, mut.c:621), a homozygous one writes the same cell to both haplotypes and a heterozygous one leaves the other
        // haplotype's cell as it was -- the reference base, also under a deletion or an insertion, before and after left-justification
        // (which only moves an indel over bases equal to its own).  File-driven mutations (-m / -b / -v, above) can violate all three.
        if (c->seq_justify) launch_justify_seq(c->stream, d_ev, nc, cd);
        else launch_justify(c->stream, d_ev, nc, cd, (int32_t *)c->w_lo.p, (int32_t *)c->w_sufmin.p, (uint8_t *)c->w_bound.p);
        launch_make_view(c->stream, k.d_cells[0], k.d_cells[1], (int64_t)padded_cells, k.d_view[0], k.d_view[1]);
        HIPC(c, hipGetLastError());
        HIPC(c, hipMemcpyAsync(&c->h_counters[7], &c->d_counters[7], 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));      // [7] candidates, [8..11] the eight words
        HIPC(c, hipStreamSynchronize(c->stream));
        const uint64_t n_cand = c->h_counters[7];
        const uint32_t *h_small = reinterpret_cast<const uint32_t *>(&c->h_counters[8]);
        const bool fits = n_cand <= cap && h_small[2] <= k.cap_bases[0] && h_small[4] <= k.cap_bases[1];
        if (fits || attempt >= 2) {
            if (!fits) { c->err = "mutation walk: capacities still exceeded after an exact re-run"; return DWGSIM_HIP_ERR_FAILED; }
            k.n_cand = (uint32_t)n_cand;
            for (int h = 0; h < 2; ++h) { k.n_ins[h] = h_small[1 + 2 * h]; k.n_ins_bases[h] = h_small[2 + 2 * h]; }
            return DWGSIM_HIP_OK;
        }
        // exact sizes (the counts read back are those of the complete candidate list unless it was truncated: take generous ones then)
        cap = (uint32_t)std::min<uint64_t>((uint64_t)l, n_cand + 16);
        cap_bases = std::max<size_t>(cap_bases, (size_t)std::max(h_small[2], h_small[4]) * 2 + (size_t)cap * 8 + 4096);
    }
}

// ---- mutations.txt / mutations.vcf from the sparse list of mutated cells (mut.c:781-893) ----
namespace {
struct HostIns { std::vector<int32_t> pos; std::vector<uint32_t> len, off; std::vector<uint8_t> bases; };
const char *ins_text(const HostIns &t, int32_t pos, std::string &tmp)
{
    tmp.clear();
    auto it = std::lower_bound(t.pos.begin(), t.pos.end(), pos);
    if (it == t.pos.end() || *it != pos) return tmp.c_str();
    const size_t k = (size_t)(it - t.pos.begin());
    for (uint32_t q = 0; q < t.len[k]; ++q) tmp.push_back("ACGTN"[t.bases[t.off[k] + q] & 3]);
    return tmp.c_str();
}
void appendf(std::string &s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void appendf(std::string &s, const char *fmt, ...)
{
    char buf[1024]; va_list ap; va_start(ap, fmt); int n = vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (n < (int)sizeof buf) { s.append(buf, (size_t)n); return; }
    std::vector<char> big((size_t)n + 1); va_start(ap, fmt); vsnprintf(big.data(), big.size(), fmt, ap); va_end(ap); s.append(big.data(), (size_t)n);
}
} // namespace

int dwgsim_hip_mutations_text(dwgsim_hip_ctx_t *c, int contig, const char **txt, size_t *txt_len, const char **vcf, size_t *vcf_len)
{
    Contig *kp = get_contig(c, contig);
    if (!kp) return DWGSIM_HIP_ERR_ARG;
    Contig &k = *kp;
    if (!k.mutated) { c->err = "mutate_contig must run first"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    c->txt.clear(); c->vcf.clear();
    const int64_t l = k.l;
    std::vector<int32_t> pos; std::vector<uint16_t> cells;
    HostIns ins[2];
    if (l > 0 && k.n_cand > 0) {
        const uint32_t nblk = (uint32_t)((l + SCAN_POS_PER_BLOCK - 1) / SCAN_POS_PER_BLOCK);
        if (ensure(c, c->scratch_mask, (size_t)nblk * SCAN_THREADS * sizeof(uint16_t))) return DWGSIM_HIP_ERR_DEVICE;
        if (ensure(c, c->scratch_cnt, (size_t)nblk * sizeof(uint32_t))) return DWGSIM_HIP_ERR_DEVICE;
        uint16_t *d_mask = (uint16_t *)c->scratch_mask.p; uint32_t *d_cnt = (uint32_t *)c->scratch_cnt.p;
        launch_collect_mask(c->stream, k.d_cells[0], k.d_cells[1], l, d_mask, d_cnt);
        launch_scan_excl(c->stream, d_cnt, nblk, &c->d_counters[7]);
        HIPC(c, hipMemcpyAsync(&c->h_counters[7], &c->d_counters[7], sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        const uint32_t n = (uint32_t)c->h_counters[7];
        if (n) {
            int32_t *d_pos = nullptr; uint16_t *d_cells = nullptr;
            HIPC(c, hipMalloc((void **)&d_pos, sizeof(int32_t) * (size_t)n));
            HIPC(c, hipMalloc((void **)&d_cells, sizeof(uint16_t) * (size_t)n));
            launch_compact(c->stream, d_mask, d_cnt, d_pos, l, n);
            launch_gather(c->stream, d_pos, n, k.d_cells[0], k.d_cells[1], d_cells);
            pos.resize(n); cells.resize(n);
            HIPC(c, hipMemcpyAsync(pos.data(), d_pos, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            HIPC(c, hipMemcpyAsync(cells.data(), d_cells, sizeof(uint16_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            HIPC(c, hipStreamSynchronize(c->stream));
            HIPC(c, hipFree(d_pos)); HIPC(c, hipFree(d_cells));
        }
        for (int h = 0; h < 2; ++h) if (k.n_ins[h]) {
            ins[h].pos.resize(k.n_ins[h]); ins[h].len.resize(k.n_ins[h]); ins[h].off.resize(k.n_ins[h]); ins[h].bases.resize(k.n_ins_bases[h]);
            HIPC(c, hipMemcpy(ins[h].pos.data(), k.d_ins_pos[h], sizeof(int32_t) * k.n_ins[h], hipMemcpyDeviceToHost));
            HIPC(c, hipMemcpy(ins[h].len.data(), k.d_ins_len[h], sizeof(uint32_t) * k.n_ins[h], hipMemcpyDeviceToHost));
            HIPC(c, hipMemcpy(ins[h].off.data(), k.d_ins_off[h], sizeof(uint32_t) * k.n_ins[h], hipMemcpyDeviceToHost));
            HIPC(c, hipMemcpy(ins[h].bases.data(), k.d_ins_bases[h], k.n_ins_bases[h], hipMemcpyDeviceToHost));
        }
    }
    // sparse restatement of the per-position loop: only listed positions can print; "previous position
    // mutated" (mut_prev, mut.c:890-891) is "position i-1 is listed with a mutated cell on that haplotype"
    static const char B5[] = "ACGTN";       // B5[5] is the terminating NUL, as in the reference for code 5 ('-')
    const char *nm = k.name.c_str();
    std::string tmp;
    auto cell = [&](size_t e, int h) -> uint8_t { return (uint8_t)(h ? cells[e] >> 8 : cells[e] & 0xff); };
    for (size_t e = 0; e < pos.size(); ++e) {
        const int64_t i = pos[e];
        const uint8_t r0 = nt4(k.ascii[(size_t)i]), c1 = cell(e, 0), c2 = cell(e, 1);
        if (r0 >= 4) continue;
        const bool adj = e > 0 && pos[e - 1] == i - 1;
        const bool prev0 = adj && (cell(e - 1, 0) & TMASK) != T_NONE, prev1 = adj && (cell(e - 1, 1) & TMASK) != T_NONE;
        appendf(c->txt, "%s\t%lld\t", nm, (long long)i + 1);
        const bool hom = (c1 & BTMASK) == (c2 & BTMASK);
        const uint8_t t1 = c1 & TMASK, t2 = c2 & TMASK;
        if (hom ? t1 == T_SUB : (t1 == T_SUB || t2 == T_SUB)) {
            if (hom) {
                appendf(c->txt, "%c\t%c\t3\n", B5[r0], B5[c1 & 0xf]);
                appendf(c->vcf, "%s\t%lld\t.\t%c\t%c\t100\tPASS\tAF=1.0;pl=3;mt=SUBSTITUTE\n", nm, (long long)i + 1, B5[r0], B5[c1 & 0xf]);
            } else {
                const int hap = t1 == T_SUB ? 1 : 2;
                appendf(c->txt, "%c\t%c\t%d\n", B5[r0], "XACMGRSVTWYHKDBN"[1 << (c1 & 3) | 1 << (c2 & 3)], hap);
                appendf(c->vcf, "%s\t%lld\t.\t%c\t%c\t100\tPASS\tAF=0.5;pl=%d;mt=SUBSTITUTE\n", nm, (long long)i + 1, B5[r0], B5[(hap == 1 ? c1 : c2) & 0xf], hap);
            }
        } else if (hom ? t1 == T_DEL : (t1 == T_DEL || t2 == T_DEL)) {
            const int pl = hom ? 3 : (t1 == T_DEL ? 1 : 2);
            appendf(c->txt, "%c\t-\t%d\n", B5[r0], pl);
            const bool open = hom ? (!prev0 || !prev1) : !(pl == 1 ? prev0 : prev1);
            if (open) {      // one VCF record for the run, anchored at the previous reference base (mut.c:801-815)
                appendf(c->vcf, "%s\t%lld\t.\t", nm, (long long)i);
                if (i > 0) c->vcf.push_back(B5[nt4(k.ascii[(size_t)i - 1])]);
                size_t ee = e; int64_t j = i;
                for (;;) {
                    c->vcf.push_back(B5[nt4(k.ascii[(size_t)j])]);
                    if (j + 1 >= l) break;
                    // cell at j+1: listed -> its cells, else unmutated
                    if (ee + 1 < pos.size() && pos[ee + 1] == j + 1) {
                        ++ee; ++j;
                        const uint8_t a1 = cell(ee, 0), a2 = cell(ee, 1);
                        const bool h2 = (a1 & BTMASK) == (a2 & BTMASK);
                        if (!(h2 == hom && ((pl == 2 ? a2 : a1) & TMASK) == T_DEL)) break;
                    } else break;
                }
                if (i > 0) appendf(c->vcf, "\t%c", B5[nt4(k.ascii[(size_t)i - 1])]); else c->vcf.append("\t.");
                appendf(c->vcf, "\t100\tPASS\tAF=%s;pl=%d;mt=DELETE\n", hom ? "1.0" : "0.5", pl);
            }
        } else {
            const int pl = hom ? 3 : (t1 == T_INS ? 1 : 2);
            const char *seq = ins_text(ins[pl == 2 ? 1 : 0], (int32_t)i, tmp);
            appendf(c->txt, "-\t%s\t%d\n", seq, pl);
            appendf(c->vcf, "%s\t%lld\t.\t%c\t%c%s\t100\tPASS\tAF=%s;pl=%d;mt=INSERT\n", nm, (long long)i + 1, B5[r0], B5[r0], seq, hom ? "1.0" : "0.5", pl);
        }
    }
    if (txt) *txt = c->txt.data();
    if (txt_len) *txt_len = c->txt.size();
    if (vcf) *vcf = c->vcf.data();
    if (vcf_len) *vcf_len = c->vcf.size();
    return DWGSIM_HIP_OK;
}

// ---- read simulation ----
static int build_sim_args(dwgsim_hip_ctx_t *c, Contig &k, uint64_t first_ii, uint64_t n_pairs, SimArgs &a)
{
    const dwgsim_hip_params_t &p = c->prm;
    memset(&a, 0, sizeof a);
    a.p.std_dev = p.std_dev; a.p.mut_freq = p.mut_freq; a.p.rand_read = p.rand_read; a.p.quality_std = p.quality_std;
    a.p.dist = p.dist; a.p.is_inner = p.is_inner; a.p.len[0] = p.length[0]; a.p.len[1] = p.length[1]; a.p.max_n = p.max_n;
    a.p.strandedness = p.strandedness; a.p.read_one_strand = p.read_one_strand; a.p.amplicons = p.amplicons;
    a.p.fixed_quality = p.fixed_quality; a.p.data_type = p.data_type;
    a.p.has_bfast = p.reads_output_type != 1; a.p.has_bwa = p.reads_output_type != 2;
    a.p.seed = (uint32_t)p.seed;
    lazy_quality_params(p.quality_std, &a.p.q_k, &a.p.q_eps, &a.p.q_lmin, &a.p.q_near1);
    a.c = contig_dev(k);
    a.first_ii = first_ii; a.n_pairs = n_pairs; a.chain = c->d_chain;
    a.l_place = k.l_place; a.have_regions = c->has_regions ? 1 : 0; a.n_reg = k.n_reg; a.reg_start = k.d_reg; a.reg_end = k.d_reg ? k.d_reg + k.n_reg : nullptr;
    for (int j = 0; j < 2; ++j) { a.e_thr[j] = c->d_thr[j]; a.e_thr32[j] = c->d_thr32[j]; a.qbase[j] = c->d_qbase[j] ? c->d_qbase[j] : c->d_qbase[0]; }
    a.qb_words = c->qb_words;
    a.e_full = c->e_full;
    a.name_fixed = k.d_name_fixed; a.name_fixed_len = k.name_fixed_len;
    a.summ[0] = k.d_summ[0]; a.summ[1] = k.d_summ[1];      // null unless count_random built them
    a.rand_fixed = c->d_rand_fixed; a.rand_fixed_len = c->rand_fixed_len;
    // lanes per k_simulate block: the staged read (lds_words per lane) must fit LDS; long Illumina / SOLiD reads get one-wave blocks
    const int lmax0 = p.length[0] > p.length[1] ? p.length[0] : p.length[1];
    a.sim_threads = SIM_THREADS;
    // the FIFO writer unless its LDS costs a resident block per CU (Illumina reads between ~150 and ~240 bases): then 16-byte pieces from registers
    a.fifo = 1;
    if (p.data_type == 0) {
        const int cap = (p.reads_output_type == 0) ? 4 : 5;      // waves per SIMD the registers allow (k_simulate launch bounds)
        const size_t w0 = (size_t)((lmax0 + 7) / 8);
        if (sim_blocks_per_cu(sim_lds_bytes(w0, SIM_THREADS, (size_t)c->qb_words, true), cap) < sim_blocks_per_cu(sim_lds_bytes(w0, SIM_THREADS, (size_t)c->qb_words, false), cap)) a.fifo = 0;
        if (c->writer >= 0) a.fifo = c->writer ? 1 : 0;
    }
    auto lds_need = [&](int lanes) { return sim_lds_bytes((size_t)((lmax0 + 7) / 8), (size_t)lanes, (size_t)c->qb_words, a.fifo != 0); };     // staged bases + the two base-quality tables + the text FIFOs
    if (p.data_type != 2 && (lds_need(SIM_THREADS) > SIM_LDS_BUDGET || c->force_threads == SIM_THREADS_LONG)) {
        a.sim_threads = SIM_THREADS_LONG; a.fifo = 1;
        if (lds_need(SIM_THREADS_LONG) > SIM_LDS_BUDGET) {
            char b[160]; snprintf(b, sizeof b, "dwgsim-hip: reads longer than %d bases are not supported for -c 0 / -c 1\n", (int)((SIM_LDS_BUDGET - SIM_THREADS_LONG * SIM_FIFO_BYTES) / (SIM_THREADS_LONG * 4 + 2) * 8));
            c->err = b; return DWGSIM_HIP_ERR_UNSUP;
        }
    }
    const uint64_t sim_ppb = (uint64_t)(a.sim_threads / (p.length[1] > 0 ? 2 : 1));      // pairs per k_simulate block
    const uint64_t nblk = (n_pairs + sim_ppb - 1) / sim_ppb;
    const uint64_t nblk_place = (n_pairs + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK;       // k_place (count_random) writes one entry per ITS block
    const uint64_t nblk_rand = nblk > nblk_place ? nblk : nblk_place;
    if (ensure(c, c->block_rand, sizeof(uint32_t) * (size_t)(nblk_rand ? nblk_rand : 1))) return DWGSIM_HIP_ERR_DEVICE;
    if (ensure(c, c->status_all, 4 * sizeof(uint64_t) * (size_t)(nblk ? nblk : 1))) return DWGSIM_HIP_ERR_DEVICE;      // the four look-back arrays, contiguous: one memset per batch
    if (ensure(c, c->meta, sizeof(uint32_t) * ((size_t)n_pairs + 8))) return DWGSIM_HIP_ERR_DEVICE;      // (+ padding for 16-byte reads)
    a.meta = (uint32_t *)c->meta.p;
    a.block_rand = (uint32_t *)c->block_rand.p; a.counters = c->d_counters;
    for (int j = 0; j < 4; ++j) a.status[j] = (uint64_t *)c->status_all.p + (size_t)j * (size_t)(nblk ? nblk : 1);
    const int lmax = p.length[0] > p.length[1] ? p.length[0] : p.length[1];
    a.cap = lmax;
    if (p.data_type == 2) {        // room for flow-space insertions: ~2.4 empty flows per base, each inserting with probability e, plus cascades
        const double emax = p.e_start[0] > p.e_start[1] ? p.e_start[0] : p.e_start[1];
        a.cap = flow_read_capacity(lmax, emax);
    }
    a.lds_words = (a.cap + 7) / 8;
    a.flow = c->d_flow; a.flow_len = (int32_t)c->flow.size();
    a.flow_scratch = nullptr;
    if (p.data_type == 2) {
        const size_t nthr = (size_t)SIM_THREADS;
        const size_t words = (size_t)flow_words_per_lane(a.lds_words, a.cap) * nthr * (size_t)(nblk ? nblk : 1);
        if (ensure(c, c->flow_scratch, words * sizeof(uint32_t))) return DWGSIM_HIP_ERR_DEVICE;
        a.flow_scratch = (uint32_t *)c->flow_scratch.p;
    }
    return 0;
}

int dwgsim_hip_count_random(dwgsim_hip_ctx_t *c, int contig, uint64_t first_ii, uint64_t n_pairs, uint64_t *n_random)
{
    Contig *kp = get_contig(c, contig);
    if (!kp) return DWGSIM_HIP_ERR_ARG;
    if (!kp->mutated) { c->err = "mutate_contig must run first"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    if (n_random) *n_random = 0;
    if (n_pairs == 0) return DWGSIM_HIP_OK;
    if (!kp->summ_valid) {         // per-64-cell summaries of the two haplotypes: let k_place accept clean windows without walking them
        const size_t nb = (size_t)((kp->l + SUMM_CELLS - 1) / SUMM_CELLS);
        for (int h = 0; h < 2; ++h) {
            if (!kp->d_summ[h]) HIPC(c, hipMalloc((void **)&kp->d_summ[h], sizeof(uint16_t) * (nb ? nb : 1)));
            launch_summarize(c->stream, kp->d_cells[h], kp->l, kp->d_summ[h]);
        }
        kp->summ_valid = true;
    }
    SimArgs a;
    if (const int rc = build_sim_args(c, *kp, first_ii, n_pairs, a)) return rc;
    HIPC(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(uint64_t), c->stream));
    launch_place(c->stream, a);
    const uint32_t nblk = (uint32_t)((n_pairs + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK);
    launch_scan_excl(c->stream, a.block_rand, nblk, &c->d_counters[3]);
    HIPC(c, hipMemcpyAsync(c->h_counters, c->d_counters, N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->h_counters[2]) { char b[128]; snprintf(b, sizeof b, "\r[dwgsim_core] failed to generate a read after %d trials\n", MAX_ATTEMPTS + 1); c->err = b; return DWGSIM_HIP_ERR_FAILED; }
    if (n_random) *n_random = c->h_counters[3];
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_set_fail_carry(dwgsim_hip_ctx_t *c, uint64_t carry)
{
    if (!c) return DWGSIM_HIP_ERR_ARG;
    c->carry_override = carry; c->has_carry_override = true;
    return DWGSIM_HIP_OK;
}

// Enqueue one batch on the compute stream: [chain set] -> memsets -> k_simulate -> abort-rule epilogue -> counters to the slot's pinned
// mirror -> event.  Nothing here waits for the GPU (buffers only grow between batches of different shapes, and hipFree synchronises).
int dwgsim_hip_simulate_async(dwgsim_hip_ctx_t *c, int contig, uint64_t first_ii, uint64_t n_pairs, uint64_t rand_base, int slot)
{
    Contig *kp = get_contig(c, contig);
    if (!kp || slot < 0 || slot > 1) { if (c) c->err = "bad simulate arguments"; return DWGSIM_HIP_ERR_ARG; }
    if (!kp->mutated) { c->err = "mutate_contig must run first"; return DWGSIM_HIP_ERR_STATE; }
    Slot &sl = c->slot[slot];
    if (sl.pending) { c->err = "simulate: the slot still holds a batch that was not waited for"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    for (int t = 0; t < 3; ++t) sl.out_bytes[t] = sl.gz_bytes[t] = 0;
    sl.n_pairs = n_pairs; sl.empty = n_pairs == 0;
    // the reference's failure counter (dwgsim.c:635) runs over the pairs of ONE contig in index order: it is carried from the previous
    // batch only when this one continues it; any other range starts from zero unless the caller supplied the carry (sharded jobs)
    const bool continues = c->chain_contig == contig && c->chain_next_ii == first_ii && first_ii != 0;
    const bool set_carry = c->has_carry_override || !continues;
    const uint64_t carry = c->has_carry_override ? c->carry_override : 0;
    c->has_carry_override = false;
    const bool set_rand = rand_base != DWGSIM_HIP_RAND_CHAIN;
    if (set_rand || set_carry) launch_chain_set(c->stream, c->d_chain, rand_base, set_rand ? 1 : 0, carry, set_carry ? 1 : 0);
    c->chain_contig = contig; c->chain_next_ii = first_ii + n_pairs;
    if (n_pairs == 0) return DWGSIM_HIP_OK;
    SimArgs a;
    if (const int rc = build_sim_args(c, *kp, first_ii, n_pairs, a)) return rc;
    if (c->has_regions && kp->n_reg == 0 && c->prm.rand_read < 1.0) { c->err = "dwgsim-hip: this contig has no target region (-x): the reference's placement loop would not terminate\n"; return DWGSIM_HIP_ERR_ARG; }
    // upper bound of one FASTQ record (name tail: 2 positions <= 10 digits, 6 counters, 16 hex digits)
    const dwgsim_hip_params_t &p = c->prm;
    const int fixed_max = kp->name_fixed_len > c->rand_fixed_len ? kp->name_fixed_len : c->rand_fixed_len;
    size_t cap[3] = {0, 0, 0};
    for (int j = 0; j < 2; ++j) if (p.length[j] > 0) cap[j] = (size_t)n_pairs * (size_t)(1 + fixed_max + 120 + 3 + 2 * (p.data_type == 2 ? a.cap : p.length[j]) + 4);
    cap[2] = cap[0] + cap[1];
    if (!a.p.has_bwa) cap[0] = cap[1] = 0;
    if (!a.p.has_bfast) cap[2] = 0;
    for (int t = 0; t < 3; ++t) { if (ensure(c, c->out[slot][t], cap[t] + 64)) return DWGSIM_HIP_ERR_DEVICE; a.out[t] = (uint8_t *)c->out[slot][t].p; }
    a.counters = sl.d_counters;
    const uint64_t sim_ppb = (uint64_t)(a.sim_threads / (p.length[1] > 0 ? 2 : 1));
    const uint32_t nblk = (uint32_t)((n_pairs + sim_ppb - 1) / sim_ppb);
    const size_t nfb = (size_t)((n_pairs + 256ull * 64 - 1) / (256ull * 64));
    if (ensure(c, c->fail_summ, (nfb * 4 + 2) * sizeof(uint64_t))) return DWGSIM_HIP_ERR_DEVICE;
    if (sl.fetch_in_flight) { HIPC(c, hipStreamWaitEvent(c->stream, sl.ev_fetched, 0)); sl.fetch_in_flight = false; }      // the slot's text is still being copied out
    HIPC(c, hipMemsetAsync(sl.d_counters, 0, N_COUNTERS * sizeof(uint64_t), c->stream));
    HIPC(c, hipMemsetAsync(a.status[0], 0, 4 * sizeof(uint64_t) * (size_t)nblk, c->stream));
    HIPC(c, hipEventRecord(sl.ev_k0, c->stream));
    launch_simulate(c->stream, a);
    HIPC(c, hipEventRecord(sl.ev_k1, c->stream));
    launch_failrule(c->stream, a.meta, n_pairs, (uint64_t *)c->fail_summ.p, sl.d_counters, c->d_chain);
    if (c->gzip_on) {      // the .gz form of every stream, enqueued behind the text (lengths are read on the device: counters[4 + t])
        size_t nch[3], off = 0;
        for (int t = 0; t < 3; ++t) { nch[t] = (size_t)gz_chunks(cap[t]); off += nch[t]; }
        if (ensure(c, sl.gz_status, sizeof(uint64_t) * (off ? off : 1))) return DWGSIM_HIP_ERR_DEVICE;
        HIPC(c, hipMemsetAsync(sl.gz_status.p, 0, sizeof(uint64_t) * (off ? off : 1), c->stream));
        off = 0;
        for (int t = 0; t < 3; ++t) {
            if (cap[t] == 0) continue;
            const size_t gcap = (size_t)gz_capacity(cap[t]);
            if (ensure(c, sl.gz_out[t], gcap + 64)) return DWGSIM_HIP_ERR_DEVICE;
            launch_gzip(c->stream, a.out[t], &sl.d_counters[4 + t], cap[t], (uint8_t *)sl.gz_out[t].p, gcap, (uint64_t *)sl.gz_status.p + off, &sl.d_counters[28 + t], &sl.d_counters[24 + t], &sl.d_counters[2],
                        c->d_crc_table, c->d_crc_shift);
            off += nch[t];
        }
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(sl.h_counters, sl.d_counters, N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipEventRecord(sl.ev_done, c->stream));
    sl.pending = true;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_wait(dwgsim_hip_ctx_t *c, int slot, dwgsim_hip_batch_t *out)
{
    if (!c || slot < 0 || slot > 1) { if (c) c->err = "bad slot"; return DWGSIM_HIP_ERR_ARG; }
    Slot &sl = c->slot[slot];
    if (out) memset(out, 0, sizeof *out);
    if (sl.empty) { sl.pending = false; return DWGSIM_HIP_OK; }
    if (!sl.pending) { c->err = "wait: no batch was enqueued on this slot"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipEventSynchronize(sl.ev_done));
    sl.pending = false;
    const uint64_t *h = sl.h_counters;
    if (h[2] & 4) { c->err = "dwgsim-hip: no fragment placement satisfied the target regions (-x) after 2^20 tries (the reference would not terminate)\n"; return DWGSIM_HIP_ERR_FAILED; }
    if (h[2] & 2) { c->err = "dwgsim-hip: a read outgrew its buffer (or degenerated) in the flow-error model\n"; return DWGSIM_HIP_ERR_FAILED; }
    if ((h[2] & ~8ull) || h[20]) {      // one pair used up its 10 001 attempts, or the counter of failed attempts over the pairs of the contig passed the limit (dwgsim.c:635, :833-843)
        char b[128]; snprintf(b, sizeof b, "\r[dwgsim_core] failed to generate a read after %d trials\n", MAX_ATTEMPTS + 1); c->err = b; return DWGSIM_HIP_ERR_FAILED;
    }
    if (h[2] & 8) { c->err = "dwgsim-hip: the gzip output buffer is too small for this text\n"; return DWGSIM_HIP_ERR_FAILED; }
    for (int t = 0; t < 3; ++t) { sl.out_bytes[t] = h[4 + t]; sl.gz_bytes[t] = h[24 + t]; }
    if (c->phases) {   // only meaningful with the -DDW_PHASE_TIMING build (tools/phase_profile.sh)
        uint64_t tot = 0; for (int k = 0; k < 8; ++k) tot += h[8 + k];
        fprintf(stderr, "[phases]");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " p%d=%.1f%%", k, tot ? 100.0 * h[8 + k] / tot : 0.0);
        fprintf(stderr, " (ticks %llu)\n", (unsigned long long)tot);
    }
    if (out) {
        out->n_pairs = sl.n_pairs; out->n_random = h[3]; out->n_retries = h[1];
        for (int t = 0; t < 3; ++t) { out->bytes[t] = sl.out_bytes[t]; out->dev_ptr[t] = c->out[slot][t].p; }
        for (int t = 0; t < 3; ++t) out->gz_bytes[t] = sl.gz_bytes[t];
        for (int t = 0; t < 4; ++t) out->fail_seg[t] = h[16 + t];
        out->fail_carry = h[21];
        HIPC(c, hipEventElapsedTime(&out->sim_kernel_ms, sl.ev_k0, sl.ev_k1));
        out->kernel_ms = out->sim_kernel_ms;
    }
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_simulate(dwgsim_hip_ctx_t *c, int contig, uint64_t first_ii, uint64_t n_pairs, uint64_t rand_base, int slot, dwgsim_hip_batch_t *out)
{
    if (out) memset(out, 0, sizeof *out);
    const int rc = dwgsim_hip_simulate_async(c, contig, first_ii, n_pairs, rand_base, slot);
    if (rc != DWGSIM_HIP_OK) return rc;
    return dwgsim_hip_wait(c, slot, out);
}

void *dwgsim_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void dwgsim_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

// Copies on the context's second stream: a batch that is being copied out of one slot overlaps with the kernels filling the other.
int dwgsim_hip_fetch_async(dwgsim_hip_ctx_t *c, int slot, int stream, void *host_dst, size_t cap)
{
    if (!c || slot < 0 || slot > 1 || stream < 0 || stream > 2 || (!host_dst && cap)) { if (c) c->err = "bad fetch arguments"; return DWGSIM_HIP_ERR_ARG; }
    Slot &sl = c->slot[slot];
    if (sl.pending) { c->err = "fetch: wait for the batch first (its sizes are not known yet)"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    const size_t n = (size_t)sl.out_bytes[stream];
    if (n > cap) { c->err = "fetch: destination too small"; return DWGSIM_HIP_ERR_ARG; }
    if (n == 0) return DWGSIM_HIP_OK;
    HIPC(c, hipMemcpyAsync(host_dst, c->out[slot][stream].p, n, hipMemcpyDeviceToHost, c->copy_stream));
    HIPC(c, hipEventRecord(sl.ev_fetched, c->copy_stream));
    sl.fetch_in_flight = true;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_set_gzip(dwgsim_hip_ctx_t *c, int on)
{
    if (!c) return DWGSIM_HIP_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    if (on && !c->d_crc_table) {
        std::vector<uint32_t> tab(4 * 256), sh(16 * 1024);
        gz_host_tables(tab.data(), sh.data());
        HIPC(c, hipMalloc((void **)&c->d_crc_table, tab.size() * 4)); HIPC(c, hipMalloc((void **)&c->d_crc_shift, sh.size() * 4));
        HIPC(c, hipMemcpy(c->d_crc_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        HIPC(c, hipMemcpy(c->d_crc_shift, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
    }
    c->gzip_on = on != 0;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_fetch_gz_async(dwgsim_hip_ctx_t *c, int slot, int stream, void *host_dst, size_t cap)
{
    if (!c || slot < 0 || slot > 1 || stream < 0 || stream > 2 || (!host_dst && cap)) { if (c) c->err = "bad fetch arguments"; return DWGSIM_HIP_ERR_ARG; }
    Slot &sl = c->slot[slot];
    if (sl.pending) { c->err = "fetch: wait for the batch first (its sizes are not known yet)"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    const size_t n = (size_t)sl.gz_bytes[stream];
    if (n > cap) { c->err = "fetch: destination too small"; return DWGSIM_HIP_ERR_ARG; }
    if (n == 0) return DWGSIM_HIP_OK;
    HIPC(c, hipMemcpyAsync(host_dst, sl.gz_out[stream].p, n, hipMemcpyDeviceToHost, c->copy_stream));
    HIPC(c, hipEventRecord(sl.ev_fetched, c->copy_stream));
    sl.fetch_in_flight = true;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_fetch_wait(dwgsim_hip_ctx_t *c, int slot)
{
    if (!c || slot < 0 || slot > 1) { if (c) c->err = "bad slot"; return DWGSIM_HIP_ERR_ARG; }
    Slot &sl = c->slot[slot];
    if (!sl.fetch_in_flight) return DWGSIM_HIP_OK;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipEventSynchronize(sl.ev_fetched));
    sl.fetch_in_flight = false;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_fetch(dwgsim_hip_ctx_t *c, int slot, int stream, void *host_dst, size_t cap)
{
    if (!c || slot < 0 || slot > 1 || stream < 0 || stream > 2 || (!host_dst && cap)) return DWGSIM_HIP_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    Slot &sl = c->slot[slot];
    if (sl.pending) { c->err = "fetch: wait for the batch first (its sizes are not known yet)"; return DWGSIM_HIP_ERR_STATE; }
    const size_t n = (size_t)sl.out_bytes[stream];
    if (n > cap) { c->err = "fetch: destination too small"; return DWGSIM_HIP_ERR_ARG; }
    if (n == 0) return DWGSIM_HIP_OK;
    {   // a pinned (page-locked / registered) destination takes one direct copy at link speed
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, host_dst) == hipSuccess && at.type == hipMemoryTypeHost) {
            HIPC(c, hipMemcpyAsync(host_dst, c->out[slot][stream].p, n, hipMemcpyDeviceToHost, c->copy_stream));
            HIPC(c, hipStreamSynchronize(c->copy_stream));
            return DWGSIM_HIP_OK;
        }
        (void)hipGetLastError();      // pageable memory: the query reports an error that must not stick
    }
    // double-buffered pinned staging: D2H of chunk k+1 overlaps the host copy of chunk k
    const size_t CH = (size_t)16 << 20;
    if (!c->h_stage) { HIPC(c, hipHostMalloc(&c->h_stage, 2 * CH, hipHostMallocDefault)); c->h_stage_cap = 2 * CH; }
    const uint8_t *src = (const uint8_t *)c->out[slot][stream].p;
    uint8_t *stage[2] = {(uint8_t *)c->h_stage, (uint8_t *)c->h_stage + CH};
    size_t done = 0; int b = 0;
    size_t cur = n < CH ? n : CH;
    HIPC(c, hipMemcpyAsync(stage[0], src, cur, hipMemcpyDeviceToHost, c->copy_stream));
    while (done < n) {
        HIPC(c, hipStreamSynchronize(c->copy_stream));
        const size_t next_off = done + cur, next = next_off < n ? ((n - next_off) < CH ? (n - next_off) : CH) : 0;
        if (next) HIPC(c, hipMemcpyAsync(stage[b ^ 1], src + next_off, next, hipMemcpyDeviceToHost, c->copy_stream));
        memcpy((uint8_t *)host_dst + done, stage[b], cur);
        done += cur; cur = next; b ^= 1;
    }
    return DWGSIM_HIP_OK;
}

// Test / analysis hook (not part of the drop-in surface): occurrences of `byte` in one finished stream of a waited-for slot, counted on the device.
int dwgsim_hip_debug_count_byte(dwgsim_hip_ctx_t *c, int slot, int stream, int byte, uint64_t *count)
{
    if (!c || slot < 0 || slot > 1 || stream < 0 || stream > 2 || !count) return DWGSIM_HIP_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    Slot &sl = c->slot[slot];
    if (sl.pending) { c->err = "count_byte: wait for the batch first"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipMemsetAsync(&c->d_counters[15], 0, sizeof(uint64_t), c->stream));
    if (sl.out_bytes[stream]) launch_count_byte(c->stream, (const uint8_t *)c->out[slot][stream].p, sl.out_bytes[stream], (uint32_t)(byte & 0xff), &c->d_counters[15]);
    HIPC(c, hipMemcpyAsync(&c->h_counters[15], &c->d_counters[15], sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    *count = c->h_counters[15];
    return DWGSIM_HIP_OK;
}

// Test hook (not part of the drop-in surface): the gzip kernel on arbitrary host bytes (the product only ever feeds it FASTQ text): the members
// of `n` bytes of `text` into `out` (cap bytes), *out_n = their total size.  Synchronous.
int dwgsim_hip_debug_gzip(dwgsim_hip_ctx_t *c, const void *text, size_t n, void *out, size_t cap, size_t *out_n)
{
    if (!c || (!text && n) || !out_n) return DWGSIM_HIP_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    if (const int rc = dwgsim_hip_set_gzip(c, c->gzip_on ? 1 : 0); rc < 0) return rc;
    if (!c->d_crc_table) { const bool was = c->gzip_on; if (const int rc = dwgsim_hip_set_gzip(c, 1); rc < 0) return rc; c->gzip_on = was; }
    *out_n = 0;
    if (n == 0) return DWGSIM_HIP_OK;
    const size_t gcap = (size_t)gz_capacity(n), nch = (size_t)gz_chunks(n);
    uint8_t *d_text = nullptr, *d_out = nullptr; uint64_t *d_aux = nullptr;       // aux: [0] length, [1] ticket, [2] total, [3] flags, [4..] look-back words
    auto cleanup = [&]() { hipFree(d_text); hipFree(d_out); hipFree(d_aux); };
    if (hipMalloc((void **)&d_text, n + 64) != hipSuccess || hipMalloc((void **)&d_out, gcap + 64) != hipSuccess || hipMalloc((void **)&d_aux, sizeof(uint64_t) * (4 + nch)) != hipSuccess) { cleanup(); c->err = "out of device memory"; return DWGSIM_HIP_ERR_DEVICE; }
    const uint64_t n64 = n;
    bool ok = hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c->stream) == hipSuccess && hipMemsetAsync(d_aux, 0, sizeof(uint64_t) * (4 + nch), c->stream) == hipSuccess &&
              hipMemcpyAsync(d_aux, &n64, sizeof n64, hipMemcpyHostToDevice, c->stream) == hipSuccess;
    if (ok) {
        launch_gzip(c->stream, d_text, &d_aux[0], n, d_out, gcap, &d_aux[4], &d_aux[1], &d_aux[2], &d_aux[3], c->d_crc_table, c->d_crc_shift);
        uint64_t res[4] = {0, 0, 0, 0};
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(res, d_aux, sizeof res, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        if (ok && (res[3] & 8)) { cleanup(); c->err = "dwgsim-hip: the gzip output buffer is too small for this text\n"; return DWGSIM_HIP_ERR_FAILED; }
        if (ok && res[2] > cap) { cleanup(); c->err = "debug_gzip: destination too small"; return DWGSIM_HIP_ERR_ARG; }
        if (ok) { ok = hipMemcpy(out, d_out, (size_t)res[2], hipMemcpyDeviceToHost) == hipSuccess; *out_n = (size_t)res[2]; }
    }
    cleanup();
    if (!ok) { c->err = "debug_gzip: device error"; return DWGSIM_HIP_ERR_DEVICE; }
    return DWGSIM_HIP_OK;
}

// Test / analysis hooks (not part of the drop-in surface): "justify_seq" = 1 runs the left-justification from one thread (cross-check),
// "walk_cap" = n starts the mutation walk with a capacity of n candidates and a 1-byte inserted-base pool (exercises the exact re-run),
// "phases" = 1 prints the phase split of the -DDW_PHASE_TIMING analysis build,
// "writer" = 0 / 1 forces the register / FIFO record writer of the Illumina kernels (-1: chosen by LDS occupancy),
// "sim_threads" = 64 forces the one-wave blocks of the long-read variant (measured: 25 % slower on 2 x 150 bp, small jobs included),
// "walk_seg_min" = n runs the walk's two serial scans in their segmented form from a capacity of n candidates on (default 16384; 0 restores it).
int dwgsim_hip_debug_option(dwgsim_hip_ctx_t *c, const char *key, int64_t value)
{
    if (!c || !key) return DWGSIM_HIP_ERR_ARG;
    if (!strcmp(key, "justify_seq")) c->seq_justify = value != 0;
    else if (!strcmp(key, "walk_cap")) c->walk_cap = value;
    else if (!strcmp(key, "phases")) c->phases = value != 0;
    else if (!strcmp(key, "writer")) c->writer = (int)value;
    else if (!strcmp(key, "sim_threads")) c->force_threads = (int)value;
    else if (!strcmp(key, "walk_seg_min")) walk_debug_seg_min((uint32_t)value);      // (process-wide)
    else { c->err = "unknown debug option"; return DWGSIM_HIP_ERR_ARG; }
    return DWGSIM_HIP_OK;
}

} // extern "C"
