// dw_host.cpp -- C-ABI (include/dwgsim_hip.h) over the HIP kernels of dw_walk.hip / dw_simulate.hip.
//
// Host responsibilities, mirroring what dwgsim_core() does around its two hot loops:
//   * option defaults / checks / error-ramp tables      (dwgsim_opt.c:40-80, :307-371, :459-460)
//   * contig scheduling arithmetic                       (dwgsim.c:535-537, :582-590, :595-618)
//   * device residency of contigs and mutated haplotypes (replaces seq_t / mutseq_t, mut.h:12-47)
//   * mutations.txt / .vcf text from the sparse list of mutated cells (mut.c:781-893)
// There is no CPU implementation of the hot path here: without a HIP device create() fails.
//
// Contigs are resident in GROUPS: the contigs of one dwgsim_hip_add_contigs call share one coordinate space (contig k at a multiple of
// GROUP_ALIGN cells, unmutated N between them), one chain of walk kernels mutates all of them, and one k_simulate launch can cover read-index
// ranges of several of them -- so a job of thousands of small contigs (the reference's contig loop, dwgsim.c:519-625, has no fixed cost per
// contig) pays the walk chain, the launches and the host synchronisations once per group instead of once per contig.  A group of one contig
// is the plain case.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <ctype.h>
#include <stdarg.h>
#include <stdint.h>
#include <sys/mman.h>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/dwgsim_hip.h"
#include "dw_kernels.hpp"
#include "dw_launch.hpp"
#include "dw_mutin.hpp"

using namespace dw;

namespace {

struct HostIns { std::vector<int32_t> pos; std::vector<uint32_t> len, off; std::vector<uint8_t> bases; };

struct Member {                      // one contig of a group
    std::string name;
    int64_t l = 0, l_place = 0;      // length; fragment-placement length (region length with -x)
    uint32_t contig_index = 0;       // ordinal in the FASTA (RNG key)
    int32_t start = 0;               // first cell in the group's coordinate space
    int32_t reg_off = 0, n_reg = 0;  // -x: [starts | ends] of this contig in the group's region pool
    uint32_t name_off = 0; int32_t name_fixed_len = 0;      // "@[prefix_]name" in the group's name pool
    ResolvedContig rc;               // -m / -b / -v: the file's entries for this contig, resolved when the contig was added
};

struct Group {
    bool alive = false, mutated = false, walk_pending = false;
    int first_handle = -1;
    std::vector<Member> m;
    int64_t total = 0;               // cells of the coordinate space (a multiple of 16; allocations add CELL_PAD = 64: a whole 64-cell chunk past the last cell stays inside)
    uint8_t *d_ref = nullptr, *d_cells[2] = {nullptr, nullptr}, *d_view[2] = {nullptr, nullptr};      // reference codes, byte cells, 4-bit read views
    int32_t *d_ins_pos[2] = {nullptr, nullptr};
    uint32_t *d_ins_len[2] = {nullptr, nullptr}, *d_ins_off[2] = {nullptr, nullptr};
    uint8_t *d_ins_bases[2] = {nullptr, nullptr};
    uint32_t n_ins[2] = {0, 0}, n_ins_bases[2] = {0, 0};
    size_t cap_ins[2] = {0, 0}, cap_bases[2] = {0, 0};
    uint8_t *d_names = nullptr; int32_t *d_reg = nullptr; int32_t *d_seg = nullptr;      // name pool, region pool, segment table (start[n + 1] | len[n] | cindex[n])
    std::vector<uint8_t> h_names; std::vector<int32_t> h_reg, h_seg;                     // their host sources (alive while asynchronous copies may read them)
    int fixed_max = 0;               // longest "[prefix_]name"
    uint32_t n_cand = 0;
    uint16_t *d_summ[2] = {nullptr, nullptr}, *d_summ2[2] = {nullptr, nullptr};      // haplotype summaries (per 64 and per 1024 cells) for count_random: kept in step with the read views
    // the walk as sparse work (dw_walk.hip k_mark_dirty / k_dirty_chunks): the view and the summaries of the UNMUTATED group, made once at upload, and
    // the bitmap of 64-cell chunks the last walk may have written (dirty_any) -- or "everything" after a walk that kept no bitmap (dirty_all)
    uint8_t *d_refview = nullptr; uint16_t *d_refsumm = nullptr, *d_refsumm2 = nullptr; uint32_t *d_dirty = nullptr; uint32_t n_dirty_words = 0;
    bool dirty_any = false, dirty_all = false;
    // what the allocations above were made for (a dropped group's memory is kept for the next one: dwgsim_hip_ctx::pool)
    size_t cap_cells = 0, cap_names = 0, cap_reg = 0, cap_seg = 0;
    // a walk that was enqueued and not yet waited for
    int walk_attempt = 0; uint32_t walk_cap = 0; size_t walk_cap_bases = 0; bool walk_reset = false;
    uint32_t n_patch = 0, n_patch_ev = 0;       // file-driven mutations: patched cells / indel events
    hipEvent_t ev_walk = nullptr, ev_walk0 = nullptr;      // end / start of the walk chain on the walk stream
    uint64_t *h_wc = nullptr;        // page-locked mirror of the walk's counters (the context's d_wcounters) at the end of THIS group's walk: several groups' walks can be in flight
    // the mutated cells of the finished walk, fetched once for mutations_text
    bool list_valid = false; std::vector<int32_t> pos; std::vector<uint32_t> cells; HostIns ins[2];
};

struct HandleRef { int group = -1, k = 0; };

struct DevBuf {                     // grow-only device buffer
    void *p = nullptr; size_t cap = 0;
};

constexpr int N_COUNTERS = 32;      // u64 words of a counter block (SimArgs::counters)

struct Slot {                       // one of the two batches a context can have in flight
    uint64_t *d_counters = nullptr, *h_counters = nullptr;      // device block + pinned mirror
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_end = nullptr, ev_done = nullptr, ev_fetched = nullptr;
    bool pending = false, empty = true, fetch_in_flight = false;
    int group = -1;                 // the group the batch in flight reads
    uint64_t n_pairs = 0, out_bytes[3] = {0, 0, 0}, gz_bytes[3] = {0, 0, 0};
    DevBuf gz_out[3], gz_status, segs;        // GPU gzip: the members of each stream, look-back words; the range table of the launch
    SimSeg *h_segs = nullptr; size_t h_segs_cap = 0;      // ... and its page-locked source
    std::vector<dwgsim_hip_range_t> ranges;      // what the batch in flight covers (it is enqueued again, with larger read buffers, when an Ion Torrent read outgrew them)
    uint64_t *d_rerun_chain = nullptr;           // [2]: the chain words such a second run starts from
    int cap_mult = 1;                            // the context's flow_cap_mult this batch was enqueued with
};

} // namespace

struct dwgsim_hip_ctx {
    dwgsim_hip_params_t prm;
    std::string read_prefix;
    std::vector<uint8_t> flow;            // Ion Torrent flow order as base codes
    uint8_t *d_flow = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;            // simulate: kernels of the batches
    hipStream_t copy_stream = nullptr;       // device -> host copies of finished text
    hipStream_t walk_stream = nullptr;       // uploads, the mutation walk, the mutated-cell list: a group can be prepared while another one is being simulated
    hipStream_t count_stream = nullptr;      // count_random (k_place ..): behind the walk of ITS group only (an event), not behind the walks of groups issued later --
                                             // on the walk stream the count of step k+1 waited for the walks of steps k+2, k+3 (round 6: profiles/r06_solo_rank_entry.txt)
    std::string err;
    double e_by[2] = {0, 0};
    uint64_t *d_thr[2] = {nullptr, nullptr};
    uint32_t *d_thr32[2] = {nullptr, nullptr}; int e_full = 0;
    uint32_t *d_qbase[2] = {nullptr, nullptr}; int32_t qb_words = 1;
    uint8_t *d_rand_fixed = nullptr; int32_t rand_fixed_len = 0;
    std::vector<Group> groups;
    std::vector<Group> pool;                 // the device memory, events and page-locked mirrors of dropped groups, handed to the next add_contigs whose cells fit: a
                                             // job of many groups pays its hipMalloc / hipFree (each a device-wide synchronisation) once, not at every group end
    std::vector<HandleRef> handles;          // contig handle -> (group, member); handles are never reused
    // simulate() working set
    DevBuf meta, fail_summ, block_rand, status_all, out[DWGSIM_HIP_SLOTS][3], split_state, split_hand, split_agg, split_pre, split_chunk;
    // walk-stream working set (grow-only)
    DevBuf w_slots, w_slot_aux;      // the site scan's per-block slots (dw_walk.hip k_site_scan_slots) and their counts / bases
    int site_slots = -1; int64_t site_slot_cap = -1;      // "site_slots": -1 choose, 0 the look-back form, 1 the slot form; "site_slot_cap": a slot size to start from (tests: the overflow re-run)
    DevBuf scratch_mask, scratch_cnt, scratch_status, w_cand, w_ev, w_flags, w_lo, w_sufmin, w_bound, w_ppos, w_pcells, up_ascii, l_pos, l_cells, place_segs, place_rand, place_list, place_aux;
    uint8_t *h_up = nullptr; size_t h_up_cap = 0; hipEvent_t ev_up = nullptr; bool up_in_flight = false;      // page-locked staging of a group's sequence
    SimSeg *h_place_segs = nullptr; size_t h_place_segs_cap = 0;
    uint64_t *h_range_rand = nullptr; size_t h_range_rand_cap = 0;      // page-locked: count_random's result per range
    std::vector<int32_t> h_ppos; std::vector<uint16_t> h_pcells; std::vector<Event> h_pev;      // file-driven mutations of the group being walked
    bool seq_justify = false, dense_view = false;      // "justify_seq", "dense_view": the cross-check forms of the walk (one thread justifies a whole group; the views are made from every cell)
    MutInput mutin; bool has_mutin = false;                             // -m / -b / -v
    Regions regions; bool has_regions = false;                           // -x
    DevBuf flow_scratch, flow_free;
    uint64_t *d_counters = nullptr, *h_counters = nullptr;          // N_COUNTERS x u64 + pinned mirror: calibrate / count_random / debug hooks (compute stream)
    uint64_t *d_wcounters = nullptr;                                // 16 x u64 (mirrored per group: Group::h_wc): the walk ([7] candidates, [8..11] eight words, [12], [13] mut_debug, [14] listed cells)
    uint64_t *d_pcounters = nullptr, *h_pcounters = nullptr;        // N_COUNTERS x u64 + pinned mirror: count_random (walk stream)
    Slot slot[DWGSIM_HIP_SLOTS];             // simulate(): up to three batches in flight (kernels | copy-out issued | copy-out landing: dw_job.cpp)
    uint64_t *d_chain = nullptr;             // [0] random reads emitted before the next batch, [1] the abort rule's carry: handed from batch to batch on the device
    int chain_contig = -1; uint64_t chain_next_ii = 0;      // which (contig, read index) the carry continues
    bool has_carry_override = false; uint64_t carry_override = 0;
    int flow_cap_forced = 0;               // "flow_cap" (tests): the capacity a job starts from, instead of flow_read_capacity()
    int flow_cap_mult = 1;                 // Ion Torrent: the read capacity is flow_read_capacity() times this; doubled when a read outgrew it (the reference doubles its buffers, dwgsim.c:296-311)
    int ion_lds = -1;                      // "ion_lds" (tests): where the Ion Torrent read buffers live (fill_sim_args)
    int n_cu = 0; int flow_slots = 0;      // compute units of the device; "flow_slots": scratch slots per XCD forced by the tests (0: as many as an XCD can hold blocks)
    int64_t walk_cap = -1; bool phases = false; int writer = -1, force_threads = 0; int64_t place_cap = -1; uint64_t place_open = 0; double walk_us = 0, count_us = 0; int split = -1;      // dwgsim_hip_debug_option / _debug_get
    hipEvent_t ev_cnt0 = nullptr, ev_cnt1 = nullptr;
    bool gzip_on = false; uint32_t *d_crc_table = nullptr, *d_crc_shift = nullptr;      // dwgsim_hip_set_gzip
    void *h_stage = nullptr; size_t h_stage_cap = 0;   // pinned staging for fetch
    std::string txt, vcf;
};

namespace {

#define HIPC(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char b_[512]; snprintf(b_, sizeof b_, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); (ctx)->err = b_; return DWGSIM_HIP_ERR_DEVICE; } } while (0)

void free_group(struct Group &g);

// hipMalloc; when the device is out of memory the sets of dropped groups the context keeps for reuse (dwgsim_hip_ctx::pool) are given back and the
// allocation is tried once more -- a job that fitted when groups were freed at drop (rounds 1-4) must not fail with gigabytes idle in the pool
hipError_t dev_malloc(dwgsim_hip_ctx *c, void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess || c->pool.empty()) return e;
    (void)hipGetLastError();
    hipStreamSynchronize(c->walk_stream);
    for (auto &g : c->pool) free_group(g);
    c->pool.clear();
    return hipMalloc(p, bytes);
}

int ensure(dwgsim_hip_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) HIPC(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    HIPC(c, dev_malloc(c, &b.p, want));
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
    *k = (float)(sqrt(two_ln2) * sigma);
    *eps = (float)(1.5 * sigma * (3.4e-7 / sqrt(lm) + 7.2e-6) + 0x1p-18);       // (+inf for an absurd -Q: every value then takes the exact path)
}

WalkParams walk_params(const dwgsim_hip_ctx *c)
{
    WalkParams w; w.mut_rate = c->prm.mut_rate; w.indel_frac = c->prm.indel_frac; w.indel_extend = c->prm.indel_extend;
    w.indel_min = c->prm.indel_min; w.is_hap = c->prm.is_hap; w.seed = (uint32_t)c->prm.seed;
    w.mut_thr = !(c->prm.mut_rate > 0) ? 0 : c->prm.mut_rate >= 1.0 ? 0x100000000ull : (uint64_t)ceil(c->prm.mut_rate * 4294967296.0);   // exact scaling by 2^32
    flow_gap_params(w.mut_thr, &w.gap_r, &w.gap_s);
    w.lg = reinterpret_cast<const uint32_t *>(c->d_flow + 64);      // (behind the flow order: made in dwgsim_hip_create for every context)
    return w;
}

size_t padded_cells(const Group &g) { return (size_t)g.total + CELL_PAD; }

SegTab seg_tab(const Group &g)
{
    const int n = (int)g.m.size();
    SegTab t; t.start = g.d_seg; t.len = g.d_seg + (n + 1); t.cindex = reinterpret_cast<const uint32_t *>(g.d_seg + (2 * n + 1)); t.n = n;
    return t;
}

void fill_haps(const Group &g, HapDev (&hap)[2])
{
    for (int h = 0; h < 2; ++h) {
        hap[h].cells = g.d_cells[h]; hap[h].view = g.d_view[h]; hap[h].ins_pos = g.d_ins_pos[h]; hap[h].ins_len = g.d_ins_len[h];
        hap[h].ins_off = g.d_ins_off[h]; hap[h].ins_bases = g.d_ins_bases[h]; hap[h].n_ins = g.n_ins[h]; hap[h].pos_off = 0;
    }
}

ContigDev group_dev(const Group &g)
{
    ContigDev d;
    fill_haps(g, d.hap);
    d.ref = g.d_ref; d.l = g.total; d.seg = seg_tab(g);
    d.tot4 = nullptr; d.cap_bases[0] = d.cap_bases[1] = 0;
    return d;
}

void free_group(Group &g)
{
    hipFree(g.d_ref);
    for (int h = 0; h < 2; ++h) { hipFree(g.d_cells[h]); hipFree(g.d_view[h]); hipFree(g.d_ins_pos[h]); hipFree(g.d_ins_len[h]); hipFree(g.d_ins_off[h]); hipFree(g.d_ins_bases[h]); hipFree(g.d_summ[h]); hipFree(g.d_summ2[h]); }
    hipFree(g.d_refview); hipFree(g.d_refsumm); hipFree(g.d_refsumm2); hipFree(g.d_dirty);
    hipFree(g.d_names); hipFree(g.d_reg); hipFree(g.d_seg);
    if (g.ev_walk) hipEventDestroy(g.ev_walk);
    if (g.ev_walk0) hipEventDestroy(g.ev_walk0);
    if (g.h_wc) hipHostFree(g.h_wc);
    g = Group();
}

// contig handle -> its group and member (nullptr + error text for a handle that is unknown or was dropped)
Group *get_group(dwgsim_hip_ctx_t *c, int contig, int *k = nullptr)
{
    if (!c) return nullptr;
    if (contig < 0 || (size_t)contig >= c->handles.size() || c->handles[(size_t)contig].group < 0) { c->err = "unknown contig handle"; return nullptr; }
    const HandleRef r = c->handles[(size_t)contig];
    Group &g = c->groups[(size_t)r.group];
    if (!g.alive) { c->err = "unknown contig handle"; return nullptr; }
    if (k) *k = r.k;
    return &g;
}

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

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
    for (int i = 0; i < 2; ++i) {      // dwgsim_opt.c:329-344
        if (p->e_start[i] < 0.0 || 1.0 < p->e_start[i]) { if (msg) snprintf(msg, cap, "End %s: the start error is out of range (-e)\n", i ? "two" : "one"); return DWGSIM_HIP_ERR_ARG; }
        if (p->e_end[i] < 0.0 || 1.0 < p->e_end[i]) { if (msg) snprintf(msg, cap, "End %s: the end error is out of range (-e)\n", i ? "two" : "one"); return DWGSIM_HIP_ERR_ARG; }
        if (p->data_type == 2 && p->e_end[i] != p->e_start[i]) { if (msg) snprintf(msg, cap, "End %s: a uniform error rate must be given for Ion Torrent data\n", i ? "two" : "one"); return DWGSIM_HIP_ERR_ARG; }
    }
    CHK(p->mut_rate, 0, 1.0, "-r"); CHK(p->indel_frac, 0, 1.0, "-R"); CHK(p->indel_extend, 0, 1.0, "-X");
    CHK(p->indel_min, 1, INT32_MAX, "-I"); CHK(p->data_type, 0, 2, "-c"); CHK(p->strandedness, 0, 2, "-S");
    CHK(p->read_one_strand, 0, 2, "-A"); CHK(p->max_n, 0, INT32_MAX, "-n"); CHK(p->rand_read, 0, 1.0, "-y");
    if (p->data_type == 2 && !p->flow_order) { if (msg) snprintf(msg, cap, "Error: command line option -f is required\n"); return DWGSIM_HIP_ERR_ARG; }
    CHK(p->use_base_error, 0, 1, "-B"); CHK(p->is_hap, 0, 1, "-H");
    if (p->fixed_quality < -1 || p->fixed_quality > 255) { if (msg) snprintf(msg, cap, "Error: command line option -q requires one character\n"); return DWGSIM_HIP_ERR_ARG; }      // (-1: none; dwgsim_opt.c:364-367)
    CHK(p->quality_std, 0, INT32_MAX, "-Q"); CHK(p->reads_output_type, 0, 2, "-o");
    // (not checked by the reference, which treats every other -M as 0 and every non-zero -a as set; the library wants them clean)
    CHK(p->output_type, 0, 2, "-M"); CHK(p->amplicons, 0, 1, "-a");
    if (p->data_type == 2) {       // dwgsim_opt.c:396-413
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
    if (p->amplicons == 1) { if (l < max_len) return DWGSIM_HIP_SKIP_AMPLICON; }                                          // #2 :596-603
    else if (0 < size1 && l < p->dist + 3 * p->std_dev) return DWGSIM_HIP_SKIP_SHORT_INSERT;                                 // #3 :605-611
    else if (l < size0 || (0 < size1 && l < size1)) return DWGSIM_HIP_SKIP_SHORT_READ;                                     // #4 :612-618
    return n_pairs < 0 ? DWGSIM_HIP_SKIP_NO_PAIRS : n_pairs;
}

int dwgsim_hip_device_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);      // sysfs spells the bus id in lower case
    char path[160]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
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

int dwgsim_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
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

int64_t dwgsim_hip_group_layout(const int64_t *lens, int n, int64_t *starts)
{
    int64_t at = 0;
    for (int k = 0; k < n; ++k) {
        if (!lens || lens[k] < 0) return -1;
        at = align_up(at, GROUP_ALIGN);
        if (starts) starts[k] = at;
        at += lens[k];
    }
    return align_up(at, 16);
}

// Ion Torrent: room for a read after the flow model.  Every empty flow in front of a base inserts Geometric(e) bases, and inserted bases are
// examined again.  How many empty flows a base has in front of it is a property of the flow order: m = the mean distance from a flow to the next
// flow of a given base (about 2.5 for the usual 32-flow orders, 1.5 for TACG, but 15 for an order that keeps three bases away for 37 flows).  The
// mean growth is g = m e / (1 - e) per base, with the cascade 1 / (1 - g); two and a half times that (m is taken as at least 3, twice what the
// usual orders have) plus slack keeps overflow out of reach for realistic error rates -- and what does overflow is run again with twice the room
// (dwgsim_hip_wait), never written out of bounds.  The room is LDS (dw_read.hpp flow_errors: one in-place buffer of cap / 4 bytes per lane), so it is
// not handed out as generously as rounds 1-4 did with global scratch (len + 64 + 4 len g).
static int flow_read_capacity(int len, double e, const std::vector<uint8_t> &flow)
{
    const double ec = !(e > 0) ? 0 : e > 0.9 ? 0.9 : e;       // (NaN, e.g. -B with -e 0: no flow errors at all)
    double m = 3.0;
    const int F = (int)flow.size();
    if (F > 0) {
        double sum = 0;
        for (int f = 0; f < F; ++f) for (uint8_t b = 0; b < 4; ++b) { int k = 0, g = f; while (flow[(size_t)g] != b && k < F) { ++k; g = g + 1 == F ? 0 : g + 1; } sum += k; }
        m = sum / (4.0 * F);
        if (m < 3.0) m = 3.0;
    }
    double g = m * ec / (1.0 - ec);
    g = g / (1.0 - (g < 0.75 ? g : 0.75));
    return len + 32 + (int)(len * 2.5 * g);
}

static int set_err(int *err, int v) { if (err) *err = v; return v; }

dwgsim_hip_ctx_t *dwgsim_hip_create(const dwgsim_hip_params_t *p, int device, int *err)
{
    char msg[512];
    int rc = dwgsim_hip_params_check(p, msg, sizeof msg);
    if (rc != DWGSIM_HIP_OK) { fprintf(stderr, "%s", msg); set_err(err, rc); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "dwgsim-hip: no usable HIP device (requested %d of %d); the hot path has no CPU fallback\n", device, ndev);
        set_err(err, DWGSIM_HIP_ERR_DEVICE); return nullptr;
    }
    dwgsim_hip_ctx *c = new dwgsim_hip_ctx();
    c->prm = *p;
    if (p->read_prefix) c->read_prefix = p->read_prefix;
    if (p->data_type == 2) for (const char *q = p->flow_order; *q; ++q) c->flow.push_back(nt4((unsigned char)*q));
    c->prm.read_prefix = nullptr; c->prm.flow_order = nullptr;
    c->device = device;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) c->n_cu = pr.multiProcessorCount; if (c->n_cu <= 0) c->n_cu = 256; }
    int fail_code = DWGSIM_HIP_ERR_DEVICE;       // (the -B calibration reports reads that outgrow their buffers as DWGSIM_HIP_ERR_FAILED: an option set, not the device)
    auto fail = [&](const char *what) { fprintf(stderr, "dwgsim-hip: %s: %s\n", what, c->err.c_str()); set_err(err, fail_code); dwgsim_hip_destroy(c); return (dwgsim_hip_ctx *)nullptr; };
    auto init = [&]() -> int {
        HIPC(c, hipSetDevice(device));
        // What prepares the NEXT group (upload, walk, random-read count: walk stream) runs at the batches' own priority.  Rounds 3-5 put it below them ("fills the
        // thinning tail of every launch instead of taking slots"): its fifteen small kernels then ran one after the other in the gap between two k_simulate launches --
        // 0.07 of the 0.46 ms of an E. coli-sized step.  At equal priority they slip in as slots fall free while the big kernel runs and are done when it ends: E. coli-sized
        // 1 146-1 159 against 980-1 060 M pairs/s, chr20-sized / whole genome / N-rank weak lines unchanged (profiles/r06_bench_lines_final.txt 15).
        int prio_least = 0, prio_greatest = 0;
        HIPC(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        // (DWGSIM_HIP_WALK_PRIO=low|mid|above: analysis -- the walk stream below the batches (rounds 3-5), between the two, or the batches below the walk)
        int prio_walk = prio_greatest, prio_batch = prio_greatest;
        if (const char *e = getenv("DWGSIM_HIP_WALK_PRIO")) {
            prio_walk = !strcmp(e, "high") || !strcmp(e, "above") ? prio_greatest : !strcmp(e, "mid") ? (prio_least + prio_greatest) / 2 : prio_least;
            if (!strcmp(e, "above")) prio_batch = prio_least;
        }
        HIPC(c, hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_batch));
        HIPC(c, hipStreamCreateWithPriority(&c->copy_stream, hipStreamDefault, prio_greatest));
        HIPC(c, hipStreamCreateWithPriority(&c->walk_stream, hipStreamDefault, prio_walk));
        HIPC(c, hipStreamCreateWithPriority(&c->count_stream, hipStreamDefault, prio_walk));
        HIPC(c, hipEventCreate(&c->ev_up)); HIPC(c, hipEventCreate(&c->ev_cnt0)); HIPC(c, hipEventCreate(&c->ev_cnt1));
        HIPC(c, hipMalloc((void **)&c->d_counters, N_COUNTERS * sizeof(uint64_t)));
        HIPC(c, hipHostMalloc((void **)&c->h_counters, N_COUNTERS * sizeof(uint64_t), hipHostMallocDefault));
        HIPC(c, hipMalloc((void **)&c->d_wcounters, 16 * sizeof(uint64_t)));
        HIPC(c, hipMalloc((void **)&c->d_pcounters, N_COUNTERS * sizeof(uint64_t)));
        HIPC(c, hipHostMalloc((void **)&c->h_pcounters, N_COUNTERS * sizeof(uint64_t), hipHostMallocDefault));
        HIPC(c, hipMalloc((void **)&c->d_chain, 4 * sizeof(uint64_t)));
        HIPC(c, hipMemset(c->d_chain, 0, 4 * sizeof(uint64_t)));
        for (Slot &sl : c->slot) {
            HIPC(c, hipMalloc((void **)&sl.d_counters, N_COUNTERS * sizeof(uint64_t)));
            HIPC(c, hipHostMalloc((void **)&sl.h_counters, N_COUNTERS * sizeof(uint64_t), hipHostMallocDefault));
            HIPC(c, hipMalloc((void **)&sl.d_rerun_chain, 2 * sizeof(uint64_t)));
            HIPC(c, hipEventCreate(&sl.ev_k0)); HIPC(c, hipEventCreate(&sl.ev_k1)); HIPC(c, hipEventCreate(&sl.ev_end)); HIPC(c, hipEventCreate(&sl.ev_done)); HIPC(c, hipEventCreate(&sl.ev_fetched));
        }
        {   // the flow order (64 bytes) and, behind it, the log2 table the flow model's gap draws interpolate in (dw_kernels.hpp flow_log2_table)
            std::vector<uint8_t> fl(64 + sizeof(uint32_t) * FLOW_LG_ENTRIES, 4); for (size_t i = 0; i < c->flow.size() && i < 64; ++i) fl[i] = c->flow[i];
            flow_log2_table(reinterpret_cast<uint32_t *>(fl.data() + 64));
            HIPC(c, hipMalloc((void **)&c->d_flow, fl.size())); HIPC(c, hipMemcpy(c->d_flow, fl.data(), fl.size(), hipMemcpyHostToDevice));
        }
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
                flow_gap_params(ca.thr, &ca.gap_r, &ca.gap_s);
                ca.flow = c->d_flow; ca.flow_len = (int32_t)c->flow.size();
                // a read that outgrows its buffers: once more with twice the room (the reference doubles its buffers, dwgsim.c:296-311) -- by the rule of
                // dwgsim_hip_wait: up to FLOW_CAP_MAX bases per read.  The scratch holds a CHUNK of the 10^6 reads (at most ~2 GiB), so the room a read
                // may take does not depend on how many reads there are (round 5 stopped at 16 x: -B then refused flow orders the simulate path handles)
                const int64_t base_cap = flow_read_capacity(len, e, c->flow);
                int mult = 1;
                for (;; mult *= 2) {
                    ca.lds_words = (int32_t)((base_cap * mult + 15) / 16);       // the flow model's one buffer per lane: 16 bases per word
                    ca.stack_words = std::min(FLOW_STACK_WORDS * mult, FLOW_STACK_WORDS_MAX);
                    const size_t per_block = (size_t)ca.lds_words * PAIRS_PER_BLOCK * sizeof(uint32_t);
                    const size_t nblk_all = (size_t)((ca.n_reads + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK);
                    const size_t nblk = std::max<size_t>(1, std::min(nblk_all, ((size_t)2 << 30) / per_block));
                    ca.chunk_reads = (uint64_t)nblk * PAIRS_PER_BLOCK;
                    if (ensure(c, c->flow_scratch, per_block * nblk)) return -1;
                    ca.scratch = (uint32_t *)c->flow_scratch.p; ca.counters = c->d_counters;
                    HIPC(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(uint64_t), c->stream));
                    for (ca.first_read = 0; ca.first_read < ca.n_reads; ca.first_read += ca.chunk_reads) launch_calibrate(c->stream, ca);
                    HIPC(c, hipMemcpyAsync(c->h_counters, c->d_counters, N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
                    HIPC(c, hipStreamSynchronize(c->stream));
                    if (!c->h_counters[2] || base_cap * (int64_t)mult * 2 > (int64_t)FLOW_CAP_MAX) break;
                }
                // the job's reads will need the room the calibration's needed: start there instead of running the first overflowing batch twice
                if (mult > c->flow_cap_mult) c->flow_cap_mult = mult;
                if (c->h_counters[2]) { c->err = "-B calibration: a read outgrew its flow-space buffer (the flow model's growth at this error rate and flow order: INTEGRATION.md)"; fail_code = DWGSIM_HIP_ERR_FAILED; return -1; }
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
            {   // packed table: n entries, then the last one repeated (>= 8 copies), the same word count for both read ends
                const int lmax = c->prm.length[0] > c->prm.length[1] ? c->prm.length[0] : c->prm.length[1];
                c->qb_words = (lmax + 4 + 3) / 4 + 2;      // (a block of the quality stream looks at eight positions: three words from any position <= lmax)
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
        // device copy: '@' + "[prefix_]rand", zero padded to >= 256 + 16 bytes (the kernel stages 128 bytes in LDS)
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

// Not part of the drop-in ABI either: the number formatters of the name line (dw_read.hpp put_dec / put_hex) against one division per digit, on the
// values first + i * stride, i < n (dw_simulate.hip k_selftest_text).  out[0] / out[1] = decimal / hexadecimal texts that differ, out[2] = values compared.
extern "C" int dwgsim_hip_selftest_text(int device, uint64_t first, uint64_t n, uint64_t stride, uint64_t *out)
{
    if (!out || hipSetDevice(device) != hipSuccess) return DWGSIM_HIP_ERR_DEVICE;
    uint64_t *d = nullptr;
    if (hipMalloc((void **)&d, 4 * sizeof(uint64_t)) != hipSuccess) return DWGSIM_HIP_ERR_NOMEM;
    hipMemset(d, 0, 4 * sizeof(uint64_t));
    launch_selftest_text(nullptr, first, n, stride, d);
    const hipError_t e = hipMemcpy(out, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? DWGSIM_HIP_OK : DWGSIM_HIP_ERR_DEVICE;
}

// Not part of the drop-in ABI either: self-test of the lazy quality normals (dw_simulate.hip k_selftest_lazy).  out[0..4] = counters of the
// comparison with the exact form on the n tries w = first, first + 1, ... (n = 2^32: EVERY try) at quality_std = sigma, out[5] = max
// |estimate - exact| / eps, out[6..8] = worst error of v_log_f32 / v_rcp_f32 / v_sqrt_f32 over EVERY float of their operand ranges, in units of
// the bounds the error budget assumes (doubles).
extern "C" int dwgsim_hip_selftest_lazy(int device, uint32_t first, uint64_t n, double sigma, int exhaustive, uint64_t *out)
{
    if (!out || hipSetDevice(device) != hipSuccess) return DWGSIM_HIP_ERR_DEVICE;
    uint64_t *d = nullptr;
    if (hipMalloc((void **)&d, 12 * sizeof(uint64_t)) != hipSuccess) return DWGSIM_HIP_ERR_NOMEM;
    hipMemset(d, 0, 12 * sizeof(uint64_t));
    float qk, qeps, qlmin; int32_t qnear1;
    lazy_quality_params(sigma, &qk, &qeps, &qlmin, &qnear1);
    launch_selftest_lazy(nullptr, 0, first, n, sigma, qk, qeps, qlmin, qnear1, d);
    if (exhaustive) {
        launch_selftest_lazy(nullptr, 1, 0, 0x3F800000u - 0x30800000u, 0, 0, 0, 0, 0, d);           // [2^-30, 1)
        launch_selftest_lazy(nullptr, 2, 0, 0x4E800000u - 0x3F800000u + 1u, 0, 0, 0, 0, 0, d);      // [1, 2^30]
        launch_selftest_lazy(nullptr, 3, 0, 0x42000000u - 0x2B000000u, 0, 0, 0, 0, 0, d);           // [2^-41, 2^5)
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
    if (c->walk_stream) hipStreamSynchronize(c->walk_stream);
    if (c->count_stream) hipStreamSynchronize(c->count_stream);
    for (auto &g : c->groups) if (g.alive) free_group(g);
    for (auto &g : c->pool) free_group(g);
    for (int j = 0; j < 2; ++j) { hipFree(c->d_thr[j]); hipFree(c->d_thr32[j]); hipFree(c->d_qbase[j]); }
    for (DevBuf *b : {&c->meta, &c->fail_summ, &c->block_rand, &c->status_all, &c->split_state, &c->split_hand, &c->split_agg, &c->split_pre, &c->split_chunk, &c->place_segs, &c->place_rand, &c->place_list, &c->place_aux, &c->scratch_mask, &c->scratch_cnt, &c->scratch_status, &c->w_cand, &c->w_ev, &c->w_flags, &c->w_lo, &c->w_sufmin,
                      &c->w_bound, &c->w_ppos, &c->w_pcells, &c->up_ascii, &c->l_pos, &c->l_cells, &c->flow_scratch, &c->flow_free, &c->w_slots, &c->w_slot_aux}) hipFree(b->p);
    for (int s = 0; s < DWGSIM_HIP_SLOTS; ++s) for (int t = 0; t < 3; ++t) hipFree(c->out[s][t].p);
    hipFree(c->d_rand_fixed); hipFree(c->d_counters); hipFree(c->d_wcounters); hipFree(c->d_pcounters); hipFree(c->d_flow); hipFree(c->d_chain); hipFree(c->d_crc_table); hipFree(c->d_crc_shift);
    if (c->h_counters) hipHostFree(c->h_counters);
    if (c->h_pcounters) hipHostFree(c->h_pcounters);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_up) hipHostFree(c->h_up);
    if (c->h_place_segs) hipHostFree(c->h_place_segs);
    if (c->h_range_rand) hipHostFree(c->h_range_rand);
    if (c->ev_up) hipEventDestroy(c->ev_up);
    if (c->ev_cnt0) hipEventDestroy(c->ev_cnt0);
    if (c->ev_cnt1) hipEventDestroy(c->ev_cnt1);
    for (Slot &sl : c->slot) {
        hipFree(sl.d_counters); hipFree(sl.d_rerun_chain); hipFree(sl.gz_status.p); hipFree(sl.segs.p);
        for (int t = 0; t < 3; ++t) hipFree(sl.gz_out[t].p);
        if (sl.h_counters) hipHostFree(sl.h_counters);
        if (sl.h_segs) hipHostFree(sl.h_segs);
        for (hipEvent_t e : {sl.ev_k0, sl.ev_k1, sl.ev_end, sl.ev_done, sl.ev_fetched}) if (e) hipEventDestroy(e);
    }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->walk_stream) hipStreamDestroy(c->walk_stream);
    if (c->count_stream) hipStreamDestroy(c->count_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *dwgsim_hip_last_error(const dwgsim_hip_ctx_t *c) { return c ? c->err.c_str() : "no context"; }

int dwgsim_hip_get_params(const dwgsim_hip_ctx_t *c, dwgsim_hip_params_t *out)
{
    if (!c || !out) return DWGSIM_HIP_ERR_ARG;
    *out = c->prm;      // (read_prefix / flow_order were cleared when the context took its copies)
    return DWGSIM_HIP_OK;
}

static bool in_host_registry(const void *p);
static bool is_page_locked(const void *p)
{
    if (in_host_registry(p)) return true;      // (dwgsim_hip_host_alloc)
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) return true;
    (void)hipGetLastError();      // pageable memory: the query reports an error that must not stick
    return false;
}

int dwgsim_hip_add_contigs(dwgsim_hip_ctx_t *c, int n, const char *const *names, const uint8_t *const *ascii, const int64_t *lens, const uint32_t *contig_index)
{
    if (!c || n < 1 || !names || !ascii || !lens || !contig_index) { if (c) c->err = "bad contig arguments"; return DWGSIM_HIP_ERR_ARG; }
    for (int k = 0; k < n; ++k) if (!names[k] || (!ascii[k] && lens[k] > 0) || lens[k] < 0 || lens[k] > INT32_MAX) { c->err = "bad contig arguments"; return DWGSIM_HIP_ERR_ARG; }
    std::vector<int64_t> starts((size_t)n);
    const int64_t total = dwgsim_hip_group_layout(lens, n, starts.data());
    if (total < 0 || total > (int64_t)INT32_MAX - 2 * GROUP_ALIGN) { c->err = "dwgsim-hip: the contigs of one group must stay below 2^31 cells in all (add them in smaller groups)\n"; return DWGSIM_HIP_ERR_ARG; }
    HIPC(c, hipSetDevice(c->device));
    int gid = -1;
    for (size_t i = 0; i < c->groups.size(); ++i) if (!c->groups[i].alive) { gid = (int)i; break; }
    if (gid < 0) { c->groups.emplace_back(); gid = (int)c->groups.size() - 1; }
    Group &g = c->groups[(size_t)gid];
    g = Group();
    {   // memory of a dropped group, if one is large enough (the smallest such; none more than four times too large)
        const size_t want = (size_t)total + CELL_PAD;
        int best = -1;
        for (size_t i = 0; i < c->pool.size(); ++i) if (c->pool[i].cap_cells >= want && c->pool[i].cap_cells <= 4 * want + (1u << 20) && (best < 0 || c->pool[i].cap_cells < c->pool[(size_t)best].cap_cells)) best = (int)i;
        if (best >= 0) { g = c->pool[(size_t)best]; c->pool.erase(c->pool.begin() + best); }
        // what describes the group that was dropped goes; the allocations and their capacities stay
        g.mutated = g.walk_pending = false; g.m.clear(); g.h_names.clear(); g.h_reg.clear(); g.h_seg.clear(); g.fixed_max = 0; g.n_cand = 0;
        g.n_ins[0] = g.n_ins[1] = g.n_ins_bases[0] = g.n_ins_bases[1] = 0; g.walk_attempt = 0; g.walk_cap = 0; g.walk_cap_bases = 0; g.walk_reset = false; g.n_patch = g.n_patch_ev = 0;
        g.list_valid = false; g.pos.clear(); g.cells.clear(); g.ins[0] = HostIns(); g.ins[1] = HostIns(); g.dirty_any = g.dirty_all = false;
    }
    g.alive = true; g.total = total; g.first_handle = (int)c->handles.size();
    g.m.resize((size_t)n);
    for (int k = 0; k < n; ++k) { Member &m = g.m[(size_t)k]; m.name = names[k]; m.l = m.l_place = lens[k]; m.contig_index = contig_index[k]; m.start = (int32_t)starts[(size_t)k]; }
    const size_t padded = padded_cells(g);
    bool synced = true;
    auto fill = [&]() -> int {
        if (!g.ev_walk) HIPC(c, hipEventCreate(&g.ev_walk));
        if (!g.ev_walk0) HIPC(c, hipEventCreate(&g.ev_walk0));
        if (!g.h_wc) HIPC(c, hipHostMalloc((void **)&g.h_wc, 16 * sizeof(uint64_t), hipHostMallocDefault));
        if (g.cap_cells < padded) {
            hipFree(g.d_ref); hipFree(g.d_refview); hipFree(g.d_refsumm); hipFree(g.d_refsumm2); hipFree(g.d_dirty);
            g.d_ref = g.d_refview = nullptr; g.d_refsumm = g.d_refsumm2 = nullptr; g.d_dirty = nullptr;
            for (int h = 0; h < 2; ++h) { hipFree(g.d_cells[h]); hipFree(g.d_view[h]); hipFree(g.d_summ[h]); hipFree(g.d_summ2[h]); g.d_cells[h] = g.d_view[h] = nullptr; g.d_summ[h] = g.d_summ2[h] = nullptr; }
            g.cap_cells = 0;
            HIPC(c, dev_malloc(c, (void **)&g.d_ref, padded));
            for (int h = 0; h < 2; ++h) {
                HIPC(c, dev_malloc(c, (void **)&g.d_cells[h], padded)); HIPC(c, dev_malloc(c, (void **)&g.d_view[h], padded / 2 + 32));
                HIPC(c, dev_malloc(c, (void **)&g.d_summ[h], sizeof(uint16_t) * (padded / SUMM_CELLS + 16))); HIPC(c, dev_malloc(c, (void **)&g.d_summ2[h], sizeof(uint16_t) * (padded / SUMM2_CELLS + 16)));
            }
            HIPC(c, dev_malloc(c, (void **)&g.d_refview, padded / 2 + 32)); HIPC(c, dev_malloc(c, (void **)&g.d_refsumm, sizeof(uint16_t) * (padded / SUMM_CELLS + 16))); HIPC(c, dev_malloc(c, (void **)&g.d_refsumm2, sizeof(uint16_t) * (padded / SUMM2_CELLS + 16)));
            HIPC(c, dev_malloc(c, (void **)&g.d_dirty, sizeof(uint32_t) * ((padded + 32 * SUMM_CELLS - 1) / (32 * SUMM_CELLS) + 2)));
            g.cap_cells = padded;
        }
        g.n_dirty_words = (uint32_t)((padded + 32 * SUMM_CELLS - 1) / (32 * SUMM_CELLS));
        if (ensure(c, c->up_ascii, padded)) return DWGSIM_HIP_ERR_DEVICE;
        uint8_t *d_ascii = (uint8_t *)c->up_ascii.p;
        // The sequence goes up on the walk stream.  One copy when the caller's buffers already are the group layout inside ONE page-locked
        // allocation (ascii[k] = ascii[0] + start[k], zero bytes between the contigs): nothing is staged and the call does not wait -- the
        // buffers must then stay as they are until dwgsim_hip_mutate_wait returned.  A few contigs: one copy each into a zeroed device buffer.
        // Many: packed into page-locked staging first (one copy instead of thousands).
        bool laid_out = true;
        for (int k = 0; k < n && laid_out; ++k) if (lens[k] > 0 && ascii[k] != ascii[0] + starts[(size_t)k]) laid_out = false;
        if (laid_out && n > 0 && ascii[0] && is_page_locked(ascii[0])) {
            HIPC(c, hipMemcpyAsync(d_ascii, ascii[0], (size_t)total, hipMemcpyHostToDevice, c->walk_stream));
            HIPC(c, hipMemsetAsync(d_ascii + total, 0, padded - (size_t)total, c->walk_stream));
            synced = false;
        } else if (n <= 8) {
            HIPC(c, hipMemsetAsync(d_ascii, 0, padded, c->walk_stream));
            for (int k = 0; k < n; ++k) if (lens[k] > 0) HIPC(c, hipMemcpyAsync(d_ascii + starts[(size_t)k], ascii[k], (size_t)lens[k], hipMemcpyHostToDevice, c->walk_stream));
        } else {
            if (c->up_in_flight) { HIPC(c, hipEventSynchronize(c->ev_up)); c->up_in_flight = false; }
            if ((size_t)total > c->h_up_cap) {
                if (c->h_up) HIPC(c, hipHostFree(c->h_up));
                c->h_up = nullptr; c->h_up_cap = 0;
                const size_t want = (size_t)total + (size_t)total / 4 + 4096;
                HIPC(c, hipHostMalloc((void **)&c->h_up, want, hipHostMallocDefault));
                c->h_up_cap = want;
            }
            for (int k = 0; k < n; ++k) {
                const int64_t end = starts[(size_t)k] + lens[k], next = k + 1 < n ? starts[(size_t)k + 1] : total;
                if (lens[k] > 0) memcpy(c->h_up + starts[(size_t)k], ascii[k], (size_t)lens[k]);
                memset(c->h_up + end, 0, (size_t)(next - end));
            }
            HIPC(c, hipMemcpyAsync(d_ascii, c->h_up, (size_t)total, hipMemcpyHostToDevice, c->walk_stream));
            HIPC(c, hipMemsetAsync(d_ascii + total, 0, padded - (size_t)total, c->walk_stream));
            HIPC(c, hipEventRecord(c->ev_up, c->walk_stream)); c->up_in_flight = true;
        }
        launch_pack(c->walk_stream, d_ascii, g.d_ref, g.d_cells[0], g.d_cells[1], (int64_t)padded & ~(int64_t)15);
        // the read views and summaries of the unmutated group, once: the pristine copies, and what both haplotypes start from (a walk then rewrites
        // only the chunks it touches)
        launch_make_view(c->walk_stream, g.d_ref, g.d_ref, (int64_t)padded & ~(int64_t)15, g.total, g.d_refview, g.d_view[0], g.d_refsumm, g.d_summ[0], g.d_refsumm2, g.d_summ2[0]);
        HIPC(c, hipMemcpyAsync(g.d_view[1], g.d_refview, padded / 2, hipMemcpyDeviceToDevice, c->walk_stream));
        HIPC(c, hipMemcpyAsync(g.d_summ[1], g.d_refsumm, sizeof(uint16_t) * (padded / SUMM_CELLS), hipMemcpyDeviceToDevice, c->walk_stream));
        HIPC(c, hipMemcpyAsync(g.d_summ2[1], g.d_refsumm2, sizeof(uint16_t) * (padded / SUMM2_CELLS), hipMemcpyDeviceToDevice, c->walk_stream));
        HIPC(c, hipMemsetAsync(g.d_dirty, 0, sizeof(uint32_t) * ((size_t)g.n_dirty_words + 2), c->walk_stream));
        HIPC(c, hipGetLastError());
        // segment table, name pool, target regions
        g.h_seg.resize((size_t)(3 * n + 1));
        for (int k = 0; k < n; ++k) { g.h_seg[(size_t)k] = g.m[(size_t)k].start; g.h_seg[(size_t)(n + 1 + k)] = (int32_t)g.m[(size_t)k].l; g.h_seg[(size_t)(2 * n + 1 + k)] = (int32_t)g.m[(size_t)k].contig_index; }
        g.h_seg[(size_t)n] = (int32_t)total;
        if (g.cap_seg < g.h_seg.size()) { hipFree(g.d_seg); g.d_seg = nullptr; g.cap_seg = 0; HIPC(c, hipMalloc((void **)&g.d_seg, sizeof(int32_t) * (g.h_seg.size() + 64))); g.cap_seg = g.h_seg.size() + 64; }
        HIPC(c, hipMemcpyAsync(g.d_seg, g.h_seg.data(), sizeof(int32_t) * g.h_seg.size(), hipMemcpyHostToDevice, c->walk_stream));
        for (int k = 0; k < n; ++k) {      // '@' + "[prefix_]name", zero padded to >= 256 + 16 bytes (the kernel stages 128 bytes in LDS), entries 16-byte aligned
            Member &m = g.m[(size_t)k];
            const std::string nf = c->read_prefix.empty() ? m.name : c->read_prefix + "_" + m.name;
            m.name_fixed_len = (int32_t)nf.size(); m.name_off = (uint32_t)g.h_names.size();
            if (m.name_fixed_len > g.fixed_max) g.fixed_max = m.name_fixed_len;
            const size_t room = ((nf.size() + 1 < 256 ? 272 : nf.size() + 1 + 16) + 15) & ~(size_t)15;
            g.h_names.resize(g.h_names.size() + room, 0);
            g.h_names[m.name_off] = '@'; memcpy(&g.h_names[m.name_off + 1], nf.data(), nf.size());
        }
        if (g.cap_names < g.h_names.size()) { hipFree(g.d_names); g.d_names = nullptr; g.cap_names = 0; HIPC(c, hipMalloc((void **)&g.d_names, g.h_names.size() + 4096)); g.cap_names = g.h_names.size() + 4096; }
        HIPC(c, hipMemcpyAsync(g.d_names, g.h_names.data(), g.h_names.size(), hipMemcpyHostToDevice, c->walk_stream));
        if (c->has_regions) {
            for (int k = 0; k < n; ++k) {
                Member &m = g.m[(size_t)k];
                std::vector<int32_t> st, en; int64_t tot = 0;
                for (size_t q = 0; q < c->regions.contig.size(); ++q) if (c->regions.contig[q] == m.contig_index) { st.push_back((int32_t)c->regions.start[q]); en.push_back((int32_t)c->regions.end[q]); tot += c->regions.end[q] - c->regions.start[q]; }
                m.reg_off = (int32_t)g.h_reg.size(); m.n_reg = (int32_t)st.size(); m.l_place = tot;
                g.h_reg.insert(g.h_reg.end(), st.begin(), st.end()); g.h_reg.insert(g.h_reg.end(), en.begin(), en.end());
            }
            g.h_reg.push_back(0);
            if (g.cap_reg < g.h_reg.size()) { hipFree(g.d_reg); g.d_reg = nullptr; g.cap_reg = 0; HIPC(c, hipMalloc((void **)&g.d_reg, sizeof(int32_t) * (g.h_reg.size() + 64))); g.cap_reg = g.h_reg.size() + 64; }
            HIPC(c, hipMemcpyAsync(g.d_reg, g.h_reg.data(), sizeof(int32_t) * g.h_reg.size(), hipMemcpyHostToDevice, c->walk_stream));
        }
        // -m / -b / -v: the file's entries for these contigs are resolved now, while the sequence is at hand (mut.c:644-745)
        if (c->has_mutin) for (int k = 0; k < n; ++k) resolve_mutation_input(c->mutin, g.m[(size_t)k].contig_index, ascii[k], lens[k], (uint32_t)c->prm.seed, c->prm.is_hap != 0, g.m[(size_t)k].rc);
        if (synced) HIPC(c, hipStreamSynchronize(c->walk_stream));
        return 0;
    };
    const int rc = fill();
    if (rc != 0) { (void)hipStreamSynchronize(c->walk_stream); free_group(g); return rc; }      // no half-built group stays behind a failed call
    for (int k = 0; k < n; ++k) c->handles.push_back(HandleRef{gid, k});
    return g.first_handle;
}

int dwgsim_hip_add_contig(dwgsim_hip_ctx_t *c, const char *name, const uint8_t *ascii, int64_t len, uint32_t contig_index)
{
    return dwgsim_hip_add_contigs(c, 1, &name, &ascii, &len, &contig_index);
}

int dwgsim_hip_drop_contig(dwgsim_hip_ctx_t *c, int contig)
{
    Group *g = get_group(c, contig);
    if (!g) return DWGSIM_HIP_ERR_ARG;
    hipSetDevice(c->device);
    // what may still read the group's memory: the kernels of batches that were enqueued and not waited for (not the whole compute stream: it may
    // already carry batches of the NEXT group -- the job level keeps its three batches in flight across a group's end -- and not the copy stream:
    // copies read the slots' output buffers), and the group's own walk (not the whole walk stream: it may carry the upload and the walk of the next group)
    const int gid = (int)(g - c->groups.data());
    // a batch of the group that was enqueued and not yet waited for: its wait may have to run it AGAIN (an Ion Torrent read that outgrew its buffers,
    // dwgsim_hip_wait), which needs the group -- wait first, then drop (include/dwgsim_hip.h)
    for (int s = 0; s < DWGSIM_HIP_SLOTS; ++s) if (c->slot[s].pending && !c->slot[s].empty && c->slot[s].group == gid) { c->err = "drop_contig: a batch that reads this group has not been waited for (dwgsim_hip_wait first)"; return DWGSIM_HIP_ERR_STATE; }
    if (g->walk_pending) hipEventSynchronize(g->ev_walk);
    for (size_t k = 0; k < g->m.size(); ++k) { if (c->chain_contig == g->first_handle + (int)k) c->chain_contig = -1; c->handles[(size_t)g->first_handle + k].group = -1; }
    // the group's memory waits for the next group (at most three sets are kept: the one that has waited longest goes -- a set that no later
    // group can use, e.g. one more than four times too large, does not stay for the life of the context; round 5 dropped the smallest)
    g->alive = false;
    c->pool.push_back(*g);
    *g = Group();
    if (c->pool.size() > 3) {
        hipStreamSynchronize(c->walk_stream);
        free_group(c->pool[0]); c->pool.erase(c->pool.begin());
    }
    return DWGSIM_HIP_OK;
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

int64_t dwgsim_hip_contig_region_length(dwgsim_hip_ctx_t *c, uint32_t contig_index, const uint8_t *ascii, int64_t len, int64_t *non_acgt, int64_t *region_bases)
{
    if (non_acgt) *non_acgt = 0;
    if (region_bases) *region_bases = len;
    if (!c || !c->has_regions) return len;
    int64_t m = 0, num_n = 0;
    for (size_t q = 0; q < c->regions.contig.size(); ++q) if (c->regions.contig[q] == contig_index) {
        m += c->regions.end[q] - c->regions.start[q];
        // the reference walks start..end INCLUSIVE with 1-based indexing (dwgsim.c:558-559); its read of
        // s[-1] for start == 0 is out of bounds there and is counted as non-ACGT here
        for (int64_t p = c->regions.start[q]; p <= (int64_t)c->regions.end[q]; ++p) { const int ch = (p >= 1 && p - 1 < len) ? ascii[p - 1] : 'N'; if (nt4(ch) >= 4) ++num_n; }
    }
    if (non_acgt) *non_acgt = num_n;
    if (region_bases) *region_bases = m;
    if (m == 0) return DWGSIM_HIP_SKIP_NO_REGION;
    if (0.95 < num_n / (double)m) return DWGSIM_HIP_SKIP_NON_ACGT;
    return m;
}

int dwgsim_hip_contig_set_placement_length(dwgsim_hip_ctx_t *c, int contig, int64_t l)
{
    int k = 0;
    Group *g = get_group(c, contig, &k);
    if (!g || l < 0) { if (c) c->err = "bad placement-length arguments"; return DWGSIM_HIP_ERR_ARG; }
    g->m[(size_t)k].l_place = l;
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

// mut_debug (mut.c:379-425): where the reference's asserts end the run (SIGABRT) the call returns DWGSIM_HIP_ERR_FAILED with the assert's text.
// The reference checks contig after contig, each before (mut.c:753) and after (:757) its justification: of the two verdicts of the group
// (smallest failing position of each pass) the one in the earlier contig -- the pre-justification one on a tie -- is the one it would have hit.
static int mut_debug_verdict(dwgsim_hip_ctx_t *c, const Group &g, uint64_t pre, uint64_t post)
{
    auto member_of = [&](uint64_t v) -> int { const int64_t p = (int64_t)(v >> 8); int k = 0; while (k + 1 < (int)g.m.size() && (int64_t)g.m[(size_t)k + 1].start <= p) ++k; return k; };
    uint64_t v = ~0ull;
    if (pre != ~0ull && post != ~0ull) v = member_of(post) < member_of(pre) ? post : pre;
    else v = pre != ~0ull ? pre : post;
    if (v == ~0ull) return DWGSIM_HIP_OK;
    static const char *what[4] = {"", "(c[0]&0x3) != (c[1]&0x3)", "(c[1]&0x3) != (c[2]&0x3)", "(c[0]&0x3) == (c[1]&0x3) || (c[0]&0x3) == (c[2]&0x3)"};
    const Member &m = g.m[(size_t)member_of(v)];
    char b[256]; snprintf(b, sizeof b, "dwgsim: src/mut.c: mut_debug: Assertion `%s' failed. [%s:%lld]\n", what[v & 3], m.name.c_str(), (long long)(v >> 8) - m.start + 1);
    c->err = b;
    return DWGSIM_HIP_ERR_FAILED;
}

// One attempt of the walk of a whole group, enqueued on the walk stream without any host read-back in between: buffers are sized for a
// capacity (candidate sites are a Binomial(l, mut_rate) draw: mean + 8 sigma), the kernels take their element counts from device memory, and
// the read-back at the end (dwgsim_hip_mutate_wait) also tells whether a capacity was exceeded -- then the walk is simply run again with
// exact sizes.
static int enqueue_walk(dwgsim_hip_ctx_t *c, Group &g)
{
    const WalkParams wp = walk_params(c);
    const int64_t total = g.total;
    const size_t padded = padded_cells(g);
    hipStream_t st = c->walk_stream;
    const SegTab seg = seg_tab(g);
    HIPC(c, hipEventRecord(g.ev_walk0, st));
    if (c->has_mutin) {      // file-driven mutations (mut.c:644-745): the host resolved the entries, the GPU scatters and left-justifies
        const uint32_t np = g.n_patch, nev = g.n_patch_ev;
        if (g.walk_reset) for (int h = 0; h < 2; ++h) HIPC(c, hipMemcpyAsync(g.d_cells[h], g.d_ref, padded, hipMemcpyDeviceToDevice, st));
        HIPC(c, hipMemsetAsync(&c->d_wcounters[12], 0xff, 2 * sizeof(uint64_t), st));
        if (np) {
            if (ensure(c, c->w_ppos, sizeof(int32_t) * np) || ensure(c, c->w_pcells, sizeof(uint16_t) * np) || ensure(c, c->w_ev, sizeof(Event) * (nev ? nev : 1)) ||
                ensure(c, c->w_lo, sizeof(int32_t) * (nev ? nev : 1)) || ensure(c, c->w_sufmin, sizeof(int32_t) * ((nev ? nev : 1) + 64)) || ensure(c, c->w_bound, nev ? nev : 1)) return DWGSIM_HIP_ERR_DEVICE;      // (+ 64: segment minima of k_sufmin)
            HIPC(c, hipMemcpyAsync(c->w_ppos.p, c->h_ppos.data(), sizeof(int32_t) * np, hipMemcpyHostToDevice, st));
            HIPC(c, hipMemcpyAsync(c->w_pcells.p, c->h_pcells.data(), sizeof(uint16_t) * np, hipMemcpyHostToDevice, st));
            if (nev) HIPC(c, hipMemcpyAsync(c->w_ev.p, c->h_pev.data(), sizeof(Event) * nev, hipMemcpyHostToDevice, st));
            launch_apply_patches(st, (const int32_t *)c->w_ppos.p, (const uint16_t *)c->w_pcells.p, np, g.d_cells[0], g.d_cells[1]);
            const ContigDev cd = group_dev(g);
            launch_mut_debug(st, g.d_ref, g.d_cells[0], g.d_cells[1], total, &c->d_wcounters[12]);      // mut.c:753
            if (nev) {
                if (c->seq_justify) launch_justify_seq(st, (const Event *)c->w_ev.p, Count{nullptr, nev}, cd);
                else launch_justify(st, (const Event *)c->w_ev.p, Count{nullptr, nev}, cd, (int32_t *)c->w_lo.p, (int32_t *)c->w_sufmin.p, (uint8_t *)c->w_bound.p);
            }
            launch_mut_debug(st, g.d_ref, g.d_cells[0], g.d_cells[1], total, &c->d_wcounters[13]);      // mut.c:757
        }
        launch_make_view(st, g.d_cells[0], g.d_cells[1], (int64_t)padded & ~(int64_t)15, g.total, g.d_view[0], g.d_view[1], g.d_summ[0], g.d_summ[1], g.d_summ2[0], g.d_summ2[1]);
        g.dirty_all = true;
        HIPC(c, hipGetLastError());
        HIPC(c, hipMemcpyAsync(&g.h_wc[12], &c->d_wcounters[12], 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPC(c, hipEventRecord(g.ev_walk, st));
        return DWGSIM_HIP_OK;
    }
    const uint32_t nblk = (uint32_t)((total + SCAN_POS_PER_BLOCK - 1) / SCAN_POS_PER_BLOCK);
    if (ensure(c, c->scratch_status, ((size_t)nblk + 2) * sizeof(uint64_t))) return DWGSIM_HIP_ERR_DEVICE;      // look-back words of k_site_scan_list + its ticket
    uint64_t *d_status = (uint64_t *)c->scratch_status.p, *d_ticket = d_status + nblk;
    // where the group stands: fresh from the upload, or with the chunks the previous walk wrote -- those go back to the pristine copies --, or (a
    // capacity re-run, a walk that kept no bitmap) all of it
    if (g.walk_attempt > 0 || g.dirty_all) {
        for (int h = 0; h < 2; ++h) {
            HIPC(c, hipMemcpyAsync(g.d_cells[h], g.d_ref, padded, hipMemcpyDeviceToDevice, st));
            HIPC(c, hipMemcpyAsync(g.d_view[h], g.d_refview, padded / 2, hipMemcpyDeviceToDevice, st));
            HIPC(c, hipMemcpyAsync(g.d_summ[h], g.d_refsumm, sizeof(uint16_t) * (padded / SUMM_CELLS), hipMemcpyDeviceToDevice, st));
            HIPC(c, hipMemcpyAsync(g.d_summ2[h], g.d_refsumm2, sizeof(uint16_t) * (padded / SUMM2_CELLS), hipMemcpyDeviceToDevice, st));
        }
        HIPC(c, hipMemsetAsync(g.d_dirty, 0, sizeof(uint32_t) * ((size_t)g.n_dirty_words + 2), st));
        g.dirty_all = false; g.dirty_any = false;
    } else if (g.dirty_any) {
        launch_dirty_chunks(st, true, g.d_dirty, g.n_dirty_words, g.total, g.d_ref, g.d_refview, g.d_refsumm, g.d_refsumm2, g.d_cells[0], g.d_cells[1], g.d_view[0], g.d_view[1], g.d_summ[0], g.d_summ[1], g.d_summ2[0], g.d_summ2[1]);
        HIPC(c, hipMemsetAsync(g.d_dirty, 0, sizeof(uint32_t) * ((size_t)g.n_dirty_words + 2), st));
        g.dirty_any = false;
    }
    const uint32_t cap = g.walk_cap; const size_t cap_bases = g.walk_cap_bases;
    const size_t ncap = cap ? cap : 1;
    if (ensure(c, c->w_cand, sizeof(int32_t) * ncap) || ensure(c, c->w_ev, sizeof(Event) * ncap) ||
        ensure(c, c->w_flags, sizeof(uint4) * (ncap + 64)) ||      // (+ 64 rows / entries: the segment totals of k_scan4, the segment minima of k_sufmin)
        ensure(c, c->w_lo, sizeof(int32_t) * ncap) || ensure(c, c->w_sufmin, sizeof(int32_t) * (ncap + 64)) ||
        ensure(c, c->w_bound, ncap)) return DWGSIM_HIP_ERR_DEVICE;
    for (int h = 0; h < 2; ++h) {      // insertion tables: at most one entry per candidate; the base pools are checked on the device
        if (ncap > g.cap_ins[h]) {
            hipFree(g.d_ins_pos[h]); hipFree(g.d_ins_len[h]); hipFree(g.d_ins_off[h]);
            g.d_ins_pos[h] = nullptr; g.d_ins_len[h] = g.d_ins_off[h] = nullptr;
            g.cap_ins[h] = ncap + ncap / 4 + 64;
            HIPC(c, hipMalloc((void **)&g.d_ins_pos[h], sizeof(int32_t) * g.cap_ins[h]));
            HIPC(c, hipMalloc((void **)&g.d_ins_len[h], sizeof(uint32_t) * g.cap_ins[h]));
            HIPC(c, hipMalloc((void **)&g.d_ins_off[h], sizeof(uint32_t) * g.cap_ins[h]));
        }
        if (cap_bases > g.cap_bases[h]) {
            hipFree(g.d_ins_bases[h]); g.d_ins_bases[h] = nullptr;
            g.cap_bases[h] = cap_bases + cap_bases / 4 + 256;
            HIPC(c, hipMalloc((void **)&g.d_ins_bases[h], g.cap_bases[h] + 16));
        }
    }
    int32_t *d_cand = (int32_t *)c->w_cand.p; Event *d_ev = (Event *)c->w_ev.p; uint4 *d_flags = (uint4 *)c->w_flags.p;
    uint32_t *d_small = reinterpret_cast<uint32_t *>(&c->d_wcounters[8]);   // [0] max_del, [1..4] tot4: eight words in counters[8..11], so that one copy brings counters[7..11] back
    const Count nc{&c->d_wcounters[7], cap};
    HIPC(c, hipMemsetAsync(&c->d_wcounters[7], 0, 5 * sizeof(uint64_t), st));
    // K1: candidate sites -> ordered list, from the pristine 4-bit view.  First attempt: every block into a slot of its own, no block waits for another
    // (dw_walk.hip k_site_scan_slots: mean + 8 sigma + 32 entries per block of 65 536 positions); a re-run, a mutation rate at which the slots would
    // be as large as the list itself, or "site_slots" = 0: one kernel with a decoupled look-back
    {
        const uint32_t nbs = site_scan_blocks(total);
        const double m = (double)site_scan_block_positions() * (c->prm.mut_rate > 0 ? c->prm.mut_rate : 0.0);
        uint32_t slot_cap = (uint32_t)std::min<double>((double)site_scan_block_positions(), m + 8.0 * sqrt(m + 1.0) + 32.0);
        if (c->site_slot_cap >= 0) slot_cap = (uint32_t)std::max<int64_t>(1, c->site_slot_cap);
        const size_t slot_bytes = (size_t)nbs * slot_cap * sizeof(int32_t);
        const bool use_slots = g.walk_attempt == 0 && c->site_slots != 0 && nbs > 0 && (c->site_slots > 0 || slot_bytes <= std::max<size_t>((size_t)64 << 20, (size_t)ncap * 16));
        if (use_slots) {
            if (ensure(c, c->w_slots, slot_bytes) || ensure(c, c->w_slot_aux, 2 * (size_t)nbs * sizeof(uint32_t))) return DWGSIM_HIP_ERR_DEVICE;
            launch_site_scan_slots(st, g.d_refview, total, seg, wp, (int32_t *)c->w_slots.p, slot_cap, (uint32_t *)c->w_slot_aux.p, d_cand, cap, &c->d_wcounters[7], &d_small[6]);
        } else {
            HIPC(c, hipMemsetAsync(d_status, 0, ((size_t)nblk + 2) * sizeof(uint64_t), st));
            launch_site_scan_list(st, g.d_refview, total, seg, wp, d_status, d_ticket, d_cand, cap, &c->d_wcounters[7]);
        }
    }
    // K2: events, liveness, insertion-table allocation
    launch_events(st, d_cand, nc, g.d_ref, seg, wp, d_ev, &d_small[0]);
    launch_resolve(st, d_ev, nc, &d_small[0], d_flags, &d_small[1]);
    // K3 + K4
    ContigDev cd = group_dev(g);
    cd.tot4 = &d_small[1]; cd.cap_bases[0] = (uint32_t)std::min<size_t>(g.cap_bases[0], 0xFFFFFFFFu); cd.cap_bases[1] = (uint32_t)std::min<size_t>(g.cap_bases[1], 0xFFFFFFFFu);
    launch_apply(st, d_ev, nc, d_flags, cd, wp);
    // mut_debug (mut.c:753, :757) cannot fire on randomly drawn mutations and is not run here: a substitution always changes the base
    // ((c + 1..3) & 3, mut.c:621), a homozygous one writes the same cell to both haplotypes and a heterozygous one leaves the other
    // haplotype's cell as it was -- the reference base, also under a deletion or an insertion, before and after left-justification
    // (which only moves an indel over bases equal to its own).  File-driven mutations (-m / -b / -v, above) can violate all three.
    if (c->seq_justify || c->dense_view) {      // (the cross-check forms: one thread justifies the whole group / the views are made from every cell as rounds 1-4 did)
        if (c->seq_justify) launch_justify_seq(st, d_ev, nc, cd); else launch_justify(st, d_ev, nc, cd, (int32_t *)c->w_lo.p, (int32_t *)c->w_sufmin.p, (uint8_t *)c->w_bound.p);
        launch_make_view(st, g.d_cells[0], g.d_cells[1], (int64_t)padded & ~(int64_t)15, g.total, g.d_view[0], g.d_view[1], g.d_summ[0], g.d_summ[1], g.d_summ2[0], g.d_summ2[1]);
        g.dirty_all = true;
    } else {
        launch_justify(st, d_ev, nc, cd, (int32_t *)c->w_lo.p, (int32_t *)c->w_sufmin.p, (uint8_t *)c->w_bound.p);
        // the chunks the walk may have written (every live event from the lower end of its justification scan to its last cell): their views and summaries
        launch_mark_dirty(st, d_ev, nc, (const int32_t *)c->w_lo.p, g.d_dirty);
        launch_dirty_chunks(st, false, g.d_dirty, g.n_dirty_words, g.total, g.d_ref, g.d_refview, g.d_refsumm, g.d_refsumm2, g.d_cells[0], g.d_cells[1], g.d_view[0], g.d_view[1], g.d_summ[0], g.d_summ[1], g.d_summ2[0], g.d_summ2[1]);
        g.dirty_any = true;
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(&g.h_wc[7], &c->d_wcounters[7], 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));      // [7] candidates, [8..11] the eight words
    HIPC(c, hipEventRecord(g.ev_walk, st));
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_mutate_async(dwgsim_hip_ctx_t *c, int contig)
{
    Group *gp = get_group(c, contig);
    if (!gp) return DWGSIM_HIP_ERR_ARG;
    Group &g = *gp;
    if (g.walk_pending) { c->err = "mutate: the group's previous walk was not waited for"; return DWGSIM_HIP_ERR_STATE; }
    // (walks of several groups may be in flight: they run one after the other on the walk stream and share its device scratch in stream order;
    // what the host reads back afterwards -- counts, mut_debug verdicts -- lands in the group's own page-locked mirror g.h_wc)
    for (const Slot &sl : c->slot) if (sl.pending && sl.group == c->handles[(size_t)contig].group) { c->err = "mutate: a batch that reads this group is still in flight (wait for it first)"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    g.walk_reset = g.mutated;      // walked before: the cells start again from the resident packed reference
    for (int h = 0; h < 2; ++h) g.n_ins[h] = g.n_ins_bases[h] = 0;
    g.mutated = true; g.n_cand = 0; g.list_valid = false;
    g.walk_attempt = 0;
    if (g.total == 0) return DWGSIM_HIP_OK;
    if (c->has_mutin) {      // patches, indel events and insertion tables of the whole group, in group coordinates
        HIPC(c, hipStreamSynchronize(c->walk_stream));      // (an earlier group's walk may still be copying from the host lists below)
        c->h_ppos.clear(); c->h_pcells.clear(); c->h_pev.clear();
        HostIns hi[2];
        for (size_t k = 0; k < g.m.size(); ++k) {
            const Member &m = g.m[k];
            for (size_t q = 0; q < m.rc.pos.size(); ++q) {
                c->h_ppos.push_back(m.rc.pos[q] + m.start); c->h_pcells.push_back(m.rc.cells[q]);
                if (m.rc.cells[q] & 0x3030) { Event e; e.pos = m.rc.pos[q] + m.start; e.type = 4; e.hap = 3; e.base = 0; e.live = 1; e.len = 1; e.seg = (uint32_t)k; c->h_pev.push_back(e); }
            }
            for (int h = 0; h < 2; ++h) for (const InsPayload &ip : m.rc.ins[h]) {
                hi[h].pos.push_back(ip.pos + m.start); hi[h].len.push_back((uint32_t)ip.bases.size()); hi[h].off.push_back((uint32_t)hi[h].bases.size());
                hi[h].bases.insert(hi[h].bases.end(), ip.bases.begin(), ip.bases.end());
            }
        }
        g.n_patch = (uint32_t)c->h_ppos.size(); g.n_patch_ev = (uint32_t)c->h_pev.size();
        g.n_cand = g.n_patch_ev;
        HIPC(c, hipStreamSynchronize(c->walk_stream));      // (the tables below are copied from short-lived host vectors)
        for (int h = 0; h < 2; ++h) {
            const size_t n = hi[h].pos.size(), nb = hi[h].bases.size();
            g.n_ins[h] = (uint32_t)n; g.n_ins_bases[h] = (uint32_t)nb;
            const size_t nn = n ? n : 1, nbb = nb ? nb : 1;
            if (nn > g.cap_ins[h]) {
                hipFree(g.d_ins_pos[h]); hipFree(g.d_ins_len[h]); hipFree(g.d_ins_off[h]);
                g.d_ins_pos[h] = nullptr; g.d_ins_len[h] = g.d_ins_off[h] = nullptr;
                g.cap_ins[h] = nn + nn / 4 + 64;
                HIPC(c, hipMalloc((void **)&g.d_ins_pos[h], sizeof(int32_t) * g.cap_ins[h]));
                HIPC(c, hipMalloc((void **)&g.d_ins_len[h], sizeof(uint32_t) * g.cap_ins[h]));
                HIPC(c, hipMalloc((void **)&g.d_ins_off[h], sizeof(uint32_t) * g.cap_ins[h]));
            }
            if (nbb > g.cap_bases[h]) { hipFree(g.d_ins_bases[h]); g.d_ins_bases[h] = nullptr; g.cap_bases[h] = nbb + nbb / 4 + 256; HIPC(c, hipMalloc((void **)&g.d_ins_bases[h], g.cap_bases[h] + 16)); }
            if (n) {
                HIPC(c, hipMemcpy(g.d_ins_pos[h], hi[h].pos.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
                HIPC(c, hipMemcpy(g.d_ins_len[h], hi[h].len.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
                HIPC(c, hipMemcpy(g.d_ins_off[h], hi[h].off.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
                HIPC(c, hipMemcpy(g.d_ins_bases[h], hi[h].bases.data(), nb, hipMemcpyHostToDevice));
            }
        }
    } else {
        double mean = 0;
        for (const Member &m : g.m) mean += (double)m.l * c->prm.mut_rate;
        g.walk_cap = (uint32_t)std::min<double>((double)g.total, mean + 8.0 * sqrt(mean + 1.0) + 256.0);
        g.walk_cap_bases = (size_t)g.walk_cap * 8 + 4096;
        if (c->walk_cap >= 0) { g.walk_cap = (uint32_t)c->walk_cap; g.walk_cap_bases = 1; }      // dwgsim_hip_debug_option("walk_cap"): start too small, exercise the re-run
    }
    if (const int rc = enqueue_walk(c, g)) return rc;
    g.walk_pending = true;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_mutate_wait(dwgsim_hip_ctx_t *c, int contig)
{
    Group *gp = get_group(c, contig);
    if (!gp) return DWGSIM_HIP_ERR_ARG;
    Group &g = *gp;
    if (!g.walk_pending) return g.mutated ? DWGSIM_HIP_OK : (c->err = "mutate_wait: no walk was enqueued for this group", DWGSIM_HIP_ERR_STATE);
    HIPC(c, hipSetDevice(c->device));
    for (;;) {
        HIPC(c, hipEventSynchronize(g.ev_walk));
        { float ms = 0; if (hipEventElapsedTime(&ms, g.ev_walk0, g.ev_walk) == hipSuccess) c->walk_us += 1e3 * ms; else (void)hipGetLastError(); }      // (analysis: dwgsim_hip_debug_get "walk_us")
        if (c->has_mutin) {
            g.walk_pending = false;
            return g.n_patch ? mut_debug_verdict(c, g, g.h_wc[12], g.h_wc[13]) : DWGSIM_HIP_OK;
        }
        const uint64_t n_cand = g.h_wc[7];
        const uint32_t *h_small = reinterpret_cast<const uint32_t *>(&g.h_wc[8]);
        const bool slot_over = h_small[6] != 0;      // a block of the site scan outgrew its slot (nothing behind it ran: the candidate count was set to 0; h_small[7] = the real one)
        const bool fits = !slot_over && n_cand <= g.walk_cap && h_small[2] <= g.cap_bases[0] && h_small[4] <= g.cap_bases[1];
        if (fits || g.walk_attempt >= 2) {
            g.walk_pending = false;
            if (!fits) { c->err = "mutation walk: capacities still exceeded after an exact re-run"; return DWGSIM_HIP_ERR_FAILED; }
            g.n_cand = (uint32_t)n_cand;
            for (int h = 0; h < 2; ++h) { g.n_ins[h] = h_small[1 + 2 * h]; g.n_ins_bases[h] = h_small[2 + 2 * h]; }
            return DWGSIM_HIP_OK;
        }
        // exact sizes (the counts read back are those of the complete candidate list unless it was truncated: take generous ones then)
        g.walk_cap = (uint32_t)std::min<uint64_t>((uint64_t)g.total, (slot_over ? (uint64_t)h_small[7] : n_cand) + 16);
        g.walk_cap_bases = std::max<size_t>(g.walk_cap_bases, (size_t)std::max(h_small[2], h_small[4]) * 2 + (size_t)g.walk_cap * 8 + 4096);
        ++g.walk_attempt;
        if (const int rc = enqueue_walk(c, g)) { g.walk_pending = false; return rc; }
    }
}

int dwgsim_hip_mutate_poll(dwgsim_hip_ctx_t *c, int contig)
{
    Group *gp = get_group(c, contig);
    if (!gp) return DWGSIM_HIP_ERR_ARG;
    if (!gp->walk_pending) return 1;
    HIPC(c, hipSetDevice(c->device));
    const hipError_t e = hipEventQuery(gp->ev_walk);
    if (e == hipSuccess) return 1;
    (void)hipGetLastError();
    return e == hipErrorNotReady ? 0 : (c->err = "mutate_poll: device error", DWGSIM_HIP_ERR_DEVICE);
}

int dwgsim_hip_mutate_contig(dwgsim_hip_ctx_t *c, int contig)
{
    const int rc = dwgsim_hip_mutate_async(c, contig);
    if (rc != DWGSIM_HIP_OK) return rc;
    return dwgsim_hip_mutate_wait(c, contig);
}

// ---- mutations.txt / mutations.vcf from the sparse list of mutated cells (mut.c:781-893) ----
namespace {
// The text is made by plain appends (a vsnprintf per field cost 0.2 us per mutation: 50 ms for a 250 Mb chromosome, during which the worker of
// round 4's job level enqueued nothing); it is also a pure function of the list, so that it can be made by another thread than the one that
// drives the device (dwgsim_hip_mutations_take / dwgsim_hip_mutlist_text).
struct TextOut {
    std::string &s;
    void ch(char c) { s.push_back(c); }
    void str(const char *p, size_t n) { s.append(p, n); }
    void lit(const char *p) { s.append(p); }
    void num(long long v)
    {
        char b[24]; int n = 0; unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) s.push_back('-');
        while (n) s.push_back(b[--n]);
    }
};
struct ListView {      // the mutated cells of a group (group coordinates) and its insertion tables
    const int32_t *pos; const uint32_t *cells; size_t n; const HostIns *ins;
};
void ins_text(const HostIns &t, int32_t pos, std::string &tmp)
{
    tmp.clear();
    auto it = std::lower_bound(t.pos.begin(), t.pos.end(), pos);
    if (it == t.pos.end() || *it != pos) return;
    const size_t k = (size_t)(it - t.pos.begin());
    for (uint32_t q = 0; q < t.len[k]; ++q) tmp.push_back("ACGTN"[t.bases[t.off[k] + q] & 3]);
}
// mut_print (mut.c:781-893) for one contig: name, first cell s0 and length l in the group's coordinates
void format_mutations(const ListView &g, const std::string &name, int64_t s0, int64_t l, std::string &txt_s, std::string &vcf_s)
{
    txt_s.clear(); vcf_s.clear();
    // this contig's slice of the list, positions inside the contig
    const size_t e0 = (size_t)(std::lower_bound(g.pos, g.pos + g.n, (int32_t)s0) - g.pos);
    const size_t e1 = (size_t)(std::lower_bound(g.pos, g.pos + g.n, (int32_t)(s0 + l)) - g.pos);
    txt_s.reserve((e1 - e0) * (name.size() + 20)); vcf_s.reserve((e1 - e0) * (name.size() + 56));
    // sparse restatement of the per-position loop: only listed positions can print; "previous position
    // mutated" (mut_prev, mut.c:890-891) is "position i-1 is listed with a mutated cell on that haplotype"
    static const char B5[] = "ACGTN";       // B5[5] is the terminating NUL, as in the reference for code 5 ('-')
    TextOut txt{txt_s}, vcf{vcf_s};
    // (a %c of the NUL of B5[5] prints a NUL byte in the reference; r0 < 4 and substituted / inserted bases are 0-3 here, deleted reference bases too)
    std::string tmp;
    auto cell = [&](size_t e, int h) -> uint8_t { return (uint8_t)(h ? g.cells[e] >> 8 : g.cells[e]); };
    auto refc = [&](size_t e) -> uint8_t { return (uint8_t)(g.cells[e] >> 16); };          // nst_nt4_table code of the reference base
    auto prevc = [&](size_t e) -> uint8_t { return (uint8_t)(g.cells[e] >> 24); };         // ... of the base in front of it
    auto vcf_head = [&](long long at) { vcf.str(name.data(), name.size()); vcf.ch('\t'); vcf.num(at); vcf.lit("\t.\t"); };
    for (size_t e = e0; e < e1; ++e) {
        const int64_t i = g.pos[e] - s0;
        const uint8_t r0 = refc(e), c1 = cell(e, 0), c2 = cell(e, 1);
        if (r0 >= 4) continue;
        const bool adj = e > e0 && g.pos[e - 1] == g.pos[e] - 1;
        const bool prev0 = adj && (cell(e - 1, 0) & TMASK) != T_NONE, prev1 = adj && (cell(e - 1, 1) & TMASK) != T_NONE;
        txt.str(name.data(), name.size()); txt.ch('\t'); txt.num((long long)i + 1); txt.ch('\t');
        const bool hom = (c1 & BTMASK) == (c2 & BTMASK);
        const uint8_t t1 = c1 & TMASK, t2 = c2 & TMASK;
        if (hom ? t1 == T_SUB : (t1 == T_SUB || t2 == T_SUB)) {
            if (hom) {
                txt.ch(B5[r0]); txt.ch('\t'); txt.ch(B5[c1 & 0xf]); txt.lit("\t3\n");
                vcf_head((long long)i + 1); vcf.ch(B5[r0]); vcf.ch('\t'); vcf.ch(B5[c1 & 0xf]); vcf.lit("\t100\tPASS\tAF=1.0;pl=3;mt=SUBSTITUTE\n");
            } else {
                const int hap = t1 == T_SUB ? 1 : 2;
                txt.ch(B5[r0]); txt.ch('\t'); txt.ch("XACMGRSVTWYHKDBN"[1 << (c1 & 3) | 1 << (c2 & 3)]); txt.ch('\t'); txt.ch((char)('0' + hap)); txt.ch('\n');
                vcf_head((long long)i + 1); vcf.ch(B5[r0]); vcf.ch('\t'); vcf.ch(B5[(hap == 1 ? c1 : c2) & 0xf]); vcf.lit("\t100\tPASS\tAF=0.5;pl="); vcf.ch((char)('0' + hap)); vcf.lit(";mt=SUBSTITUTE\n");
            }
        } else if (hom ? t1 == T_DEL : (t1 == T_DEL || t2 == T_DEL)) {
            const int pl = hom ? 3 : (t1 == T_DEL ? 1 : 2);
            txt.ch(B5[r0]); txt.lit("\t-\t"); txt.ch((char)('0' + pl)); txt.ch('\n');
            const bool open = hom ? (!prev0 || !prev1) : !(pl == 1 ? prev0 : prev1);
            if (open) {      // one VCF record for the run, anchored at the previous reference base (mut.c:801-815)
                vcf_head((long long)i);
                if (i > 0) vcf.ch(B5[prevc(e)]);
                size_t ee = e; int64_t j = i;
                for (;;) {
                    vcf.ch(B5[refc(ee)]);
                    if (j + 1 >= l) break;
                    // cell at j+1: listed -> its cells, else unmutated
                    if (ee + 1 < e1 && g.pos[ee + 1] - s0 == j + 1) {
                        ++ee; ++j;
                        const uint8_t a1 = cell(ee, 0), a2 = cell(ee, 1);
                        const bool h2 = (a1 & BTMASK) == (a2 & BTMASK);
                        if (!(h2 == hom && ((pl == 2 ? a2 : a1) & TMASK) == T_DEL)) break;
                    } else break;
                }
                if (i > 0) { vcf.ch('\t'); vcf.ch(B5[prevc(e)]); } else vcf.lit("\t.");
                vcf.lit("\t100\tPASS\tAF="); vcf.lit(hom ? "1.0" : "0.5"); vcf.lit(";pl="); vcf.ch((char)('0' + pl)); vcf.lit(";mt=DELETE\n");
            }
        } else {
            const int pl = hom ? 3 : (t1 == T_INS ? 1 : 2);
            ins_text(g.ins[pl == 2 ? 1 : 0], g.pos[e], tmp);
            txt.lit("-\t"); txt.str(tmp.data(), tmp.size()); txt.ch('\t'); txt.ch((char)('0' + pl)); txt.ch('\n');
            vcf_head((long long)i + 1); vcf.ch(B5[r0]); vcf.ch('\t'); vcf.ch(B5[r0]); vcf.str(tmp.data(), tmp.size());
            vcf.lit("\t100\tPASS\tAF="); vcf.lit(hom ? "1.0" : "0.5"); vcf.lit(";pl="); vcf.ch((char)('0' + pl)); vcf.lit(";mt=INSERT\n");
        }
    }
}

// the mutated cells of a walked group (positions in group coordinates, cells + reference codes) and its insertion tables, fetched once
int fetch_mutated_list(dwgsim_hip_ctx_t *c, Group &g)
{
    if (g.list_valid) return DWGSIM_HIP_OK;
    g.pos.clear(); g.cells.clear();
    for (int h = 0; h < 2; ++h) g.ins[h] = HostIns();
    hipStream_t st = c->walk_stream;
    if (g.total > 0 && g.n_cand > 0) {
        const uint32_t nblk = (uint32_t)((g.total + SCAN_POS_PER_BLOCK - 1) / SCAN_POS_PER_BLOCK);
        if (ensure(c, c->scratch_mask, (size_t)nblk * SCAN_THREADS * sizeof(uint16_t))) return DWGSIM_HIP_ERR_DEVICE;
        if (ensure(c, c->scratch_cnt, (size_t)nblk * sizeof(uint32_t))) return DWGSIM_HIP_ERR_DEVICE;
        uint16_t *d_mask = (uint16_t *)c->scratch_mask.p; uint32_t *d_cnt = (uint32_t *)c->scratch_cnt.p;
        launch_collect_mask(st, g.d_cells[0], g.d_cells[1], g.total, d_mask, d_cnt);
        launch_scan_excl(st, d_cnt, nblk, &c->d_wcounters[14]);
        HIPC(c, hipMemcpyAsync(&g.h_wc[14], &c->d_wcounters[14], sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPC(c, hipStreamSynchronize(st));
        const uint32_t n = (uint32_t)g.h_wc[14];
        if (n) {
            if (ensure(c, c->l_pos, sizeof(int32_t) * (size_t)n) || ensure(c, c->l_cells, sizeof(uint32_t) * (size_t)n)) return DWGSIM_HIP_ERR_DEVICE;
            launch_compact(st, d_mask, d_cnt, (int32_t *)c->l_pos.p, g.total, n);
            launch_gather(st, (const int32_t *)c->l_pos.p, n, g.d_ref, g.d_cells[0], g.d_cells[1], (uint32_t *)c->l_cells.p);
            g.pos.resize(n); g.cells.resize(n);
            HIPC(c, hipMemcpyAsync(g.pos.data(), c->l_pos.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
            HIPC(c, hipMemcpyAsync(g.cells.data(), c->l_cells.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
        }
        for (int h = 0; h < 2; ++h) if (g.n_ins[h]) {
            g.ins[h].pos.resize(g.n_ins[h]); g.ins[h].len.resize(g.n_ins[h]); g.ins[h].off.resize(g.n_ins[h]); g.ins[h].bases.resize(g.n_ins_bases[h]);
            HIPC(c, hipMemcpyAsync(g.ins[h].pos.data(), g.d_ins_pos[h], sizeof(int32_t) * g.n_ins[h], hipMemcpyDeviceToHost, st));
            HIPC(c, hipMemcpyAsync(g.ins[h].len.data(), g.d_ins_len[h], sizeof(uint32_t) * g.n_ins[h], hipMemcpyDeviceToHost, st));
            HIPC(c, hipMemcpyAsync(g.ins[h].off.data(), g.d_ins_off[h], sizeof(uint32_t) * g.n_ins[h], hipMemcpyDeviceToHost, st));
            HIPC(c, hipMemcpyAsync(g.ins[h].bases.data(), g.d_ins_bases[h], g.n_ins_bases[h], hipMemcpyDeviceToHost, st));
        }
        HIPC(c, hipStreamSynchronize(st));
    }
    g.list_valid = true;
    return DWGSIM_HIP_OK;
}
} // namespace

int dwgsim_hip_mutations_text(dwgsim_hip_ctx_t *c, int contig, const char **txt, size_t *txt_len, const char **vcf, size_t *vcf_len)
{
    int km = 0;
    Group *gp = get_group(c, contig, &km);
    if (!gp) return DWGSIM_HIP_ERR_ARG;
    Group &g = *gp;
    if (!g.mutated || g.walk_pending) { c->err = "mutate_contig must run first"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    if (const int rc = fetch_mutated_list(c, g)) return rc;
    const Member &m = g.m[(size_t)km];
    format_mutations(ListView{g.pos.data(), g.cells.data(), g.pos.size(), g.ins}, m.name, m.start, m.l, c->txt, c->vcf);
    if (txt) *txt = c->txt.data();
    if (txt_len) *txt_len = c->txt.size();
    if (vcf) *vcf = c->vcf.data();
    if (vcf_len) *vcf_len = c->vcf.size();
    return DWGSIM_HIP_OK;
}

// The same in two halves, for callers that keep the device busy meanwhile (the job level): `take` fetches the group's list of mutated cells (device
// work, on the calling thread) and hands it over as an object of its own; `mutlist_text` makes the text of one of the group's contigs from it --
// no context, no device: any thread may call it while the context goes on with the group's reads.
struct dwgsim_hip_mutlist {
    std::vector<int32_t> pos; std::vector<uint32_t> cells; HostIns ins[2];
    struct M { std::string name; int64_t l; int32_t start; };
    std::vector<M> m;
    std::string txt, vcf;
};

dwgsim_hip_mutlist_t *dwgsim_hip_mutations_take(dwgsim_hip_ctx_t *c, int contig, int *n_contigs)
{
    if (n_contigs) *n_contigs = 0;
    if (!c) return nullptr;
    Group *gp = get_group(c, contig);
    if (!gp) return nullptr;
    Group &g = *gp;
    if (!g.mutated || g.walk_pending) { c->err = "mutate_contig must run first"; return nullptr; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "mutations_take: device error"; return nullptr; }
    if (fetch_mutated_list(c, g)) return nullptr;
    auto *L = new dwgsim_hip_mutlist();
    L->pos.swap(g.pos); L->cells.swap(g.cells);
    for (int h = 0; h < 2; ++h) { L->ins[h] = std::move(g.ins[h]); g.ins[h] = HostIns(); }
    g.list_valid = false;      // (a later mutations_text for this group fetches it again)
    for (const Member &m : g.m) L->m.push_back({m.name, m.l, m.start});
    if (n_contigs) *n_contigs = (int)L->m.size();
    return L;
}

int dwgsim_hip_mutlist_text(dwgsim_hip_mutlist_t *L, int k, const char **txt, size_t *txt_len, const char **vcf, size_t *vcf_len)
{
    if (!L || k < 0 || k >= (int)L->m.size()) return DWGSIM_HIP_ERR_ARG;
    const auto &m = L->m[(size_t)k];
    format_mutations(ListView{L->pos.data(), L->cells.data(), L->pos.size(), L->ins}, m.name, m.start, m.l, L->txt, L->vcf);
    if (txt) *txt = L->txt.data();
    if (txt_len) *txt_len = L->txt.size();
    if (vcf) *vcf = L->vcf.data();
    if (vcf_len) *vcf_len = L->vcf.size();
    return DWGSIM_HIP_OK;
}

void dwgsim_hip_mutlist_free(dwgsim_hip_mutlist_t *L) { delete L; }

// ---- read simulation ----
// The ranges of one launch: all of one walked group, in file order.  ppb = pairs per block of the kernel that will run.
static int build_ranges(dwgsim_hip_ctx_t *c, const dwgsim_hip_range_t *r, int n, uint64_t ppb, Group **gout, std::vector<SimSeg> &segs, uint64_t *n_pairs, uint32_t *n_blocks, int *fixed_max)
{
    if (!r || n < 1) { c->err = "bad range arguments"; return DWGSIM_HIP_ERR_ARG; }
    Group *g = nullptr;
    uint64_t pairs = 0, blocks = 0; int fmax = 0;
    segs.clear();
    for (int q = 0; q < n; ++q) {
        int k = 0;
        Group *gq = get_group(c, r[q].contig, &k);
        if (!gq) return DWGSIM_HIP_ERR_ARG;
        if (g && gq != g) { c->err = "the ranges of one call must belong to contigs that were added together (dwgsim_hip_add_contigs)"; return DWGSIM_HIP_ERR_ARG; }
        g = gq;
        if (r[q].n_pairs == 0) continue;
        const Member &m = g->m[(size_t)k];
        if (c->has_regions && m.n_reg == 0 && c->prm.rand_read < 1.0) { c->err = "dwgsim-hip: this contig has no target region (-x): the reference's placement loop would not terminate\n"; return DWGSIM_HIP_ERR_ARG; }
        SimSeg s; memset(&s, 0, sizeof s);
        s.first_block = (uint32_t)blocks; s.contig_index = m.contig_index; s.start = m.start; s.l = (int32_t)m.l;
        s.first_ii = r[q].first_ii; s.n_pairs = r[q].n_pairs; s.pair_off = pairs; s.l_place = m.l_place;
        s.reg_off = m.reg_off; s.n_reg = m.n_reg; s.name_off = m.name_off; s.name_fixed_len = m.name_fixed_len;
        s.contig_start = r[q].first_ii == 0 ? 1u : 0u;
        segs.push_back(s);
        pairs += r[q].n_pairs; blocks += (r[q].n_pairs + ppb - 1) / ppb;
        if (m.name_fixed_len > fmax) fmax = m.name_fixed_len;
        if (blocks > 0x7fffffffull) { c->err = "too many pairs in one call"; return DWGSIM_HIP_ERR_ARG; }
    }
    if (!g->mutated || g->walk_pending) { c->err = "mutate_contig must run first"; return DWGSIM_HIP_ERR_STATE; }
    *gout = g; *n_pairs = pairs; *n_blocks = (uint32_t)blocks; *fixed_max = fmax;
    return DWGSIM_HIP_OK;
}

static int fill_sim_args(dwgsim_hip_ctx_t *c, Group &g, SimArgs &a)
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
    fill_haps(g, a.hap);
    a.chain = c->d_chain;
    a.have_regions = c->has_regions ? 1 : 0; a.reg = g.d_reg;
    for (int j = 0; j < 2; ++j) { a.e_thr[j] = c->d_thr[j]; a.e_thr32[j] = c->d_thr32[j]; a.qbase[j] = c->d_qbase[j] ? c->d_qbase[j] : c->d_qbase[0]; }
    a.qb_words = c->qb_words;
    a.e_full = c->e_full;
    a.names = g.d_names;
    a.summ[0] = g.d_summ[0]; a.summ[1] = g.d_summ[1]; a.summ2[0] = g.d_summ2[0]; a.summ2[1] = g.d_summ2[1];
    // k_place decides most pairs without their insert size: |normal| <= sqrt(-2 ln 2^-104) < 12.01 for the polar method on 53-bit uniforms (dw_simulate.hip pair_surely_accepted)
    a.place_fast = (!c->has_regions && !p.amplicons && p.std_dev * 12.1 + 2.0 < 1e9) ? 1 : 0;
    a.place_k = a.place_fast ? (int32_t)ceil(p.std_dev * 12.1) + 2 : 0;
    a.rand_fixed = c->d_rand_fixed; a.rand_fixed_len = c->rand_fixed_len;
    // lanes per k_simulate block: the staged read (lds_words per lane) must fit LDS; long Illumina / SOLiD reads get one-wave blocks
    const int lmax0 = p.length[0] > p.length[1] ? p.length[0] : p.length[1];
    a.sim_threads = SIM_THREADS;
    // the FIFO writer unless its LDS costs a resident block per CU (Illumina reads between ~150 and ~240 bases): then 16-byte pieces from registers
    a.fifo = 1;
    if (p.data_type == 0) {
        // waves per SIMD the registers allow (k_simulate launch bounds): five, except with both output families (-o 0) through two register writers;
        // through the FIFO both families leave from one image (dw_read.hpp FifoWriter DUAL) and the kernel needs no more registers than for one
        const int cap_fifo = 5, cap_reg = (p.reads_output_type == 0) ? 4 : 5;
        const size_t w0 = (size_t)((lmax0 + 7) / 8);
        const int b_fifo = sim_blocks_per_cu(sim_lds_bytes(w0, SIM_THREADS, (size_t)c->qb_words, true), cap_fifo), b_reg = sim_blocks_per_cu(sim_lds_bytes(w0, SIM_THREADS, (size_t)c->qb_words, false), cap_reg);
        // (-o 0: one assembly for both families outweighs a resident block: 2 x 250 bp 7.72 ms with three blocks per CU against 8.13 ms with four, profiles/r05_o0.txt)
        if (b_fifo < b_reg && !(p.reads_output_type == 0 && b_fifo >= 3)) a.fifo = 0;
        if (c->writer >= 0) a.fifo = c->writer ? 1 : 0;
    }
    auto lds_need = [&](int lanes) { return sim_lds_bytes((size_t)((lmax0 + 7) / 8), (size_t)lanes, (size_t)c->qb_words, a.fifo != 0); };     // staged bases + the two base-quality tables + the text FIFOs
    // (from where LDS staging leaves room for ONE 256-lane block per CU -- reads of ~650 bases on -- the one-wave blocks are the faster form: 800 / 1 000 /
    // 2 000 bases 13.1 / 13.0 / 22.7 ms in LDS against 7.8 / 8.0 / 8.6 ms staged in scratch slots, 600 bases 7.4 against 7.9: profiles/r04_long_reads.txt)
    if (p.data_type != 2 && (lds_need(SIM_THREADS) > SIM_LDS_BUDGET || ((sim_blocks_per_cu(lds_need(SIM_THREADS), 8) < 2 || c->force_threads == SIM_THREADS_LONG) && c->force_threads != SIM_THREADS))) {
        // reads too long to stage in LDS: one-wave blocks whose reads are staged in scratch slots (global memory, dw_simulate.hip GS); what is left in LDS
        // is the two base-quality tables (2 bytes per base) and the FIFOs, which bounds a read at ~70 000 bases (the reference has no bound, dwgsim.c:75-153)
        a.sim_threads = SIM_THREADS_LONG; a.fifo = 1;
        if (sim_lds_bytes(0, SIM_THREADS_LONG, (size_t)c->qb_words, true) > SIM_LDS_BUDGET) {
            char b[160]; snprintf(b, sizeof b, "dwgsim-hip: reads longer than %d bases are not supported for -c 0 / -c 1\n", (int)((SIM_LDS_BUDGET - SIM_THREADS_LONG * SIM_FIFO_BYTES) / 2 - 16));
            c->err = b; return DWGSIM_HIP_ERR_UNSUP;
        }
    }
    const int lmax = p.length[0] > p.length[1] ? p.length[0] : p.length[1];
    a.cap = lmax;
    if (p.data_type == 2) {        // room for flow-space insertions: ~2.4 empty flows per base, each inserting with probability e, plus cascades
        const double emax = p.e_start[0] > p.e_start[1] ? p.e_start[0] : p.e_start[1];
        a.cap = (c->flow_cap_forced > 0 ? std::max(c->flow_cap_forced, lmax) : flow_read_capacity(lmax, emax, c->flow)) * c->flow_cap_mult;      // (the read as extracted must fit: a forced capacity is a test's starting point, never below the read length)
    }
    a.lds_words = (a.cap + 7) / 8;
    a.flow = c->d_flow; a.flow_len = (int32_t)c->flow.size();
    for (int j = 0; j < 2; ++j) {      // Illumina / SOLiD: the gap chain of a read end's error sites runs at the largest threshold of its ramp (dw_simulate.hip; thresholds as dwgsim_hip_create made them)
        const int n = c->prm.length[j];
        uint64_t tmax = 0, tmin = ~0ull;
        for (int i = 0; i < n; ++i) {
            const double ei = c->prm.e_start[j] + c->e_by[j] * i;
            const uint64_t t = !(ei > 0) ? 0 : ei >= 1.0 ? 0x100000000ull : (uint64_t)ceil(ei * 4294967296.0);
            tmax = std::max(tmax, t); tmin = std::min(tmin, t);
        }
        a.err_thr_max[j] = n > 0 ? tmax : 0; a.err_ramp[j] = (n > 0 && tmin != tmax) ? 1 : 0;
        flow_gap_params(a.err_thr_max[j], &a.err_gap_r[j], &a.err_gap_s[j]);
    }
    for (int j = 0; j < 2; ++j) {      // Ion Torrent: the gap draws of the flow model, from the read end's (uniform) threshold as dwgsim_hip_create made it (thr[0])
        const double e0 = c->prm.e_start[j];
        flow_gap_params(!(e0 > 0) ? 0 : e0 >= 1.0 ? 0x100000000ull : (uint64_t)ceil(e0 * 4294967296.0), &a.flow_gap_r[j], &a.flow_gap_s[j]);
    }
    a.flow_scratch = nullptr; a.flow_free = nullptr; a.flow_slots = 0;
    if (p.data_type == 2) {
        // the flow model's one in-place buffer per lane, 2 bits per base (dw_read.hpp flow_errors), and the run stack of its pass 2.  In LDS while at
        // least two 256-lane blocks -- or else four one-wave blocks -- fit a CU; beyond that (very long reads, error rates at which reads grow
        // severalfold) in scratch slots of global memory ("ion_lds": 0 slots, 1 / 2 LDS with 256 / 64 lanes, -1 choose)
        a.lds_words = (a.cap + 15) / 16; a.cap = 16 * a.lds_words;
        a.flow_stack_words = std::min(FLOW_STACK_WORDS * c->flow_cap_mult, FLOW_STACK_WORDS_MAX);
        const size_t per_lane = (size_t)(a.lds_words + a.flow_stack_words);
        const size_t need256 = sim_lds_bytes(per_lane, SIM_THREADS, (size_t)c->qb_words, true), need64 = sim_lds_bytes(per_lane, ION_THREADS_SMALL, (size_t)c->qb_words, true);
        int mode = c->ion_lds;
        if (mode < 0) mode = (need256 <= SIM_LDS_BUDGET && sim_blocks_per_cu(need256, 8) >= 2) ? 1 : (need64 <= SIM_LDS_BUDGET && sim_blocks_per_cu(need64, 32) >= 4) ? 2 : 0;
        if ((mode == 1 && need256 > SIM_LDS_BUDGET) || (mode == 2 && need64 > SIM_LDS_BUDGET)) mode = 0;
        a.ion_lds = mode != 0; a.sim_threads = mode == 2 ? ION_THREADS_SMALL : SIM_THREADS;
    }
    // Short Illumina reads run as two kernels with the offsets computed in between (dw_simulate.hip SPLIT): no look-backs, and the second half --
    // no staged bases in LDS -- writes the text in 64-byte bursts.  What does not scale with the read length (placement, the look-backs, the name)
    // is most of the work there: 2 x 36 / 2 x 50 / 2 x 75 bp and 100 bp single-end run 16 / 17 / 9 / 15 % faster than in the single kernel, 2 x 100
    // the same, 2 x 150 2-3 % slower (the state crosses HBM, 2.5 GB per chr20-sized launch): profiles/r04_split.txt.  "split" = 0 / 1 forces either.
    // Round 6, the single kernel with ONE look-back (dw_simulate.hip ONE_LB), re-measured (profiles/r06_bench_lines_final.txt 14): 2 x 36 bp +5.5 % as two kernels, 2 x 50 +0.8 %, 2 x 75
    // -3 %, 2 x 100 -10 %, 100 bp single-end -9 %, 2 x 150 -12 %: the cut moves from 100 bases to 50.
    const bool split_wins = lmax0 <= 50;
    a.split = (p.data_type == 0 && a.sim_threads == SIM_THREADS && (c->split < 0 ? split_wins : c->split != 0)) ? 1 : 0;
    // Ion Torrent with its buffers in LDS: as two kernels as well (the flow model | qualities + text).  The first half holds no text FIFOs, so a fourth
    // block fits a CU, and no block waits for the record sizes of the blocks in front of it ("split" = 0 forces the single kernel)
    if (p.data_type == 2 && a.ion_lds && a.sim_threads == SIM_THREADS && a.cap < 32768 && c->split != 0) a.split = 1;
    if (a.split && c->writer < 0) a.fifo = 1;      // (its LDS holds no bases: the FIFO writer always fits)
    return 0;
}

int dwgsim_hip_count_random_ranges(dwgsim_hip_ctx_t *c, const dwgsim_hip_range_t *r, int n, uint64_t *n_random, uint64_t *per_range)
{
    if (!c) return DWGSIM_HIP_ERR_ARG;
    if (n_random) *n_random = 0;
    if (per_range) for (int q = 0; q < n; ++q) per_range[q] = 0;
    Group *gp = nullptr; std::vector<SimSeg> segs; uint64_t n_pairs = 0; uint32_t n_blocks = 0; int fixed_max = 0;
    if (const int rc = build_ranges(c, r, n, PLACE_PAIRS, &gp, segs, &n_pairs, &n_blocks, &fixed_max)) return rc;
    if (n_pairs == 0) return DWGSIM_HIP_OK;
    Group &g = *gp;
    HIPC(c, hipSetDevice(c->device));
    hipStream_t st = c->count_stream;     // (the count of one group can run while batches of another -- or of this one -- are being simulated)
    // (the haplotype summaries k_place reads -- per 64 and per 1024 cells -- were written with the read views at the end of the walk: the count follows
    // its group's walk by that walk's event, whatever else has been put on the walk stream since)
    if (g.walk_pending) { if (const int rc = dwgsim_hip_mutate_wait(c, g.first_handle)) return rc; }      // (a walk that exceeded a capacity is run again inside the wait: only then are the summaries final)
    if (g.ev_walk && g.mutated) HIPC(c, hipStreamWaitEvent(st, g.ev_walk, 0));
    SimArgs a;
    if (const int rc = fill_sim_args(c, g, a)) return rc;
    const size_t ns = segs.size();
    if (ensure(c, c->place_rand, sizeof(uint32_t) * ((size_t)n_blocks + 1))) return DWGSIM_HIP_ERR_DEVICE;
    if (ensure(c, c->place_segs, sizeof(SimSeg) * ns)) return DWGSIM_HIP_ERR_DEVICE;
    if (ensure(c, c->place_aux, PLACE_LISTS * 16 * sizeof(uint32_t) + sizeof(uint64_t) * ns)) return DWGSIM_HIP_ERR_DEVICE;
    if (ns > c->h_place_segs_cap || ns > c->h_range_rand_cap) {
        HIPC(c, hipStreamSynchronize(st));
        if (c->h_place_segs) HIPC(c, hipHostFree(c->h_place_segs));
        if (c->h_range_rand) HIPC(c, hipHostFree(c->h_range_rand));
        c->h_place_segs = nullptr; c->h_range_rand = nullptr; c->h_place_segs_cap = c->h_range_rand_cap = 0;
        HIPC(c, hipHostMalloc((void **)&c->h_place_segs, sizeof(SimSeg) * (ns + 64), hipHostMallocDefault));
        HIPC(c, hipHostMalloc((void **)&c->h_range_rand, sizeof(uint64_t) * (ns + 64), hipHostMallocDefault));
        c->h_place_segs_cap = c->h_range_rand_cap = ns + 64;
    }
    memcpy(c->h_place_segs, segs.data(), sizeof(SimSeg) * ns);
    HIPC(c, hipMemcpyAsync(c->place_segs.p, c->h_place_segs, sizeof(SimSeg) * ns, hipMemcpyHostToDevice, st));
    a.segs = (const SimSeg *)c->place_segs.p; a.n_seg = (int32_t)ns; a.n_blocks = n_blocks; a.n_pairs = n_pairs;
    a.block_rand = (uint32_t *)c->place_rand.p; a.counters = c->d_pcounters;
    // the lists of pairs k_place leaves open: room for an eighth of the pairs (the usual share is a per cent); if that does not do -- contigs
    // made of N runs, a read length close to the contig's -- the count is run once more with room for every pair
    const uint64_t waves = (uint64_t)n_blocks * (PLACE_PAIRS / 64);
    const uint32_t cap_full = (uint32_t)((waves + PLACE_LISTS - 1) / PLACE_LISTS) * 64u;
    uint32_t cap = (uint32_t)std::min<uint64_t>(cap_full, 1024 + n_pairs / PLACE_LISTS / 8);
    if (c->place_cap >= 0 && (uint64_t)c->place_cap < cap) cap = (uint32_t)c->place_cap;      // dwgsim_hip_debug_option("place_cap"): start too small, exercise the second run
    for (int attempt = 0;; ++attempt) {
        if (ensure(c, c->place_list, sizeof(uint32_t) * (size_t)cap * PLACE_LISTS)) return DWGSIM_HIP_ERR_DEVICE;
        a.place_list = (uint32_t *)c->place_list.p; a.place_list_cap = cap;
        a.place_list_n = (uint32_t *)c->place_aux.p; a.range_rand = reinterpret_cast<uint64_t *>((uint8_t *)c->place_aux.p + PLACE_LISTS * 16 * sizeof(uint32_t));
        HIPC(c, hipMemsetAsync(c->d_pcounters, 0, N_COUNTERS * sizeof(uint64_t), st));
        HIPC(c, hipMemsetAsync(c->place_aux.p, 0, PLACE_LISTS * 16 * sizeof(uint32_t) + sizeof(uint64_t) * ns, st));
        HIPC(c, hipEventRecord(c->ev_cnt0, st));
        launch_place(st, a);
        HIPC(c, hipEventRecord(c->ev_cnt1, st));
        HIPC(c, hipGetLastError());
        HIPC(c, hipMemcpyAsync(c->h_pcounters, c->d_pcounters, N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPC(c, hipMemcpyAsync(c->h_range_rand, a.range_rand, sizeof(uint64_t) * ns, hipMemcpyDeviceToHost, st));
        HIPC(c, hipStreamSynchronize(st));
        { float ms = 0; if (hipEventElapsedTime(&ms, c->ev_cnt0, c->ev_cnt1) == hipSuccess) c->count_us += 1e3 * ms; else (void)hipGetLastError(); }
        if (!(c->h_pcounters[2] & 16) || attempt > 0) break;
        cap = cap_full;
    }
    if (c->h_pcounters[2] & 16) { c->err = "count_random: the list of undecided pairs overflowed twice"; return DWGSIM_HIP_ERR_FAILED; }
    if (c->h_pcounters[2]) { char b[128]; snprintf(b, sizeof b, "\r[dwgsim_core] failed to generate a read after %d trials\n", MAX_ATTEMPTS + 1); c->err = b; return DWGSIM_HIP_ERR_FAILED; }
    c->place_open = c->h_pcounters[5];
    uint64_t total = 0; size_t si = 0;
    for (int q = 0; q < n; ++q) {
        if (r[q].n_pairs == 0) continue;
        total += c->h_range_rand[si];
        if (per_range) per_range[q] = c->h_range_rand[si];
        ++si;
    }
    if (n_random) *n_random = total;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_count_random(dwgsim_hip_ctx_t *c, int contig, uint64_t first_ii, uint64_t n_pairs, uint64_t *n_random)
{
    dwgsim_hip_range_t r; memset(&r, 0, sizeof r); r.contig = contig; r.first_ii = first_ii; r.n_pairs = n_pairs;
    return dwgsim_hip_count_random_ranges(c, &r, 1, n_random, nullptr);
}

int dwgsim_hip_set_fail_carry(dwgsim_hip_ctx_t *c, uint64_t carry)
{
    if (!c) return DWGSIM_HIP_ERR_ARG;
    c->carry_override = carry; c->has_carry_override = true;
    return DWGSIM_HIP_OK;
}

// Enqueue one batch on the compute stream: [chain set] -> memsets -> k_simulate -> abort-rule epilogue -> [k_gzip] -> counters to the slot's
// pinned mirror -> event.  Every buffer the batch needs is in place before anything is enqueued or the chain state moves, so a failing call
// leaves the context as it found it.
// rerun: the batch of this slot once more with larger Ion Torrent read buffers (dwgsim_hip_wait).  Nothing of the chain moves: the random reads and
// the failed attempts of a batch do not depend on the flow model, so the first run's epilogue stands; the kernels start from the chain words the
// first run started from (rand_base = its counters[22]).
static int sim_enqueue(dwgsim_hip_ctx_t *c, const dwgsim_hip_range_t *r, int n, uint64_t rand_base, int slot, bool rerun);
int dwgsim_hip_simulate_ranges_async(dwgsim_hip_ctx_t *c, const dwgsim_hip_range_t *r, int n, uint64_t rand_base, int slot)
{
    return sim_enqueue(c, r, n, rand_base, slot, false);
}
static int sim_enqueue(dwgsim_hip_ctx_t *c, const dwgsim_hip_range_t *r, int n, uint64_t rand_base, int slot, bool rerun)
{
    if (!c || slot < 0 || slot >= DWGSIM_HIP_SLOTS) { if (c) c->err = "bad simulate arguments"; return DWGSIM_HIP_ERR_ARG; }
    Slot &sl = c->slot[slot];
    if (sl.pending) { c->err = "simulate: the slot still holds a batch that was not waited for"; return DWGSIM_HIP_ERR_STATE; }
    const dwgsim_hip_params_t &p = c->prm;
    // lanes per block are a property of the options, so the range table can be laid out before the arguments are complete
    Group *gp = nullptr; std::vector<SimSeg> segs; uint64_t n_pairs = 0; uint32_t nblk = 0; int fixed_max = 0;
    SimArgs a;
    {
        // (fill_sim_args needs a group: take it from the first range)
        Group *g0 = (r && n >= 1) ? get_group(c, r[0].contig) : nullptr;
        if (!g0) { if (r && n >= 1) return DWGSIM_HIP_ERR_ARG; c->err = "bad range arguments"; return DWGSIM_HIP_ERR_ARG; }
        if (const int rc = fill_sim_args(c, *g0, a)) return rc;
    }
    const uint64_t sim_ppb = (uint64_t)(a.sim_threads / (p.length[1] > 0 ? 2 : 1));      // pairs per k_simulate block
    if (const int rc = build_ranges(c, r, n, sim_ppb, &gp, segs, &n_pairs, &nblk, &fixed_max)) return rc;
    Group &g = *gp;
    HIPC(c, hipSetDevice(c->device));
    // (rounds 4-5: a SMALL launch -- up to ~3 rounds of resident blocks: the product's 2^18-pair batches, an E. coli-sized contig -- also ran as two kernels: the
    // first round's three look-backs resolved one after the other, 2^17 / 2^18 / 2^19 pairs of 2 x 150 bp -5 / -11.5 / -2 %: profiles/r04_split.txt)
    // (round 6: with one look-back the single kernel is level with the two-kernel form on launches this small too -- 2^17 / 2^18 / 2^19 pairs of 2 x 150 bp +4 / -3 / +9 %, an E. coli-sized
    // contig +2.5 % -- and the rule is gone: profiles/r06_bench_lines_final.txt 14)
    if (a.split && p.data_type == 2) a.fifo = 1;
    if (c->rand_fixed_len > fixed_max) fixed_max = c->rand_fixed_len;
    // upper bound of one FASTQ record (name tail: 2 positions <= 10 digits, 6 counters, 16 hex digits)
    size_t cap[3] = {0, 0, 0};
    for (int j = 0; j < 2; ++j) if (p.length[j] > 0) cap[j] = (size_t)n_pairs * (size_t)(1 + fixed_max + 120 + 3 + 2 * (p.data_type == 2 ? a.cap : p.length[j]) + 4);
    {   // the one look-back word of the single Illumina kernel: 62 bits for the random reads and the bytes of stream 1 in front of a block
        const uint64_t bytes0 = (uint64_t)cap[0] | 1ull;
        int bw_bytes = 0, bw_pairs = 0;
        while (bw_bytes < 63 && (bytes0 >> bw_bytes)) ++bw_bytes;
        while (bw_pairs < 63 && ((uint64_t)n_pairs >> bw_pairs)) ++bw_pairs;
        a.lb_shift = bw_bytes;
        if (p.data_type != 2 && bw_bytes + bw_pairs > 62) { c->err = "too many pairs in one call"; return DWGSIM_HIP_ERR_ARG; }      // (Illumina and SOLiD; at 2 x 150 bp: more than 2^26 pairs)
    }
    cap[2] = cap[0] + cap[1];
    if (!a.p.has_bwa) cap[0] = cap[1] = 0;
    if (!a.p.has_bfast) cap[2] = 0;
    if (n_pairs) {
        // the slot's text may still be on its way to the host: wait for that before a buffer could be replaced
        if (sl.fetch_in_flight) { HIPC(c, hipStreamWaitEvent(c->stream, sl.ev_fetched, 0)); bool grows = false; for (int t = 0; t < 3; ++t) if (cap[t] + 64 > c->out[slot][t].cap) grows = true; if (grows) HIPC(c, hipEventSynchronize(sl.ev_fetched)); sl.fetch_in_flight = false; }
        for (int t = 0; t < 3; ++t) { if (ensure(c, c->out[slot][t], cap[t] + 64)) return DWGSIM_HIP_ERR_DEVICE; a.out[t] = (uint8_t *)c->out[slot][t].p; }
        if (ensure(c, c->block_rand, sizeof(uint32_t) * (size_t)nblk)) return DWGSIM_HIP_ERR_DEVICE;
        if (!a.split && ensure(c, c->status_all, 4 * sizeof(uint64_t) * (size_t)nblk)) return DWGSIM_HIP_ERR_DEVICE;      // the four look-back arrays, contiguous: one memset per batch
        if (a.split) {      // what the first half hands to the second
            if (ensure(c, c->split_state, sizeof(uint32_t) * (size_t)a.lds_words * SIM_THREADS * (size_t)nblk) || ensure(c, c->split_hand, 16 * (size_t)SIM_THREADS * (size_t)nblk) ||
                ensure(c, c->split_agg, 16 * (size_t)nblk) || ensure(c, c->split_pre, 32 * (size_t)nblk) || ensure(c, c->split_chunk, 32 * ((size_t)nblk / 1024 + 1))) return DWGSIM_HIP_ERR_DEVICE;
            a.split_state = (uint32_t *)c->split_state.p; a.split_hand = (uint32_t *)c->split_hand.p; a.split_agg = (uint32_t *)c->split_agg.p; a.split_pre = (uint64_t *)c->split_pre.p; a.split_chunk = (uint64_t *)c->split_chunk.p;
        }
        if (ensure(c, c->meta, sizeof(uint32_t) * ((size_t)n_pairs + 8))) return DWGSIM_HIP_ERR_DEVICE;      // (+ padding for 16-byte reads)
        const size_t nfb = (size_t)((n_pairs + 256ull * 64 - 1) / (256ull * 64));
        if (ensure(c, c->fail_summ, (nfb * 4 + 2) * sizeof(uint64_t))) return DWGSIM_HIP_ERR_DEVICE;
        if (p.data_type == 2 ? !a.ion_lds : a.sim_threads != SIM_THREADS) {
            // read buffers (Ion Torrent) / staged reads (one-wave blocks): one slot per block an XCD can hold at a time -- what the LDS admits per CU, at most the eight waves of a SIMD (registers can only
            // lower it; too few slots would make blocks wait, never fail) --, handed from block to block inside the XCD (dw_simulate.hip scratch_slot_take)
            const int cu_per_xcd = (c->n_cu >= 64 && c->n_cu % 8 == 0) ? c->n_cu / 8 : c->n_cu;
            const bool ion = p.data_type == 2;
            const int per_cu = ion ? sim_blocks_per_cu(sim_lds_bytes((size_t)a.flow_stack_words, SIM_THREADS, (size_t)a.qb_words, a.fifo != 0), 8)
                                   : sim_blocks_per_cu(sim_lds_bytes(0, SIM_THREADS_LONG, (size_t)a.qb_words, true), 32);       // (one-wave blocks: up to eight per SIMD)
            a.flow_slots = c->flow_slots > 0 ? c->flow_slots : cu_per_xcd * per_cu;
            if ((uint64_t)a.flow_slots > (uint64_t)nblk) a.flow_slots = (int32_t)nblk;
            if (ion) {      // reads that have grown far beyond their estimate (capacity re-runs): fewer slots, at most 4 GB of them (blocks wait for a slot, they never fail for want of one)
                const size_t slot_bytes = (size_t)flow_words_per_lane(a.lds_words) * (size_t)SIM_THREADS * sizeof(uint32_t);
                const size_t fit = std::max<size_t>(1, ((size_t)4 << 30) / (slot_bytes * 8));
                if ((size_t)a.flow_slots > fit) a.flow_slots = (int32_t)fit;
            }
            const size_t words = (ion ? (size_t)flow_words_per_lane(a.lds_words) * (size_t)SIM_THREADS : (size_t)a.lds_words * (size_t)SIM_THREADS_LONG) * (size_t)a.flow_slots * 8;
            if (ensure(c, c->flow_scratch, words * sizeof(uint32_t)) || ensure(c, c->flow_free, sizeof(uint64_t) * (256 + 8 * (size_t)nblk))) return DWGSIM_HIP_ERR_DEVICE;
            a.flow_scratch = (uint32_t *)c->flow_scratch.p; a.flow_free = (uint64_t *)c->flow_free.p;
        }
        if (ensure(c, sl.segs, sizeof(SimSeg) * segs.size())) return DWGSIM_HIP_ERR_DEVICE;
        if (segs.size() > sl.h_segs_cap) {
            if (sl.h_segs) HIPC(c, hipHostFree(sl.h_segs));
            sl.h_segs = nullptr; sl.h_segs_cap = 0;
            HIPC(c, hipHostMalloc((void **)&sl.h_segs, sizeof(SimSeg) * (segs.size() + 64), hipHostMallocDefault));
            sl.h_segs_cap = segs.size() + 64;
        }
        if (c->gzip_on) {
            size_t off = 0;
            for (int t = 0; t < 3; ++t) { off += (size_t)gz_chunks(cap[t]); if (cap[t] && ensure(c, sl.gz_out[t], (size_t)gz_capacity(cap[t]) + 64)) return DWGSIM_HIP_ERR_DEVICE; }
            if (ensure(c, sl.gz_status, sizeof(uint64_t) * (off ? off : 1))) return DWGSIM_HIP_ERR_DEVICE;
        }
    }
    // ---- nothing below fails for want of memory ----
    for (int t = 0; t < 3; ++t) sl.out_bytes[t] = sl.gz_bytes[t] = 0;
    sl.n_pairs = n_pairs; sl.empty = n_pairs == 0;
    // the reference's failure counter (dwgsim.c:635) runs over the pairs of ONE contig in index order: it is carried from the previous
    // batch only when this one continues it; any other range starts from zero unless the caller supplied the carry (sharded jobs)
    const dwgsim_hip_range_t *r_first = nullptr, *r_last = nullptr;
    for (int q = 0; q < n; ++q) if (r[q].n_pairs) { if (!r_first) r_first = &r[q]; r_last = &r[q]; }
    if (!r_first) { r_first = &r[0]; r_last = &r[n - 1]; }
    uint64_t *ch_ptr = c->d_chain; int ch_rand = 0, ch_carry = 0; uint64_t ch_carry_v = 0;      // the running values the launch starts from: set by launch_init with the rest
    if (!rerun) {
    const bool continues = c->chain_contig == r_first->contig && c->chain_next_ii == r_first->first_ii && r_first->first_ii != 0;
    const bool set_carry = c->has_carry_override || !continues;
    const uint64_t carry = c->has_carry_override ? c->carry_override : 0;
    c->has_carry_override = false;
    const bool set_rand = rand_base != DWGSIM_HIP_RAND_CHAIN;
    ch_ptr = c->d_chain; ch_rand = set_rand ? 1 : 0; ch_carry_v = carry; ch_carry = set_carry ? 1 : 0;
    c->chain_contig = r_last->contig; c->chain_next_ii = r_last->first_ii + r_last->n_pairs;
    if (sl.ranges.data() != r) sl.ranges.assign(r, r + n);
    } else {
        ch_ptr = sl.d_rerun_chain; ch_rand = 1; ch_carry_v = 0; ch_carry = 1;
        a.chain = sl.d_rerun_chain;
    }
    if (n_pairs == 0) { if (ch_rand || ch_carry) launch_chain_set(c->stream, ch_ptr, rand_base, ch_rand, ch_carry_v, ch_carry); return DWGSIM_HIP_OK; }
    uint32_t opens = 0;
    for (const SimSeg &s : segs) opens |= s.contig_start;
    memcpy(sl.h_segs, segs.data(), sizeof(SimSeg) * segs.size());
    HIPC(c, hipMemcpyAsync(sl.segs.p, sl.h_segs, sizeof(SimSeg) * segs.size(), hipMemcpyHostToDevice, c->stream));
    a.segs = (const SimSeg *)sl.segs.p; a.n_seg = (int32_t)segs.size(); a.n_blocks = nblk; a.n_pairs = n_pairs;
    a.meta = (uint32_t *)c->meta.p; a.block_rand = (uint32_t *)c->block_rand.p; a.counters = sl.d_counters;
    for (int j = 0; j < 4; ++j) a.status[j] = a.split ? nullptr : (uint64_t *)c->status_all.p + (size_t)j * (size_t)nblk;
    sl.group = c->handles[(size_t)g.first_handle].group;
    // one operation in front of the kernel: counters, look-back words (the four arrays are contiguous), the scratch slots' free lists, the running values
    launch_init(c->stream, sl.d_counters, (uint32_t)N_COUNTERS, a.split ? nullptr : a.status[0], a.split ? 0 : 4 * (uint64_t)nblk, a.flow_free, a.flow_free ? 256 + 8 * (uint64_t)nblk : 0,
                ch_ptr, rand_base, ch_rand, ch_carry_v, ch_carry);
    HIPC(c, hipEventRecord(sl.ev_k0, c->stream));
    launch_simulate(c->stream, a);
    HIPC(c, hipEventRecord(sl.ev_k1, c->stream));
    sl.cap_mult = c->flow_cap_mult;
    if (!rerun) launch_failrule(c->stream, a.meta, n_pairs, opens, (uint64_t *)c->fail_summ.p, sl.d_counters, c->d_chain);
    if (c->gzip_on) {      // the .gz form of every stream, enqueued behind the text (lengths are read on the device: counters[4 + t])
        size_t nch[3], off = 0;
        for (int t = 0; t < 3; ++t) { nch[t] = (size_t)gz_chunks(cap[t]); off += nch[t]; }
        HIPC(c, hipMemsetAsync(sl.gz_status.p, 0, sizeof(uint64_t) * (off ? off : 1), c->stream));
        off = 0;
        for (int t = 0; t < 3; ++t) {
            if (cap[t] == 0) continue;
            launch_gzip(c->stream, a.out[t], &sl.d_counters[4 + t], cap[t], (uint8_t *)sl.gz_out[t].p, (size_t)gz_capacity(cap[t]), (uint64_t *)sl.gz_status.p + off, &sl.d_counters[28 + t], &sl.d_counters[24 + t], &sl.d_counters[2],
                        c->d_crc_table, c->d_crc_shift);
            off += nch[t];
        }
    }
    HIPC(c, hipEventRecord(sl.ev_end, c->stream));
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(sl.h_counters, sl.d_counters, N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipEventRecord(sl.ev_done, c->stream));
    sl.pending = true;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_simulate_async(dwgsim_hip_ctx_t *c, int contig, uint64_t first_ii, uint64_t n_pairs, uint64_t rand_base, int slot)
{
    dwgsim_hip_range_t r; memset(&r, 0, sizeof r); r.contig = contig; r.first_ii = first_ii; r.n_pairs = n_pairs;
    return dwgsim_hip_simulate_ranges_async(c, &r, 1, rand_base, slot);
}

int dwgsim_hip_wait(dwgsim_hip_ctx_t *c, int slot, dwgsim_hip_batch_t *out)
{
    if (!c || slot < 0 || slot >= DWGSIM_HIP_SLOTS) { if (c) c->err = "bad slot"; return DWGSIM_HIP_ERR_ARG; }
    Slot &sl = c->slot[slot];
    if (out) memset(out, 0, sizeof *out);
    if (sl.empty) { sl.pending = false; return DWGSIM_HIP_OK; }
    if (!sl.pending) { c->err = "wait: no batch was enqueued on this slot"; return DWGSIM_HIP_ERR_STATE; }
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipEventSynchronize(sl.ev_done));
    sl.pending = false;
    uint64_t *h = sl.h_counters;
    if ((h[2] & 2) && !(h[2] & ~(2ull | 8ull)) && c->prm.data_type == 2) {
        // a read outgrew its flow-space buffers: the reference doubles them and goes on (dwgsim.c:296-311); here the batch runs again with twice the
        // capacity (kept for the rest of the job).  Its random reads and failed attempts are settled before the flow model runs, so the chain words
        // the epilogue of the first run has already moved on stand; the second run starts from the ones the first started from (counters[22]).
        uint64_t keep[8]; for (int q = 0; q < 6; ++q) keep[q] = h[16 + q];
        const uint64_t base = h[22];
        while (h[2] & 2) {
            if (sl.cap_mult >= c->flow_cap_mult) {      // (another batch may have doubled it already)
                // the reference doubles without end (dwgsim.c:296-311); here up to FLOW_CAP_MAX bases per read -- two thousand times a 400-base read, in
                // scratch slots whose number shrinks as they grow (sim_enqueue).  (Rounds 3-4 stopped at 16 x the estimate: 3 of 300 random flow orders)
                const int lmx = c->prm.length[0] > c->prm.length[1] ? c->prm.length[0] : c->prm.length[1];
                const double emx = c->prm.e_start[0] > c->prm.e_start[1] ? c->prm.e_start[0] : c->prm.e_start[1];
                const int64_t base_cap = c->flow_cap_forced > 0 ? c->flow_cap_forced : flow_read_capacity(lmx, emx, c->flow);
                if (base_cap * (int64_t)c->flow_cap_mult * 2 > (int64_t)FLOW_CAP_MAX) break;
                c->flow_cap_mult *= 2;
            }
            if (const int rc = sim_enqueue(c, sl.ranges.data(), (int)sl.ranges.size(), base, slot, true)) return rc;
            HIPC(c, hipEventSynchronize(sl.ev_done));
            sl.pending = false;
        }
        for (int q = 0; q < 6; ++q) h[16 + q] = keep[q];
    }
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
        HIPC(c, hipEventElapsedTime(&out->kernel_ms, sl.ev_k0, sl.ev_end));       // ... with the abort-rule epilogue and the gzip kernels
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

// Page-locked host memory: anonymous memory on TRANSPARENT HUGE PAGES, touched by the calling thread(s) (so it lies on their NUMA node) and then registered
// with the runtime.  Page-locking is paid per page: hipHostMalloc locks 4 KB pages at 5-6 GB/s -- 60-85 ms for the 300 MB staging of a chromosome, on
// the critical path of a job's first batch, and the KFD's per-process lock makes every other allocation of the process wait meanwhile -- while 2 MB
// pages register at 17-55 GB/s (tools/ubench_pin.hip, profiles/r05_genome_trace.txt); copies into either run at the same 49-50 GB/s.  Where the kernel
// grants no huge pages (transparent_hugepage = never) the same code simply registers 4 KB pages.
namespace {
struct HostRegion { uint8_t *base; size_t len; };
std::mutex g_host_m;
std::vector<HostRegion> g_host;      // (a handful per process)
}
static bool in_host_registry(const void *p)
{
    std::lock_guard<std::mutex> lk(g_host_m);
    for (const HostRegion &r : g_host) if ((const uint8_t *)p >= r.base && (const uint8_t *)p < r.base + r.len) return true;
    return false;
}
void *dwgsim_hip_host_alloc(size_t bytes)
{
    const size_t H = (size_t)2 << 20;
    const size_t len = ((bytes ? bytes : 1) + H - 1) / H * H;
    uint8_t *raw = (uint8_t *)mmap(nullptr, len + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw == MAP_FAILED) return nullptr;
    uint8_t *p = (uint8_t *)(((uintptr_t)raw + H - 1) & ~(uintptr_t)(H - 1));
    if (p > raw) munmap(raw, (size_t)(p - raw));
    if (p + len < raw + len + H) munmap(p + len, (size_t)(raw + len + H - (p + len)));
    (void)madvise(p, len, MADV_HUGEPAGE);
    // first touch (a fault per huge page zeroes 2 MB): large buffers by a few threads at once
    auto touch = [p](size_t from, size_t upto) { for (size_t o = from; o < upto; o += 4096) ((volatile uint8_t *)p)[o] = 0; };
    const int nt = len >= ((size_t)64 << 20) ? 4 : 1;
    if (nt == 1) touch(0, len);
    else {
        std::vector<std::thread> th; const size_t per = (len / H + (size_t)nt - 1) / (size_t)nt * H;
        for (int k = 0; k < nt; ++k) { const size_t a = std::min(len, (size_t)k * per), b = std::min(len, a + per); if (b > a) th.emplace_back(touch, a, b); }
        for (auto &t : th) t.join();
    }
    if (hipHostRegister(p, len, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); munmap(p, len); return nullptr; }
    { std::lock_guard<std::mutex> lk(g_host_m); g_host.push_back(HostRegion{p, len}); }
    return p;
}
void dwgsim_hip_host_free(void *p)
{
    if (!p) return;
    HostRegion r{nullptr, 0};
    {
        std::lock_guard<std::mutex> lk(g_host_m);
        for (size_t i = 0; i < g_host.size(); ++i) if (g_host[i].base == (uint8_t *)p) { r = g_host[i]; g_host.erase(g_host.begin() + (long)i); break; }
    }
    if (!r.base) return;      // (not one of ours)
    (void)hipHostUnregister(r.base);
    munmap(r.base, r.len);
}

// Copies on the context's second stream: a batch that is being copied out of one slot overlaps with the kernels filling the other.
int dwgsim_hip_fetch_async(dwgsim_hip_ctx_t *c, int slot, int stream, void *host_dst, size_t cap)
{
    if (!c || slot < 0 || slot >= DWGSIM_HIP_SLOTS || stream < 0 || stream > 2 || (!host_dst && cap)) { if (c) c->err = "bad fetch arguments"; return DWGSIM_HIP_ERR_ARG; }
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
    if (!c || slot < 0 || slot >= DWGSIM_HIP_SLOTS || stream < 0 || stream > 2 || (!host_dst && cap)) { if (c) c->err = "bad fetch arguments"; return DWGSIM_HIP_ERR_ARG; }
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
    if (!c || slot < 0 || slot >= DWGSIM_HIP_SLOTS) { if (c) c->err = "bad slot"; return DWGSIM_HIP_ERR_ARG; }
    Slot &sl = c->slot[slot];
    if (!sl.fetch_in_flight) return DWGSIM_HIP_OK;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipEventSynchronize(sl.ev_fetched));
    sl.fetch_in_flight = false;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_fetch(dwgsim_hip_ctx_t *c, int slot, int stream, void *host_dst, size_t cap)
{
    if (!c || slot < 0 || slot >= DWGSIM_HIP_SLOTS || stream < 0 || stream > 2 || (!host_dst && cap)) return DWGSIM_HIP_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    Slot &sl = c->slot[slot];
    if (sl.pending) { c->err = "fetch: wait for the batch first (its sizes are not known yet)"; return DWGSIM_HIP_ERR_STATE; }
    const size_t n = (size_t)sl.out_bytes[stream];
    if (n > cap) { c->err = "fetch: destination too small"; return DWGSIM_HIP_ERR_ARG; }
    if (n == 0) return DWGSIM_HIP_OK;
    if (is_page_locked(host_dst)) {   // a pinned (page-locked / registered) destination takes one direct copy at link speed
        HIPC(c, hipMemcpyAsync(host_dst, c->out[slot][stream].p, n, hipMemcpyDeviceToHost, c->copy_stream));
        HIPC(c, hipStreamSynchronize(c->copy_stream));
        return DWGSIM_HIP_OK;
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
    if (!c || slot < 0 || slot >= DWGSIM_HIP_SLOTS || stream < 0 || stream > 2 || !count) return DWGSIM_HIP_ERR_ARG;
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
// "walk_seg_min" = n runs the walk's two serial scans in their segmented form from a capacity of n candidates on (default 16384; 0 restores it),
// "place_cap" = n gives the lists of pairs that k_place leaves open room for n entries each (exercises the second, full-size run),
// "split" = 0 / 1 runs the Illumina read kernel as one kernel with look-backs / as two kernels with the offsets computed in between (-1: two for reads of
// up to 100 bases).
int dwgsim_hip_debug_option(dwgsim_hip_ctx_t *c, const char *key, int64_t value)
{
    if (!c || !key) return DWGSIM_HIP_ERR_ARG;
    if (!strcmp(key, "justify_seq")) c->seq_justify = value != 0;
    else if (!strcmp(key, "dense_view")) c->dense_view = value != 0;
    else if (!strcmp(key, "walk_cap")) c->walk_cap = value;
    else if (!strcmp(key, "site_slots")) c->site_slots = (int)value;
    else if (!strcmp(key, "site_slot_cap")) c->site_slot_cap = value;
    else if (!strcmp(key, "phases")) c->phases = value != 0;
    else if (!strcmp(key, "writer")) c->writer = (int)value;
    else if (!strcmp(key, "sim_threads")) c->force_threads = (int)value;
    else if (!strcmp(key, "walk_seg_min")) walk_debug_seg_min((uint32_t)value);      // (process-wide)
    else if (!strcmp(key, "place_cap")) c->place_cap = value;
    else if (!strcmp(key, "split")) c->split = (int)value;
    else if (!strcmp(key, "flow_slots")) c->flow_slots = (int)value;
    else if (!strcmp(key, "ion_lds")) c->ion_lds = (int)value;
    else if (!strcmp(key, "flow_cap")) c->flow_cap_forced = (int)value;
    else { c->err = "unknown debug option"; return DWGSIM_HIP_ERR_ARG; }
    return DWGSIM_HIP_OK;
}

// ... and values to read back: "place_open" = pairs the last dwgsim_hip_count_random* call could not settle from the coarse summaries;
// "walk_us" / "count_us" = accumulated HIP-event time (microseconds) of the walk chains / random-read counts of this context
int dwgsim_hip_debug_get(dwgsim_hip_ctx_t *c, const char *key, int64_t *value)
{
    if (!c || !key || !value) return DWGSIM_HIP_ERR_ARG;
    if (!strcmp(key, "place_open")) *value = (int64_t)c->place_open;
    else if (!strcmp(key, "flow_cap_mult")) *value = (int64_t)c->flow_cap_mult;
    else if (!strcmp(key, "walk_us")) *value = (int64_t)c->walk_us;           // HIP-event time of the walk chains waited for so far (start of the chain to its end, on the walk stream)
    else if (!strcmp(key, "count_us")) *value = (int64_t)c->count_us;         // ... of the random-read counts (k_place .. k_range_counts)
    else { c->err = "unknown debug value"; return DWGSIM_HIP_ERR_ARG; }
    return DWGSIM_HIP_OK;
}

} // extern "C"
