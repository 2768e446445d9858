// dw_job.cpp -- the job level of the C-ABI (include/dwgsim_hip.h, dwgsim_hip_job_*): the whole of dwgsim_core() (src/dwgsim.c:419-1121) on any
// number of GPUs, behind five calls.  It is written on top of the context-level entry points of the same header (one context per device) and
// owns what a caller of those would otherwise have to choreograph:
//   * the scheduling arithmetic of the contig loop (dwgsim.c:519-625): pairs per contig, skip rules, region lengths, -N remainders;
//   * GROUPS: consecutive contigs up to a size are resident, walked and simulated together (a scaffold-level assembly costs one walk chain
//     and a few launches per group, not per contig); their sequence is staged once in page-locked memory and uploaded asynchronously;
//   * the pipeline of every device: upload + walk of group k+1 on the walk stream while the batches of group k run; two batches in flight;
//     finished text (or the gzip members made on the GPU) copied into page-locked buffers behind the kernels;
//   * SHARDING: the pairs of a group, in file order, are cut into batches of read-index ranges; batch b belongs to device b mod n.  Every
//     device walks every group itself (deterministic, cheap: no broadcast).  What crosses devices is two integers per batch -- the random-read
//     count in front of it (rand_ii, dwgsim.c:1042,1096: every device counts the random reads of its batches with k_place before it simulates,
//     one host-side prefix sum per group) and the abort rule's summary (dwgsim.c:635, :833-843; joined in order at the end of the group).
//     No collective, no RCCL, no device-to-device traffic;
//   * ORDER: one delivery thread per output stream hands the batches to the caller's sink in file order.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stdarg.h>
#include <time.h>
#include <limits.h>
#include <algorithm>
#include <string>
#include <vector>
#include <deque>
#include <array>
#include <memory>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <sched.h>
#include "../../include/dwgsim_hip.h"

namespace {

// A device's worker thread -- it allocates the device's page-locked output buffers and waits for its copies -- runs on the cores of the NUMA node the
// GPU hangs off (dwgsim_hip_device_numa_node: /sys/bus/pci/devices/<bus id>/numa_node), so that on a two-socket node no device copies its members
// across the socket link.  Quietly does nothing where the node is unknown (-1: single-socket boxes, containers without sysfs), where the node's
// cores are not among the ones the process may use, or with DWGSIM_HIP_NO_PIN set.
void pin_thread_to_device_node(int device)
{
    if (getenv("DWGSIM_HIP_NO_PIN")) return;
    const int node = dwgsim_hip_device_numa_node(device);
    if (node < 0) return;
    char path[128]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return;
    char list[4096]; const bool got = fgets(list, sizeof list, f) != nullptr; fclose(f);
    if (!got) return;
    cpu_set_t allowed, want; CPU_ZERO(&allowed); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    int n_want = 0;
    for (const char *q = list; *q && *q != '\n';) {      // "0-31,64-95"
        char *e; long a = strtol(q, &e, 10), b = a;
        if (e == q) break;
        if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &want); ++n_want; }
        q = *e == ',' ? e + 1 : e;
    }
    if (n_want > 0) (void)sched_setaffinity(0, sizeof want, &want);
}

struct PinBuf { char *p[3] = {nullptr, nullptr, nullptr}; size_t cap[3] = {0, 0, 0}; };      // the output of one batch in page-locked memory

struct BatchOut {             // one batch on its way to the sink
    PinBuf *buf = nullptr; int lane = -1;
    size_t n[3] = {0, 0, 0}, text_n[3] = {0, 0, 0};
    bool ready = false; int left = 0;       // streams that still have to deliver it
    uint64_t pairs = 0;
};

struct GroupJob {
    int id = 0;
    std::vector<std::string> names; std::vector<int64_t> lens, l_eff, n_pairs; std::vector<uint32_t> cindex;
    int stage_slot = -1; std::vector<const uint8_t *> ptrs;      // the sequence in page-locked staging, group layout
    int stage_users = 0;
    uint64_t pairs = 0;
    int nd = 1;                                                   // devices that share the group's batches
    std::vector<std::vector<dwgsim_hip_range_t>> batches;         // ranges with contig = member ordinal (each device adds its own handle base)
    std::vector<uint64_t> batch_pairs, batch_rand;                // pairs / counted random reads per batch
    int counted = 0;                                              // devices that have published their batches' counts
    bool base_known = false; uint64_t rand_base = 0;              // random reads in front of the group
    std::vector<std::array<uint64_t, 4>> fail_seg; std::vector<uint64_t> got_rand;
    std::vector<BatchOut> out;
    // reads_at (pieces with their place in the stream, delivered by several threads): a batch's sizes are known when its kernels are done (stage A);
    // its offsets when the sizes of every batch in front of it are (dwgsim_hip_job::next_off, in file order across groups)
    std::vector<std::array<uint64_t, 3>> sz, off; std::vector<uint8_t> sized, off_ok;
    int batches_done = 0; bool closed = false;
    int joined = 0; uint64_t fail_acc[4] = {0, 0, 0, 0};          // abort rule: batches 0 .. joined-1 are simulated and their summaries joined, in order
    int mut_done = 0;                                             // (device 0) mutation text delivered
};

} // namespace

struct dwgsim_hip_job {
    dwgsim_hip_params_t prm; std::string prefix, flow;
    dwgsim_hip_job_sink_t sink; dwgsim_hip_job_options_t opt;
    std::vector<int> devices; std::vector<dwgsim_hip_ctx_t *> ctx;
    int ND = 0;
    bool want_mut = true, want_reads = true, gzip = true;
    uint64_t batch_pairs = 1u << 18, group_bp = 32u << 20, min_share = 65536;      // (batches of 2^18 pairs: 190 MB of text, 94 MB of members -- measured against 2^17 .. 2^20: the smaller the batch, the shorter a job's fill and drain and the less page-locked memory there is to hand back; below 2^18 the whole-genome rate stops improving)
    // contig table, scheduling state (dwgsim.c:465-478, :519-625)
    std::vector<std::string> tab_names; std::vector<int64_t> tab_lens; bool have_table = false;
    uint64_t tot_len = 0; int n_ref = 0; int64_t n_sim = 0; int prev_skip = 0; uint32_t next_index = 0;
    std::string regions_path, mutin_path; int mutin_type = -1;
    // staging of the sequence: page-locked buffers handed from the adding thread to the device workers
    static constexpr int N_STAGE = 2;      // (one being filled while the other's group is uploaded and walked; a third bought nothing and is 0.25 GB of page-locked memory for a genome)
    uint8_t *stage[N_STAGE] = {nullptr, nullptr}; size_t stage_cap[N_STAGE] = {0, 0}; bool stage_busy[N_STAGE] = {false, false};
    size_t stage_want = 0;        // from the contig table: room for the largest group, so that a staging buffer is page-locked once
    std::shared_ptr<GroupJob> pending; size_t pending_bytes = 0;           // the group being filled
    struct Open { bool open = false; std::string name; int64_t l = 0, st = 0, total = 0; uint32_t ci = 0; } open;      // the contig between begin_contig and commit_contig
    // shared state
    std::mutex m; std::condition_variable cv;
    std::deque<std::shared_ptr<GroupJob>> groups;      // dispatched, not yet retired (front = oldest)
    int n_dispatched = 0; bool no_more = false;
    std::vector<int> next_group;                       // per device: id of the group it takes next
    std::atomic<bool> failed{false}; std::string err;
    std::vector<std::thread> workers; std::thread deliver[3];
    std::vector<std::thread> deliver_at;               // reads_at: one thread per device and stream
    uint64_t next_off[3] = {0, 0, 0}; int off_gid = 0, off_b = 0;      // reads_at: the next piece's offsets; the batch (group id, index) they belong to
    bool has_sink() const { return sink.reads != nullptr || sink.reads_at != nullptr; }
    bool started = false, finished = false;
    uint64_t delivered_pairs = 0; uint64_t total_rand = 0;
    // page-locked output buffers per device
    std::vector<std::vector<std::unique_ptr<PinBuf>>> bufs; std::vector<std::vector<PinBuf *>> free_bufs;
    bool tracing = false; double t0 = 0;
    int max_bufs = 8;      // (the batches in flight per device, one per output set of the context, + what the delivery threads hold)
    // DWGSIM_HIP_SOLO=r/W (measurement only): the ONE device of this job does exactly what device r of a W-device job does -- walks every group it
    // takes part in, counts and simulates batches r, r + W, ... of each, copies them out, delivers them -- and nothing of the other devices' work.  The
    // path has no device-to-device traffic, so that is what device r's GPU and PCIe link would carry.  The other devices' batches are treated as
    // delivered and their random-read counts as zero: the OUTPUT of such a run is a share of the job with wrong rand_ii offsets -- for a counting
    // sink and a clock, not for files.  VD = devices the batches are dealt to (W, or ND), vrank(d) = which of them device d is.
    int solo_rank = -1, VD = 0;
    int vrank(int d) const { return solo_rank >= 0 ? solo_rank : d; }
};

namespace {

// DWGSIM_HIP_TRACE=1 (analysis): when the job's stages happen, in seconds since the job was created, on stderr
double mono_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
void trace(dwgsim_hip_job *j, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

void job_fail(dwgsim_hip_job *j, const std::string &what)
{
    std::lock_guard<std::mutex> g(j->m);
    if (!j->failed.exchange(true)) j->err = what;
    j->cv.notify_all();
}

int arg_error(dwgsim_hip_job *j, int code, const char *what)      // a call that is refused: the text for last_error, the job itself goes on
{
    if (j) { std::lock_guard<std::mutex> g(j->m); if (!j->failed.load()) j->err = what; }
    return code;
}

void say(dwgsim_hip_job *j, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void say(dwgsim_hip_job *j, const char *fmt, ...)
{
    char b[4608]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap);
    if (j->sink.message) j->sink.message(j->sink.user, b); else if (!j->opt.quiet) fputs(b, stderr);
}

void trace(dwgsim_hip_job *j, const char *fmt, ...)
{
    if (!j->tracing) return;
    char b[256]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap);
    fprintf(stderr, "[trace %8.4f] %s\n", mono_s() - j->t0, b);
}

std::shared_ptr<GroupJob> group_by_id(dwgsim_hip_job *j, int id)      // j->m held
{
    for (auto &g : j->groups) if (g->id == id) return g;
    return nullptr;
}

// reads_at: offsets for every batch whose predecessors' sizes are all known (j->m held)
void assign_offsets(dwgsim_hip_job *j)
{
    if (!j->sink.reads_at) return;
    for (;;) {
        auto g = group_by_id(j, j->off_gid);
        if (!g) { if (j->off_gid < j->n_dispatched) { ++j->off_gid; j->off_b = 0; continue; } break; }      // (retired already: it had no batches)
        const int nb = j->want_reads ? (int)g->batches.size() : 0;
        while (j->off_b < nb && g->sized[(size_t)j->off_b]) {
            const size_t b = (size_t)j->off_b;
            for (int s = 0; s < 3; ++s) { g->off[b][(size_t)s] = j->next_off[s]; j->next_off[s] += g->sz[b][(size_t)s]; }
            g->off_ok[b] = 1; ++j->off_b;
        }
        if (j->off_b < nb) break;
        ++j->off_gid; j->off_b = 0;
    }
}

// ---- one device ----
struct Worker {
    dwgsim_hip_job *j; int d; dwgsim_hip_ctx_t *x;
    std::shared_ptr<GroupJob> prepped; int prepped_handle = -1;      // the group whose upload + walk is already enqueued
    bool prepped_waited = false, prepped_counted = false;           // ... whose walk has been waited for / whose random reads this device has counted already
    bool first_batch_of_job = true;
    // The batches in flight, oldest first -- one per output set of the context, and they stay in flight ACROSS the end of a group: batch k is
    // enqueued (kernels); then the copy-out of batch k-1 is issued as soon as its kernels are done (stage A); then the copy-out of the oldest batch
    // is waited for and the batch published (stage B), which frees its slot for batch k+1.  With four slots two batches' copies are queued on the
    // copy stream while the worker does anything else, so the link does not wait for the host.  Rounds 3-4: two slots, the copy of a batch
    // issued only behind the enqueue of the next one (33 GB/s over a link that carries 54); round 5's first form: three slots, drained at every
    // group's end, where the worker then made the next group's mutation text -- 25 to 50 ms per chromosome during which the copy engine stood still
    // (profiles/r05_genome_trace.txt: busy 0.75).
    struct Pending { int slot = 0, b = 0, h = -1; bool a_done = false, last_of_group = false; std::shared_ptr<GroupJob> g; dwgsim_hip_batch_t bt; BatchOut bo; };
    std::deque<Pending> fl;
    uint64_t kk = 0;      // batches enqueued so far (batch kk takes slot kk mod DWGSIM_HIP_SLOTS)
    // (device 0) the mutation text: the worker only fetches a group's list of mutated cells; a thread of its own makes the text and hands it to the sink
    struct MutTask { dwgsim_hip_mutlist_t *list; std::vector<std::string> names; };
    std::thread mut_thread; std::mutex mm; std::condition_variable mcv; std::deque<MutTask> mq; bool mut_quit = false;

    bool ok() const { return !j->failed.load(); }
    void fail_ctx() { job_fail(j, std::string("dwgsim-hip: ") + dwgsim_hip_last_error(x)); }

    void mut_loop()      // mut_print (mut.c:781-893), groups and contigs in order
    {
        for (;;) {
            MutTask t;
            {
                std::unique_lock<std::mutex> lk(mm);
                mcv.wait(lk, [&]() { return mut_quit || !mq.empty(); });
                if (mq.empty()) return;
                t = std::move(mq.front()); mq.pop_front();
            }
            for (size_t k = 0; k < t.names.size() && ok(); ++k) {
                const char *tx, *v; size_t tl, vl;
                if (dwgsim_hip_mutlist_text(t.list, (int)k, &tx, &tl, &v, &vl) < 0) { job_fail(j, "dwgsim-hip: the mutation text could not be made"); break; }
                if (j->sink.mutations(j->sink.user, t.names[k].c_str(), tx, tl, v, vl) != 0) { job_fail(j, "dwgsim-hip: the sink refused the mutation text"); break; }
            }
            dwgsim_hip_mutlist_free(t.list);
        }
    }
    void mut_stop()
    {
        if (!mut_thread.joinable()) return;
        { std::lock_guard<std::mutex> lk(mm); mut_quit = true; }
        mcv.notify_all();
        mut_thread.join();
    }

    int prep(const std::shared_ptr<GroupJob> &g)      // upload (asynchronous: the staging is page-locked and in group layout) and enqueue the walk
    {
        const int n = (int)g->names.size();
        std::vector<const char *> nm((size_t)n);
        for (int k = 0; k < n; ++k) nm[(size_t)k] = g->names[(size_t)k].c_str();
        const int h = dwgsim_hip_add_contigs(x, n, nm.data(), g->ptrs.data(), g->lens.data(), g->cindex.data());
        if (h < 0) { fail_ctx(); return -1; }
        if (!j->regions_path.empty())      // the `l` of fragment placement: the region length -- or the full length for the last contig of an -N run (dwgsim.c:535-537)
            for (int k = 0; k < n; ++k) if (dwgsim_hip_contig_set_placement_length(x, h + k, g->l_eff[(size_t)k]) < 0) { fail_ctx(); return -1; }
        if (dwgsim_hip_mutate_async(x, h) < 0) { fail_ctx(); return -1; }
        return h;
    }

    // the walk has finished: the staging is no longer needed by this device
    bool walked(const std::shared_ptr<GroupJob> &g, int h)
    {
        if (dwgsim_hip_mutate_wait(x, h) < 0) { fail_ctx(); return false; }
        std::lock_guard<std::mutex> lk(j->m);
        if (--g->stage_users == 0) { j->stage_busy[g->stage_slot] = false; j->cv.notify_all(); }
        return true;
    }

    // random reads of my batches of the group, counted without producing them (k_place on the walk stream): one launch, one count per batch
    bool count_mine(const std::shared_ptr<GroupJob> &g, int h)
    {
        if (j->vrank(d) < g->nd) {
            std::vector<dwgsim_hip_range_t> all; std::vector<int> owner;
            for (int b = j->vrank(d); b < (int)g->batches.size(); b += g->nd) for (auto q : g->batches[(size_t)b]) { q.contig += h; all.push_back(q); owner.push_back(b); }
            std::vector<uint64_t> per(all.size(), 0); uint64_t tot = 0;
            if (!all.empty() && dwgsim_hip_count_random_ranges(x, all.data(), (int)all.size(), &tot, per.data()) < 0) { fail_ctx(); return false; }
            std::lock_guard<std::mutex> lk(j->m);
            for (size_t q = 0; q < all.size(); ++q) g->batch_rand[(size_t)owner[q]] += per[q];
            ++g->counted;
            if (j->solo_rank >= 0) g->counted = g->nd;      // (the others' counts are taken as zero)
            j->cv.notify_all();
        }
        return true;
    }

    // between two batches: whatever can be done for the NEXT group without waiting -- upload + walk as soon as it has been handed over, the count
    // of its random reads as soon as its walk has finished -- so that its first batch follows this group's last one at once
    bool look_ahead(const std::shared_ptr<GroupJob> &g)
    {
        if (!prepped) {
            std::shared_ptr<GroupJob> nx;
            { std::lock_guard<std::mutex> lk(j->m); nx = group_by_id(j, g->id + 1); }
            if (nx && takes_part(*nx)) { const int nh = prep(nx); if (nh < 0) return false; prepped = nx; prepped_handle = nh; prepped_waited = prepped_counted = false; }
        }
        if (prepped && !prepped_waited && dwgsim_hip_mutate_poll(x, prepped_handle) == 1) { if (!walked(prepped, prepped_handle)) return false; prepped_waited = true; }
        if (prepped && prepped_waited && !prepped_counted && j->VD > 1 && j->want_reads) { if (!count_mine(prepped, prepped_handle)) return false; prepped_counted = true; }
        return true;
    }

    PinBuf *acquire(const uint64_t need[3])
    {
        PinBuf *b = nullptr;
        {
            std::unique_lock<std::mutex> lk(j->m);
            auto &fr = j->free_bufs[(size_t)d]; auto &all = j->bufs[(size_t)d];
            j->cv.wait(lk, [&]() { return !fr.empty() || (int)all.size() < j->max_bufs || j->failed.load(); });
            if (j->failed.load()) return nullptr;
            if (!fr.empty()) { b = fr.back(); fr.pop_back(); }
            else { all.push_back(std::make_unique<PinBuf>()); b = all.back().get(); }
        }
        for (int s = 0; s < 3; ++s) if (need[s] > b->cap[s]) {
            dwgsim_hip_host_free(b->p[s]);
            b->cap[s] = (size_t)need[s] + (size_t)need[s] / 8 + 4096;
            b->p[s] = (char *)dwgsim_hip_host_alloc(b->cap[s]);
            if (!b->p[s]) { b->cap[s] = 0; job_fail(j, "dwgsim-hip: cannot allocate page-locked host memory for the output"); return nullptr; }
        }
        return b;
    }

    // stage A: the batch's kernels are done -- its sizes are known: the copy-out is issued; the last batch of a group also gives the group's memory back
    bool stage_a(Pending &pb)
    {
        if (pb.a_done) return true;
        if (dwgsim_hip_wait(x, pb.slot, &pb.bt) < 0) { fail_ctx(); return false; }
        pb.a_done = true;
        pb.bo = BatchOut(); pb.bo.lane = d; pb.bo.pairs = pb.bt.n_pairs;
        if (j->has_sink()) {
            PinBuf *tb = acquire(j->gzip ? pb.bt.gz_bytes : pb.bt.bytes);
            if (!tb) return false;
            pb.bo.buf = tb;
            for (int s = 0; s < 3; ++s) {
                pb.bo.n[s] = j->gzip ? pb.bt.gz_bytes[s] : pb.bt.bytes[s]; pb.bo.text_n[s] = pb.bt.bytes[s];
                if (pb.bo.n[s] && (j->gzip ? dwgsim_hip_fetch_gz_async(x, pb.slot, s, tb->p[s], tb->cap[s]) : dwgsim_hip_fetch_async(x, pb.slot, s, tb->p[s], tb->cap[s])) < 0) { fail_ctx(); return false; }
                if (pb.bo.n[s]) ++pb.bo.left;
            }
            if (j->sink.reads_at) {      // the sizes are known: this batch's offsets, and those of any batch behind it that was only waiting for them
                std::lock_guard<std::mutex> lk(j->m);
                for (int s = 0; s < 3; ++s) pb.g->sz[(size_t)pb.b][(size_t)s] = pb.bo.n[s];
                pb.g->sized[(size_t)pb.b] = 1;
                assign_offsets(j);
                j->cv.notify_all();
            }
        }
        if (pb.b < 2 * j->VD) trace(j, "dev %d group %d: batch %d kernels done, copy issued (%.1f MB)", d, pb.g->id, pb.b, (pb.bo.n[0] + pb.bo.n[1] + pb.bo.n[2]) / 1e6);
        if (pb.last_of_group && dwgsim_hip_drop_contig(x, pb.h) < 0) { fail_ctx(); return false; }      // (the kernels of the group's last batch are done: nothing reads it any more)
        return true;
    }
    // stage B: the copy-out has landed: the batch is published (the delivery threads hand it to the sink in file order, behind the abort rule's verdict)
    bool stage_b(Pending &pb)
    {
        if (j->has_sink() && dwgsim_hip_fetch_wait(x, pb.slot) < 0) { fail_ctx(); return false; }
        if (pb.b < 2 * j->VD) trace(j, "dev %d group %d: batch %d landed", d, pb.g->id, pb.b);
        const dwgsim_hip_batch_t &bt = pb.bt; BatchOut &bo = pb.bo; GroupJob *g = pb.g.get();
        uint64_t shown = 0; bool aborted = false;
        {
            std::lock_guard<std::mutex> lk(j->m);
            for (int q = 0; q < 4; ++q) g->fail_seg[(size_t)pb.b][(size_t)q] = bt.fail_seg[q];
            g->got_rand[(size_t)pb.b] = bt.n_random;
            bo.ready = true;
            if (bo.left == 0 && bo.buf) { j->free_bufs[(size_t)d].push_back(bo.buf); bo.buf = nullptr; }
            g->out[(size_t)pb.b] = bo;
            ++g->batches_done;
            // the abort rule (dwgsim.c:635, :833-843) over the batches of several devices: the summaries are joined in read-index order as soon
            // as the batches in front are complete -- a batch goes to the sink only behind its verdict (deliver_loop waits for `joined`)
            while (!aborted && g->joined < (int)g->out.size() && g->out[(size_t)g->joined].ready) {
                if (dwgsim_hip_failseg_join(g->fail_acc, g->fail_seg[(size_t)g->joined].data())) aborted = true; else ++g->joined;
            }
            shown = (j->delivered_pairs += bt.n_pairs);
            j->cv.notify_all();
        }
        if (aborted) { job_fail(j, "\r[dwgsim_core] failed to generate a read after 10001 trials\n"); return false; }
        if (!j->opt.quiet) { char t[64]; snprintf(t, sizeof t, "\r[dwgsim_core] %llu", (unsigned long long)shown); if (j->sink.message) j->sink.message(j->sink.user, t); else fputs(t, stderr); }      // (outside the lock: a sink may call back into the job)
        return true;
    }
    // everything in flight goes through both stages (before the worker waits for anything another thread can only provide once these batches are published)
    bool drain()
    {
        while (!fl.empty()) {
            for (auto &pb : fl) if (!stage_a(pb)) return false;
            if (!stage_b(fl.front())) return false;
            fl.pop_front();
        }
        return true;
    }
    // the job has failed: leave the context idle -- kernels and copies of whatever is in flight are waited for, nothing more is published
    void abandon()
    {
        for (auto &pb : fl) { if (!pb.a_done) { dwgsim_hip_batch_t bt; (void)dwgsim_hip_wait(x, pb.slot, &bt); } (void)dwgsim_hip_fetch_wait(x, pb.slot); }
        fl.clear();
    }

    void run()
    {
        pin_thread_to_device_node(j->devices[(size_t)d]);
        if (j->vrank(d) == 0 && j->want_mut && j->sink.mutations) mut_thread = std::thread([this]() { mut_loop(); });
        bool fine = true;
        for (;;) {
            std::shared_ptr<GroupJob> g;
            {
                std::unique_lock<std::mutex> lk(j->m);
                const int want = j->next_group[(size_t)d];
                auto there = [&]() { return j->failed.load() || group_by_id(j, want) || (j->no_more && want >= j->n_dispatched); };
                if (!there() && !fl.empty()) {      // the next group may only be handed over once the one in front has retired -- which takes the batches still in flight here
                    lk.unlock();
                    if (!drain()) { fine = false; break; }
                    lk.lock();
                }
                j->cv.wait(lk, there);
                if (j->failed.load()) { fine = false; break; }
                g = group_by_id(j, want);
                if (!g) break;
                j->next_group[(size_t)d] = want + 1;
            }
            if (!process(g)) { fine = false; break; }
        }
        if (fine && ok()) fine = drain();
        if (!fine || !ok()) abandon();
        if (prepped) { if (!prepped_waited) (void)dwgsim_hip_mutate_wait(x, prepped_handle); prepped.reset(); }
        mut_stop();
    }

    bool takes_part(const GroupJob &g) const { return j->vrank(d) == 0 || (j->want_reads && j->vrank(d) < g.nd); }      // device 0 also writes the mutation text

    bool process(const std::shared_ptr<GroupJob> &g)
    {
        if (!takes_part(*g)) {      // a small group is not worth a copy on every device
            std::lock_guard<std::mutex> lk(j->m);
            if (j->solo_rank >= 0) {      // (device 0, which is not here, would publish the group's counts: zero, as all the others' are)
                g->counted = g->nd;
                auto nx = group_by_id(j, g->id + 1);
                if (nx && !nx->base_known && g->base_known) { nx->rand_base = g->rand_base; nx->base_known = true; }
            }
            if (--g->stage_users == 0) { j->stage_busy[g->stage_slot] = false; }
            j->cv.notify_all();
            return true;
        }
        int h; bool waited = false, counted = false;
        if (prepped && prepped->id == g->id) { h = prepped_handle; waited = prepped_waited; counted = prepped_counted; prepped.reset(); }
        else { trace(j, "dev %d group %d: prep", d, g->id); if ((h = prep(g)) < 0) return false; trace(j, "dev %d group %d: upload + walk enqueued", d, g->id); }
        if (!waited && !walked(g, h)) return false;
        trace(j, "dev %d group %d: walked", d, g->id);
        if (mut_thread.joinable()) {      // the group's list of mutated cells goes to the text thread (a few MB; the device part takes well under a millisecond)
            int n = 0;
            dwgsim_hip_mutlist_t *L = dwgsim_hip_mutations_take(x, h, &n);
            if (!L) { fail_ctx(); return false; }
            { std::lock_guard<std::mutex> lk(mm); mq.push_back(MutTask{L, g->names}); }
            mcv.notify_all();
            trace(j, "dev %d group %d: mutation list taken", d, g->id);
        }
        const int nb = (int)g->batches.size();
        std::vector<int> mine;
        if (j->want_reads && j->vrank(d) < g->nd) for (int b = j->vrank(d); b < nb; b += g->nd) mine.push_back(b);
        auto ranges_of = [&](int b) { std::vector<dwgsim_hip_range_t> r = g->batches[(size_t)b]; for (auto &q : r) q.contig += h; return r; };
        if (j->VD > 1 && j->want_reads) {
            if (!counted && !count_mine(g, h)) return false;
            std::unique_lock<std::mutex> lk(j->m);
            auto counts_in = [&]() { return j->failed.load() || (g->counted >= g->nd && g->base_known); };
            if (!counts_in() && !fl.empty()) {      // another device may be waiting for page-locked buffers that only come back once the batches in flight here are published
                lk.unlock();
                if (!drain()) return false;
                lk.lock();
            }
            j->cv.wait(lk, counts_in);
            if (j->failed.load()) return false;
            if (g->counted == g->nd) {      // (every device computes the same thing; the first one publishes it for the next group)
                uint64_t tot = g->rand_base;
                for (uint64_t c : g->batch_rand) tot += c;
                auto nx = group_by_id(j, g->id + 1);
                if (nx && !nx->base_known) { nx->rand_base = tot; nx->base_known = true; j->cv.notify_all(); }
                j->total_rand = tot;
            }
        }
        if (mine.empty()) {      // nothing to simulate here (device 0 of a small group, or -o 2): the group's memory goes back at once
            if (dwgsim_hip_drop_contig(x, h) < 0) { fail_ctx(); return false; }
            return look_ahead(g);
        }
        // the next group, if it is already here, is uploaded and walked on the walk stream while this one's batches run
        if (!look_ahead(g)) return false;
        for (size_t q = 0; q < mine.size(); ++q) {
            const int b = mine[q];
            if (!ok()) return false;
            uint64_t rbase;
            if (j->VD > 1) { rbase = g->rand_base; for (int t = 0; t < b; ++t) rbase += g->batch_rand[(size_t)t]; }      // (batch_rand is final: all devices have published)
            else { rbase = first_batch_of_job ? 0 : DWGSIM_HIP_RAND_CHAIN; first_batch_of_job = false; }
            const auto r = ranges_of(b);
            const int slot = (int)(kk % DWGSIM_HIP_SLOTS);      // (free: at most DWGSIM_HIP_SLOTS - 1 batches are in flight here)
            if (dwgsim_hip_simulate_ranges_async(x, r.data(), (int)r.size(), rbase, slot) < 0) { fail_ctx(); return false; }
            if (q < 3 || q + 1 == mine.size()) trace(j, "dev %d group %d: batch %d enqueued", d, g->id, b);
            ++kk;
            fl.emplace_back();
            Pending &cur = fl.back(); cur.slot = slot; cur.b = b; cur.h = h; cur.g = g; cur.last_of_group = q + 1 == mine.size();
            for (size_t t = 0; t + 1 < fl.size(); ++t) if (!stage_a(fl[t])) return false;
            while ((int)fl.size() >= DWGSIM_HIP_SLOTS) { if (!stage_b(fl.front())) return false; fl.pop_front(); }
            if (!look_ahead(g)) return false;
        }
        return ok();
    }
};

// one output stream: the batches of every group, in order
void deliver_loop(dwgsim_hip_job *j, int s)
{
    int gid = 0;
    for (;;) {
        std::shared_ptr<GroupJob> g;
        {
            std::unique_lock<std::mutex> lk(j->m);
            j->cv.wait(lk, [&]() { return j->failed.load() || gid < j->n_dispatched || j->no_more; });
            if (j->failed.load()) return;
            if (gid >= j->n_dispatched) return;
            g = group_by_id(j, gid);
            if (!g) { ++gid; continue; }      // retired already: it had nothing for this stream
        }
        const int nb = j->want_reads ? (int)g->batches.size() : 0;
        for (int b = 0; b < nb; ++b) {
            BatchOut bo;
            if (j->solo_rank >= 0 && b % g->nd != j->solo_rank) continue;      // (another device's batch: not made here)
            {
                std::unique_lock<std::mutex> lk(j->m);
                j->cv.wait(lk, [&]() { return j->failed.load() || g->joined > b; });      // simulated, and the abort rule's verdict over everything up to it is in
                if (j->failed.load()) return;
                bo = g->out[(size_t)b];
            }
            if (bo.n[s] && j->sink.reads) {
                if (j->sink.reads(j->sink.user, s, bo.buf->p[s], bo.n[s], bo.text_n[s], j->gzip ? 1 : 0) != 0) { job_fail(j, "dwgsim-hip: writing FASTQ failed"); return; }
                std::lock_guard<std::mutex> lk(j->m);
                BatchOut &ref = g->out[(size_t)b];
                if (--ref.left == 0) { j->free_bufs[(size_t)ref.lane].push_back(ref.buf); ref.buf = nullptr; j->cv.notify_all(); }
            }
        }
        ++gid;
    }
}

// reads_at: the batches device d made, stream s, each piece with its offset -- one thread per (device, stream): N devices deliver side by side, where the
// ordered form has ONE thread per stream (21-26 GB/s through a sink that touches the bytes: profiles/r06_solo_rank_entry.txt -- below one device's link)
void deliver_at_loop(dwgsim_hip_job *j, int d, int s)
{
    int gid = 0;
    for (;;) {
        std::shared_ptr<GroupJob> g;
        {
            std::unique_lock<std::mutex> lk(j->m);
            j->cv.wait(lk, [&]() { return j->failed.load() || gid < j->n_dispatched || j->no_more; });
            if (j->failed.load()) return;
            if (gid >= j->n_dispatched) return;
            g = group_by_id(j, gid);
            if (!g) { ++gid; continue; }      // retired already: it had nothing for this thread
        }
        const int nb = j->want_reads ? (int)g->batches.size() : 0;
        for (int b = j->vrank(d); b < nb && j->vrank(d) < g->nd; b += g->nd) {
            BatchOut bo; uint64_t off = 0;
            {
                std::unique_lock<std::mutex> lk(j->m);
                j->cv.wait(lk, [&]() { return j->failed.load() || (g->joined > b && g->off_ok[(size_t)b]); });      // simulated, landed, behind the abort rule's verdict, and placed
                if (j->failed.load()) return;
                bo = g->out[(size_t)b]; off = g->off[(size_t)b][(size_t)s];
            }
            if (bo.n[s]) {
                if (j->sink.reads_at(j->sink.user, s, off, bo.buf->p[s], bo.n[s], bo.text_n[s], j->gzip ? 1 : 0) != 0) { job_fail(j, "dwgsim-hip: writing FASTQ failed"); return; }
                std::lock_guard<std::mutex> lk(j->m);
                BatchOut &ref = g->out[(size_t)b];
                if (--ref.left == 0) { j->free_bufs[(size_t)ref.lane].push_back(ref.buf); ref.buf = nullptr; j->cv.notify_all(); }
            }
        }
        ++gid;
    }
}

// the group is complete when every batch was simulated (finish_batch has joined the abort rule's summaries by then) and delivered: retire it
void retire_loop_step(dwgsim_hip_job *j)      // j->m held
{
    while (!j->groups.empty()) {
        auto &g = j->groups.front();
        const int nb = j->want_reads ? (int)g->batches.size() : 0;
        bool delivered = g->batches_done >= nb && g->joined >= nb;
        for (int b = 0; b < nb && delivered; ++b) if (!g->out[(size_t)b].ready || g->out[(size_t)b].left > 0) delivered = false;
        bool all_taken = true;
        for (int d = 0; d < j->ND; ++d) if (j->next_group[(size_t)d] <= g->id) all_taken = false;
        if (!delivered || !all_taken || g->stage_users > 0) break;
        if (j->VD == 1) for (uint64_t r : g->got_rand) j->total_rand += r;
        j->groups.pop_front();
    }
}

int dispatch_pending(dwgsim_hip_job *j)
{
    if (!j->pending) return DWGSIM_HIP_OK;
    auto g = j->pending; j->pending.reset(); j->pending_bytes = 0;
    if (g->names.empty()) {      // every contig that was begun for it was skipped: only the staging goes back
        std::lock_guard<std::mutex> lk(j->m);
        j->stage_busy[g->stage_slot] = false; j->cv.notify_all();
        return DWGSIM_HIP_OK;
    }
    {   // the sequences in the staging as it stands now (begin_contig may have reallocated it after earlier commits, also for a contig that was then skipped)
        std::vector<int64_t> starts(g->lens.size());
        (void)dwgsim_hip_group_layout(g->lens.data(), (int)g->lens.size(), starts.data());
        g->ptrs.clear();
        for (size_t k = 0; k < g->lens.size(); ++k) g->ptrs.push_back(j->stage[g->stage_slot] + starts[k]);
    }
    // the group's pairs in file order, cut into batches; batch b belongs to device b mod nd
    g->pairs = 0; for (int64_t n : g->n_pairs) g->pairs += (uint64_t)n;
    g->nd = j->VD;
    while (g->nd > 1 && g->pairs / (uint64_t)g->nd < j->min_share) --g->nd;
    if (j->want_reads && g->pairs) {
        // a multiple of nd near-equal batches of at most batch_pairs pairs: every device gets the same number of them, of the same size
        const uint64_t nbt = (uint64_t)g->nd * ((g->pairs + (uint64_t)g->nd * j->batch_pairs - 1) / ((uint64_t)g->nd * j->batch_pairs));
        const uint64_t per = (g->pairs + nbt - 1) / nbt;
        std::vector<dwgsim_hip_range_t> cur; uint64_t room = per, cur_pairs = 0;
        for (size_t k = 0; k < g->n_pairs.size(); ++k) {
            uint64_t first = 0, n = (uint64_t)g->n_pairs[k];
            while (n > 0) {
                const uint64_t take = n < room ? n : room;
                dwgsim_hip_range_t r; memset(&r, 0, sizeof r); r.contig = (int32_t)k; r.first_ii = first; r.n_pairs = take;
                cur.push_back(r); first += take; n -= take; room -= take; cur_pairs += take;
                if (room == 0) { g->batches.push_back(cur); g->batch_pairs.push_back(cur_pairs); cur.clear(); room = per; cur_pairs = 0; }
            }
        }
        if (!cur.empty()) { g->batches.push_back(cur); g->batch_pairs.push_back(cur_pairs); }
    }
    const size_t nb = g->batches.size();
    g->batch_rand.assign(nb, 0); g->fail_seg.assign(nb, std::array<uint64_t, 4>{0, 0, 0, 0}); g->got_rand.assign(nb, 0); g->out.assign(nb, BatchOut());
    g->sz.assign(nb, std::array<uint64_t, 3>{0, 0, 0}); g->off.assign(nb, std::array<uint64_t, 3>{0, 0, 0}); g->sized.assign(nb, 0); g->off_ok.assign(nb, 0);
    if (j->solo_rank >= 0) {      // (the other devices' batches: as if simulated, joined -- their abort-rule summaries are the identity -- and delivered)
        for (size_t b = 0; b < nb; ++b) if ((int)(b % (size_t)g->nd) != j->solo_rank) { g->out[b].ready = true; g->sized[b] = 1; ++g->batches_done; }
        while (g->joined < (int)nb && g->out[(size_t)g->joined].ready) ++g->joined;
    }
    g->stage_users = j->ND;
    std::unique_lock<std::mutex> lk(j->m);
    // at most two groups in front of the devices: the staging of a third one is being filled meanwhile
    j->cv.wait(lk, [&]() { retire_loop_step(j); return j->failed.load() || j->groups.size() < 2; });
    if (j->failed.load()) return DWGSIM_HIP_ERR_FAILED;
    g->id = j->n_dispatched++;
    if (g->id == 0) { g->base_known = true; g->rand_base = 0; }
    else if (auto pv = group_by_id(j, g->id - 1)) { if (pv->counted >= pv->nd && pv->base_known && j->VD > 1) { uint64_t t = pv->rand_base; for (uint64_t c : pv->batch_rand) t += c; g->rand_base = t; g->base_known = true; } }
    else { g->rand_base = j->total_rand; g->base_known = true; }      // the group in front has been retired already: its total is final
    j->groups.push_back(g);
    assign_offsets(j);
    j->cv.notify_all();
    trace(j, "group %d dispatched (%zu contigs, %llu pairs, %zu batches)", g->id, g->names.size(), (unsigned long long)g->pairs, g->batches.size());
    return DWGSIM_HIP_OK;
}

int start_threads(dwgsim_hip_job *j)
{
    if (j->started) return DWGSIM_HIP_OK;
    j->started = true;
    for (int d = 0; d < j->ND; ++d) {
        if (!j->regions_path.empty()) {      // dwgsim.c:499-506
            std::vector<const char *> nm; for (auto &s : j->tab_names) nm.push_back(s.c_str());
            uint64_t tl = 0;
            if (dwgsim_hip_set_regions(j->ctx[(size_t)d], j->regions_path.c_str(), nm.data(), j->tab_lens.data(), (int)nm.size(), &tl) < 0) { job_fail(j, dwgsim_hip_last_error(j->ctx[(size_t)d])); return DWGSIM_HIP_ERR_ARG; }
            j->tot_len = tl;
        }
        if (j->mutin_type >= 0) {            // dwgsim.c:494-497
            std::vector<const char *> nm; for (auto &s : j->tab_names) nm.push_back(s.c_str());
            if (dwgsim_hip_set_mutation_input(j->ctx[(size_t)d], j->mutin_type, j->mutin_path.c_str(), nm.data(), j->tab_lens.data(), (int)nm.size()) < 0) { job_fail(j, dwgsim_hip_last_error(j->ctx[(size_t)d])); return DWGSIM_HIP_ERR_ARG; }
        }
        if (j->gzip && j->want_reads && j->has_sink() && dwgsim_hip_set_gzip(j->ctx[(size_t)d], 1) < 0) { job_fail(j, dwgsim_hip_last_error(j->ctx[(size_t)d])); return DWGSIM_HIP_ERR_DEVICE; }
    }
    for (int d = 0; d < j->ND; ++d) j->workers.emplace_back([j, d]() { Worker w{j, d, j->ctx[(size_t)d]}; w.run(); });
    if (j->want_reads && j->sink.reads_at) { for (int d = 0; d < j->ND; ++d) for (int s = 0; s < 3; ++s) j->deliver_at.emplace_back([j, d, s]() { deliver_at_loop(j, d, s); }); }
    else if (j->want_reads && j->sink.reads) for (int s = 0; s < 3; ++s) j->deliver[s] = std::thread([j, s]() { deliver_loop(j, s); });
    return DWGSIM_HIP_OK;
}

} // namespace

extern "C" {

dwgsim_hip_job_t *dwgsim_hip_job_create(const dwgsim_hip_params_t *p, const int *devices, int n_devices, const dwgsim_hip_job_sink_t *sink,
                                        const dwgsim_hip_job_options_t *opt, int *err)
{
    auto bad = [&](int e) { if (err) *err = e; return (dwgsim_hip_job_t *)nullptr; };
    if (!p) return bad(DWGSIM_HIP_ERR_ARG);
    std::vector<int> devs;
    if (n_devices <= 0 || !devices) { const int n = dwgsim_hip_device_count(); for (int d = 0; d < n; ++d) devs.push_back(d); }      // every device the process sees
    else devs.assign(devices, devices + n_devices);
    if (devs.empty()) { fprintf(stderr, "dwgsim-hip: no usable HIP device; the hot path has no CPU fallback\n"); return bad(DWGSIM_HIP_ERR_DEVICE); }
    int solo_r = -1, solo_w = 0;
    if (const char *e = getenv("DWGSIM_HIP_SOLO")) {      // "r/W": measurement only (struct dwgsim_hip_job)
        if (sscanf(e, "%d/%d", &solo_r, &solo_w) != 2 || solo_w < 1 || solo_r < 0 || solo_r >= solo_w) { fprintf(stderr, "dwgsim-hip: DWGSIM_HIP_SOLO wants r/W with 0 <= r < W\n"); return bad(DWGSIM_HIP_ERR_ARG); }
        devs.resize(1);
    }
    auto *j = new dwgsim_hip_job();
    j->tracing = getenv("DWGSIM_HIP_TRACE") != nullptr; j->t0 = mono_s();
    j->prm = *p;
    if (p->read_prefix) { j->prefix = p->read_prefix; j->prm.read_prefix = j->prefix.c_str(); }
    if (p->flow_order) { j->flow = p->flow_order; j->prm.flow_order = j->flow.c_str(); }
    memset(&j->sink, 0, sizeof j->sink); if (sink) j->sink = *sink;
    memset(&j->opt, 0, sizeof j->opt); if (opt) j->opt = *opt; else j->opt.gzip = 1;
    j->gzip = j->opt.gzip != 0;
    if (j->opt.batch_pairs) j->batch_pairs = j->opt.batch_pairs;
    if (j->opt.group_bp) j->group_bp = j->opt.group_bp;
    if (j->opt.min_share) j->min_share = j->opt.min_share;
    j->want_mut = p->output_type != 1; j->want_reads = p->output_type != 2;
    j->devices = devs; j->ND = (int)devs.size();
    j->VD = solo_r >= 0 ? solo_w : j->ND; j->solo_rank = solo_r;
    j->ctx.assign((size_t)j->ND, nullptr);
    {   // one context per device, made side by side (a context costs about 0.1 s of runtime set-up, code objects and buffers)
        std::vector<int> errs((size_t)j->ND, 0);
        std::vector<std::thread> th;
        dwgsim_hip_params_t rest = j->prm;
        const bool calibrates = j->prm.data_type == 2 && j->prm.use_base_error;      // -B (dwgsim_opt.c:415-457): once, on the first device; the others take its result
        if (calibrates) {
            j->ctx[0] = dwgsim_hip_create(&j->prm, devs[0], &errs[0]);
            if (j->ctx[0] && dwgsim_hip_get_params(j->ctx[0], &rest) == DWGSIM_HIP_OK) { rest.use_base_error = 0; rest.read_prefix = j->prm.read_prefix; rest.flow_order = j->prm.flow_order; }
        }
        for (int d = 1; d < j->ND; ++d) th.emplace_back([&, d]() { j->ctx[(size_t)d] = dwgsim_hip_create(&rest, devs[(size_t)d], &errs[(size_t)d]); });
        if (!calibrates) j->ctx[0] = dwgsim_hip_create(&j->prm, devs[0], &errs[0]);
        for (auto &t : th) t.join();
        for (int d = 0; d < j->ND; ++d)
            if (!j->ctx[(size_t)d]) {
                const int e = errs[(size_t)d];
                fprintf(stderr, "dwgsim-hip: cannot create a GPU context on device %d (error %d)\n", devs[(size_t)d], e);
                for (auto *x : j->ctx) if (x) dwgsim_hip_destroy(x);
                delete j;
                return bad(e ? e : DWGSIM_HIP_ERR_DEVICE);
            }
    }
    j->next_group.assign((size_t)j->ND, 0);
    j->bufs.resize((size_t)j->ND); j->free_bufs.resize((size_t)j->ND);
    trace(j, "contexts made");
    if (err) *err = DWGSIM_HIP_OK;
    return j;
}

int dwgsim_hip_job_set_contig_table(dwgsim_hip_job_t *j, const char *const *names, const int64_t *lens, int n)
{
    if (!j || n < 0 || (n && (!names || !lens)) || j->started) return arg_error(j, DWGSIM_HIP_ERR_ARG, "job: the contig table must be set once, before the first contig");
    j->tab_names.clear(); j->tab_lens.clear(); j->tot_len = 0;
    for (int i = 0; i < n; ++i) { j->tab_names.push_back(names[i]); j->tab_lens.push_back(lens[i]); j->tot_len += (uint64_t)lens[i]; }
    j->n_ref = n; j->have_table = true;
    // room for the largest group (consecutive contigs up to group_bp, or one longer contig alone), so that each staging buffer is page-locked once
    int64_t longest = 0; for (int i = 0; i < n; ++i) longest = std::max<int64_t>(longest, (lens[i] + 4095) / 4096 * 4096);
    j->stage_want = (size_t)std::max<int64_t>(longest, (int64_t)std::min<uint64_t>(j->group_bp, (uint64_t)j->tot_len + 4096 * (uint64_t)n)) + 8192;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_job_set_regions(dwgsim_hip_job_t *j, const char *path)
{
    if (!j || !path || j->started) return arg_error(j, DWGSIM_HIP_ERR_ARG, "job: regions must be set before the first contig");
    j->regions_path = path;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_job_set_mutation_input(dwgsim_hip_job_t *j, int type, const char *path)
{
    if (!j || !path || type < 0 || type > 2 || j->started) return arg_error(j, DWGSIM_HIP_ERR_ARG, "job: the mutation input must be set before the first contig");
    j->mutin_type = type; j->mutin_path = path;
    return DWGSIM_HIP_OK;
}

int dwgsim_hip_job_prepare(dwgsim_hip_job_t *j, uint64_t *total_len)
{
    if (!j || !j->have_table) return arg_error(j, DWGSIM_HIP_ERR_STATE, "job: set the contig table first");
    const int rc = start_threads(j);
    if (total_len) *total_len = j->tot_len;
    return rc;
}

// The contig loop's body (dwgsim.c:519-625) in two halves, so that the caller can produce the sequence in place -- and with as many threads as it
// likes: begin reserves the contig's bytes in the page-locked staging of the group being filled (group layout), commit schedules it.
uint8_t *dwgsim_hip_job_begin_contig(dwgsim_hip_job_t *j, const char *name, int64_t l, int64_t *status)
{
    auto out = [&](int64_t st) { if (status) *status = st; return (uint8_t *)nullptr; };
    if (!j || !name || l < 0 || l > INT32_MAX) return out(arg_error(j, DWGSIM_HIP_ERR_ARG, "job: bad contig arguments"));
    if (!j->have_table) return out(arg_error(j, DWGSIM_HIP_ERR_STATE, "job: set the contig table first (the reference reads it before the first contig: dwgsim.c:465-478)"));
    if (j->finished) return out(arg_error(j, DWGSIM_HIP_ERR_STATE, "job: already finished"));
    if (j->open.open) return out(arg_error(j, DWGSIM_HIP_ERR_STATE, "job: the previous contig was neither committed nor cancelled"));
    if (start_threads(j) < 0) return out(DWGSIM_HIP_ERR_FAILED);
    if (j->failed.load()) return out(DWGSIM_HIP_ERR_FAILED);
    // into the group being filled; a contig that would take it past the group size closes it first
    const int64_t aligned_len = (l + 4095) / 4096 * 4096;
    if (j->pending && j->pending_bytes + (size_t)aligned_len > (size_t)j->group_bp) { if (dispatch_pending(j) < 0) return out(DWGSIM_HIP_ERR_FAILED); }
    if (!j->pending) {
        auto g = std::make_shared<GroupJob>();
        std::unique_lock<std::mutex> lk(j->m);
        j->cv.wait(lk, [&]() { retire_loop_step(j); if (j->failed.load()) return true; for (int s = 0; s < dwgsim_hip_job::N_STAGE; ++s) if (!j->stage_busy[s]) return true; return false; });
        if (j->failed.load()) return out(DWGSIM_HIP_ERR_FAILED);
        for (int s = 0; s < dwgsim_hip_job::N_STAGE; ++s) if (!j->stage_busy[s]) { g->stage_slot = s; j->stage_busy[s] = true; break; }
        j->pending = g; j->pending_bytes = 0;
    }
    GroupJob &g = *j->pending;
    // the contig's place in the group layout (dwgsim_hip_group_layout): the next multiple of 4096
    std::vector<int64_t> lens = g.lens; lens.push_back(l);
    std::vector<int64_t> starts(lens.size());
    const int64_t total = dwgsim_hip_group_layout(lens.data(), (int)lens.size(), starts.data());
    const int s = g.stage_slot;
    if (!j->stage[s] || (size_t)total > j->stage_cap[s]) {      // grow, keeping what the group already holds (an empty record that opens a group on a fresh slot still needs somewhere to point)
        // (the contig table says how large a group can get: exactly that much; a table that understated the lengths -- a stale .fai -- grows by a quarter)
        const size_t want = (size_t)total <= j->stage_want ? j->stage_want : std::max<size_t>((size_t)total + (size_t)total / 4, (size_t)std::min<uint64_t>(j->group_bp, 256u << 20) + 8192);
        trace(j, "staging %d: page-locking %.0f MB", s, want / 1e6);
        uint8_t *nb = (uint8_t *)dwgsim_hip_host_alloc(want);
        trace(j, "staging %d: done", s);
        if (!nb) { job_fail(j, "dwgsim-hip: cannot allocate page-locked host memory for the sequence"); return out(DWGSIM_HIP_ERR_NOMEM); }
        if (j->stage[s] && j->pending_bytes) memcpy(nb, j->stage[s], j->pending_bytes);
        dwgsim_hip_host_free(j->stage[s]);
        j->stage[s] = nb; j->stage_cap[s] = want;
    }
    const int64_t st = starts.back();
    if ((size_t)st > j->pending_bytes) memset(j->stage[s] + j->pending_bytes, 0, (size_t)st - j->pending_bytes);      // zero bytes between the contigs
    if ((size_t)total > (size_t)(st + l)) memset(j->stage[s] + st + l, 0, (size_t)total - (size_t)(st + l));
    j->open.open = true; j->open.name = name; j->open.l = l; j->open.st = st; j->open.total = total; j->open.ci = j->next_index;
    if (status) *status = DWGSIM_HIP_OK;
    return j->stage[s] + st;
}

int dwgsim_hip_job_cancel_contig(dwgsim_hip_job_t *j)
{
    if (!j || !j->open.open) return arg_error(j, DWGSIM_HIP_ERR_STATE, "job: no contig is open");
    j->open.open = false;      // (the reserved bytes are simply handed out again)
    return DWGSIM_HIP_OK;
}

int64_t dwgsim_hip_job_commit_contig(dwgsim_hip_job_t *j)
{
    if (!j || !j->open.open) return arg_error(j, DWGSIM_HIP_ERR_STATE, "job: no contig is open");
    j->open.open = false;
    if (j->failed.load()) return DWGSIM_HIP_ERR_FAILED;
    GroupJob &g = *j->pending;
    const int s = g.stage_slot;
    const char *name = j->open.name.c_str(); const int64_t l = j->open.l;
    const uint8_t *ascii = j->stage[s] + j->open.st;
    const dwgsim_hip_params_t &o = j->prm;
    const uint32_t ci = j->next_index++;
    --j->n_ref;
    int64_t n_pairs = 0, l_eff = l;
    if (j->want_reads) {      // dwgsim.c:535-625
        const bool last_takes_rest = j->n_ref == 0 && o.C < 0;     // dwgsim.c:535-537
        if (!j->regions_path.empty() && !last_takes_rest) {
            int64_t num_n = 0, m = 0;
            l_eff = dwgsim_hip_contig_region_length(j->ctx[0], ci, ascii, l, &num_n, &m);
            if (l_eff == DWGSIM_HIP_SKIP_NO_REGION) { say(j, "[dwgsim_core] #0 skip sequence '%s' as it is not in the targeted region\n", name); return l_eff; }
            if (l_eff == DWGSIM_HIP_SKIP_NON_ACGT) { say(j, "[dwgsim_core] #1 skip sequence '%s' as %d out of %d bases are non-ACGT\n", name, (int)num_n, (int)m); return l_eff; }      // dwgsim.c:575
        }
        n_pairs = dwgsim_hip_pairs_for_contig(&o, l_eff, j->tot_len, j->n_ref == 0, j->n_sim);
        if (n_pairs < 0) {
            if (!j->prev_skip) say(j, "\n");
            j->prev_skip = 1;
            // (the reference prints its `l`, which is the region length once -x is in force: dwgsim.c:552, :601, :615)
            if (n_pairs == DWGSIM_HIP_SKIP_AMPLICON) say(j, "[dwgsim_core] #2 skip sequence '%s' as it is shorter than the read length %d < %d!\n", name, (int)l_eff, o.length[0] > o.length[1] ? o.length[0] : o.length[1]);
            else if (n_pairs == DWGSIM_HIP_SKIP_SHORT_INSERT) say(j, "[dwgsim_core] #3 skip sequence '%s' as it is shorter than %f!\n", name, o.dist + 3 * o.std_dev);
            else if (n_pairs == DWGSIM_HIP_SKIP_SHORT_READ) say(j, "[dwgsim_core] #4 skip sequence '%s' as it is shorter than %d!\n", name, (l_eff < o.length[0]) ? o.length[0] : o.length[1]);
            else say(j, "[dwgsim_core] #5 skip sequence '%s' as not enough pairs found\n", name);
            return n_pairs;
        }
        j->prev_skip = 0;
        j->n_sim += n_pairs;
    }
    j->pending_bytes = (size_t)j->open.total;
    g.names.push_back(name); g.lens.push_back(l); g.l_eff.push_back(l_eff); g.n_pairs.push_back(n_pairs); g.cindex.push_back(ci);      // (where the sequences stand is resolved at dispatch: the staging may still move)
    if (j->pending_bytes >= (size_t)j->group_bp) { if (dispatch_pending(j) < 0) return DWGSIM_HIP_ERR_FAILED; }
    return n_pairs;
}

int64_t dwgsim_hip_job_add_contig(dwgsim_hip_job_t *j, const char *name, const uint8_t *ascii, int64_t l)
{
    if (j && !ascii && l > 0) return arg_error(j, DWGSIM_HIP_ERR_ARG, "job: bad contig arguments");
    int64_t st = 0;
    uint8_t *dst = dwgsim_hip_job_begin_contig(j, name, l, &st);
    if (st < 0 || !dst) return st < 0 ? st : arg_error(j, DWGSIM_HIP_ERR_STATE, "job: begin_contig returned no place for the sequence");      // (the status decides, not the pointer)
    if (l > 0) memcpy(dst, ascii, (size_t)l);
    return dwgsim_hip_job_commit_contig(j);
}

int dwgsim_hip_job_finish(dwgsim_hip_job_t *j)
{
    if (!j) return DWGSIM_HIP_ERR_ARG;
    if (j->finished) return j->failed.load() ? DWGSIM_HIP_ERR_FAILED : DWGSIM_HIP_OK;
    j->finished = true;
    if (!j->failed.load()) (void)dispatch_pending(j);
    { std::lock_guard<std::mutex> lk(j->m); j->no_more = true; j->cv.notify_all(); }
    for (auto &t : j->workers) if (t.joinable()) t.join();
    for (auto &t : j->deliver) if (t.joinable()) t.join();
    for (auto &t : j->deliver_at) if (t.joinable()) t.join();
    { std::lock_guard<std::mutex> lk(j->m); if (!j->failed.load()) retire_loop_step(j); }
    return j->failed.load() ? DWGSIM_HIP_ERR_FAILED : DWGSIM_HIP_OK;
}

const char *dwgsim_hip_job_last_error(const dwgsim_hip_job_t *j) { return j ? j->err.c_str() : "no job"; }

void dwgsim_hip_job_destroy(dwgsim_hip_job_t *j)
{
    if (!j) return;
    if (!j->finished) { job_fail(j, "job destroyed before it was finished"); (void)dwgsim_hip_job_finish(j); }
    for (auto *x : j->ctx) if (x) dwgsim_hip_destroy(x);
    for (int s = 0; s < dwgsim_hip_job::N_STAGE; ++s) dwgsim_hip_host_free(j->stage[s]);
    for (auto &lane : j->bufs) for (auto &b : lane) for (int s = 0; s < 3; ++s) dwgsim_hip_host_free(b->p[s]);
    delete j;
}

} // extern "C"
