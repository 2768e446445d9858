// dwgsim_cli.cpp -- `dwgsim-hip`: the dwgsim command line (reference src/dwgsim.c:1123-1184 main(),
// src/dwgsim_opt.c:204-472 option surface) driving the MI355X hot path through the C-ABI of
// include/dwgsim_hip.h.  Same usage, same defaults, same output file names:
//     dwgsim-hip [options] <in.ref.fa> <out.prefix>
//     <prefix>.mutations.txt / .mutations.vcf / .bfast.fastq.gz / .bwa.read1.fastq.gz / .bwa.read2.fastq.gz
// Host-only work here: option parsing, FASTA reading (mut.c:49-87), contig scheduling, file I/O and
// gzip (multi-member gzip: the reference's own test compares decompressed bytes, testdata/test.sh:23-25).
//
// Three overlapped stages replace the reference's gzprintf / gzputc stream (dwgsim.c:919-981):
//     GPU kernels (batch k)  |  device -> pinned host copy (batch k-1)  |  deflate + ordered write (batches <= k-2)
// and one or more GPUs work on disjoint read-index ranges of every contig (host threads, one context per device; the
// only cross-device quantities are two integers per range: the random-read count that offsets rand_ii, dwgsim.c:1042,1096,
// and the abort rule's failure counter, dwgsim.c:635).  Environment (all optional):
//     DWGSIM_HIP_DEVICES   "0,1,2,3" or a count "4" (default: device DWGSIM_HIP_DEVICE or 0)
//     DWGSIM_HIP_THREADS   deflate threads (default: all host cores)
//     DWGSIM_HIP_GZIP      "gpu" (default): the gzip members are made on the GPU (dwgsim_hip_set_gzip: Huffman-coded 32 KiB members, ~0.49 of the
//                          text), the host only writes them; "cpu": zlib on all host cores at DWGSIM_HIP_GZIP_LEVEL (smaller files, deflate-bound)
//     DWGSIM_HIP_GZIP_LEVEL  zlib level 0..9 for DWGSIM_HIP_GZIP=cpu (default 1: the text is produced ~1000x faster than zlib -6 packs it)
//     DWGSIM_HIP_BATCH     read pairs per GPU batch (default 2^20)
//     DWGSIM_HIP_MIN_SHARE a contig is spread over fewer devices while a device's share would be below this many pairs (default 65536)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <time.h>
#include <unistd.h>
#include <sched.h>
#include <limits.h>
#include <zlib.h>
#include <string>
#include <vector>
#include <deque>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <memory>
#include <functional>
#include "../../include/dwgsim_hip.h"

#define PACKAGE_VERSION "0.1.17-hip"

static int usage(const dwgsim_hip_params_t *o)
{
    fprintf(stderr, "\nProgram: dwgsim-hip (short read simulator, MI355X hot path)\nVersion: %s\n\n", PACKAGE_VERSION);
    fprintf(stderr, "Usage:   dwgsim-hip [options] <in.ref.fa> <out.prefix>\n\nOptions:\n");
    fprintf(stderr, "         -e FLOAT      per base/color/flow error rate of the first read [from %.3f to %.3f]\n", o->e_start[0], o->e_end[0]);
    fprintf(stderr, "         -E FLOAT      per base/color/flow error rate of the second read [from %.3f to %.3f]\n", o->e_start[1], o->e_end[1]);
    fprintf(stderr, "         -i            use the inner distance instead of the outer distance for pairs\n");
    fprintf(stderr, "         -d INT        outer/inner distance between the two ends for pairs [%d]\n", o->dist);
    fprintf(stderr, "         -s INT        standard deviation of the distance for pairs [%.3f]\n", o->std_dev);
    fprintf(stderr, "         -N INT        number of read pairs (-1 to disable) [%lld]\n", (long long)o->N);
    fprintf(stderr, "         -C FLOAT      mean coverage across available positions (-1 to disable) [%.2lf]\n", o->C);
    fprintf(stderr, "         -1 INT        length of the first read [%d]\n         -2 INT        length of the second read [%d]\n", o->length[0], o->length[1]);
    fprintf(stderr, "         -r FLOAT      rate of mutations [%.4f]\n         -F FLOAT      frequency of given mutation [%.4f]\n", o->mut_rate, o->mut_freq);
    fprintf(stderr, "         -R FLOAT      fraction of mutations that are indels [%.2f]\n         -X FLOAT      probability an indel is extended [%.2f]\n", o->indel_frac, o->indel_extend);
    fprintf(stderr, "         -I INT        the minimum length indel [%d]\n         -y FLOAT      probability of a random DNA read [%.2f]\n", o->indel_min, o->rand_read);
    fprintf(stderr, "         -n INT        maximum number of Ns allowed in a given read [%d]\n", o->max_n);
    fprintf(stderr, "         -c INT        generate reads for 0: Illumina, 1: SOLiD, 2: Ion Torrent [%d]\n", o->data_type);
    fprintf(stderr, "         -S INT        paired end orientation 0: default, 1: same strand, 2: opposite strand [%d]\n", o->strandedness);
    fprintf(stderr, "         -A INT        read one strand 0: random, 1: forward, 2: reverse [%d]\n", o->read_one_strand);
    fprintf(stderr, "         -H            haploid mode\n         -z INT        random seed (-1 uses the current time) [%d]\n", o->seed);
    fprintf(stderr, "         -M INT        output files 0: reads and mutations, 1: reads only, 2: mutations only [%d]\n", o->output_type);
    fprintf(stderr, "         -P STRING     a read prefix to prepend to each read name\n         -q STRING     a fixed base quality to apply (single character)\n");
    fprintf(stderr, "         -Q FLOAT      standard deviation of the base quality scores [%.2lf]\n", o->quality_std);
    fprintf(stderr, "         -o INT        FASTQ output 0: bfast and bwa, 1: bwa only, 2: bfast only [%d]\n", o->reads_output_type);
    fprintf(stderr, "         -a            assume each contig is an amplicon\n         -h            print this message\n\n");
    fprintf(stderr, "         -f STRING     the flow order for Ion Torrent data\n");
    fprintf(stderr, "         -m FILE       the mutations txt file to re-create\n         -b FILE       the bed-like file set of candidate mutations\n         -v FILE       the vcf file set of candidate mutations (use pl tag for strand)\n");
    fprintf(stderr, "         -x FILE       the bed of regions to cover\n");
    return 1;
}

static void get_error_rate(const char *str, double *start, double *end)   // dwgsim_opt.c:162-179
{
    size_t i, n = strlen(str);
    *start = atof(str);
    for (i = 0; i < n; ++i) if (str[i] == ',' || str[i] == '-') break;
    if (n > 0 && i < n - 1) *end = atof(str + i + 1); else *end = *start;
}
static int xatoi(const char *a, char flag, int neg_ok)                     // dwgsim_opt.c:181-202
{
    size_t n = strlen(a); int ok = n > 0;
    if (ok && '+' != a[0] && (neg_ok == 0 || '-' != a[0]) && !isdigit((unsigned char)a[0])) ok = 0;
    for (size_t i = 1; ok && i < n; ++i) if (!isdigit((unsigned char)a[i])) ok = 0;
    if (!ok) { fprintf(stderr, "Error: command line option -%c is not a number [%s]\n", flag, a); exit(1); }
    return atoi(a);
}

struct Fasta { std::vector<std::string> names; std::vector<std::vector<uint8_t>> seqs; };

// mut.c:49-87 seq_read_fasta: the name is the header up to the first blank, the sequence keeps isalpha, '-' and '.'; a '>' opens a new
// record wherever it stands.  Whole lines of plain letters (the usual case) are appended with one copy.
static bool read_fasta(const char *fn, Fasta &fa)
{
    FILE *fp = strcmp(fn, "-") ? fopen(fn, "r") : stdin;
    if (!fp) { fprintf(stderr, "[dwgsim_core] fail to open file '%s'. Abort!\n", fn); return false; }
    std::vector<char> buf((size_t)1 << 24);
    std::string name; std::vector<uint8_t> seq; int state = 0;   // 0 before first '>', 1 in name, 2 rest of header line, 3 sequence
    bool have = false;
    static bool keep[256], init = false;
    if (!init) { for (int c = 0; c < 256; ++c) keep[c] = isalpha(c) || c == '-' || c == '.'; init = true; }
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), fp)) > 0) {
        size_t i = 0;
        while (i < n) {
            const int c = (unsigned char)buf[i];
            if (state == 3) {
                // the run of sequence characters from here
                size_t j = i;
                while (j < n && keep[(unsigned char)buf[j]]) ++j;
                if (j > i) { seq.insert(seq.end(), (const uint8_t *)buf.data() + i, (const uint8_t *)buf.data() + j); i = j; continue; }
                if (c == '>') { fa.names.push_back(name); fa.seqs.emplace_back(std::move(seq)); seq = std::vector<uint8_t>(); state = 1; name.clear(); }
                ++i;
                continue;
            }
            if (state == 0) { if (c == '>') { state = 1; name.clear(); seq.clear(); have = true; } }
            else if (state == 1) { if (c == ' ' || c == '\t') state = 2; else if (c == '\n') state = 3; else if (c != '\r') name.push_back((char)c); }
            else if (state == 2) { if (c == '\n') state = 3; }
            ++i;
        }
    }
    if (have) { fa.names.push_back(name); fa.seqs.emplace_back(std::move(seq)); }
    if (fp != stdin) fclose(fp);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Output stage: text blocks -> independent gzip members (all host cores) -> files, in order.
// A multi-member .gz decompresses to exactly the concatenated text (the reference's own test compares decompressed bytes,
// testdata/test.sh:23-25); the reference itself feeds zlib one byte at a time (dwgsim.c:930-931), ~80 % of its wall time.
// ------------------------------------------------------------------------------------------------
static bool deflate_member(const char *src, size_t n, int level, std::vector<unsigned char> &out)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&zs, (uLong)n) + 64);
    zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n; zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return rc == Z_STREAM_END;
}

// One pinned host buffer set: the text of one GPU batch (up to three streams).  `left` counts the deflate chunks still reading it.
struct TextBuf {
    bool raw = false; size_t text_n[3] = {0, 0, 0};      // raw: the buffers hold finished gzip members (GPU gzip) of text_n[s] bytes of text
    char *p[3] = {nullptr, nullptr, nullptr}; size_t cap[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    std::atomic<int> left{0};
};

struct Chunk {             // one gzip member in flight
    int stream = 0; const char *src = nullptr; size_t n = 0; TextBuf *owner = nullptr;
    std::vector<unsigned char> gz; bool done = false, ok = true, raw = false; size_t text_n = 0;
};

class Output {
public:
    // lanes: ordered producers (one per device); the writer drains lane 0 of a contig completely, then lane 1, ...
    Output(FILE *f0, FILE *f1, FILE *f2, int n_lanes, unsigned n_threads, int level, int bufs_per_lane)
        : level_(level), lanes_((size_t)n_lanes)
    {
        f_[0] = f0; f_[1] = f1; f_[2] = f2;
        for (int l = 0; l < n_lanes; ++l)
            for (int b = 0; b < bufs_per_lane; ++b) {
                bufs_.push_back(std::make_unique<TextBuf>());
                lanes_[(size_t)l].free_bufs.push_back(bufs_.back().get());
            }
        for (unsigned t = 0; t < (n_threads ? n_threads : 1); ++t) workers_.emplace_back([this]() { work(); });
        for (int s = 0; s < 3; ++s) if (f_[s]) file_thr_[s] = std::thread([this, s]() { file_loop(s); });
        writer_ = std::thread([this]() { write_loop(); });
    }
    ~Output() { finish(); for (auto &b : bufs_) for (int s = 0; s < 3; ++s) dwgsim_hip_host_free(b->p[s]); }
    bool failed() const { return failed_; }
    // a free buffer set of this lane with room for need[s] bytes per stream (blocks while all of them are still being deflated:
    // back-pressure on the GPU stage).  Page-locked memory is sized by what the batches really produce, so small jobs pin little.
    TextBuf *acquire(int lane, const uint64_t need[3])
    {
        TextBuf *b = nullptr;
        {
            std::unique_lock<std::mutex> lk(m_);
            Lane &L = lanes_[(size_t)lane];
            cv_buf_.wait(lk, [&]() { return !L.free_bufs.empty() || failed_; });
            if (L.free_bufs.empty()) return nullptr;
            b = L.free_bufs.back(); L.free_bufs.pop_back();
        }
        for (int s = 0; s < 3; ++s) if (need[s] > b->cap[s]) {
            dwgsim_hip_host_free(b->p[s]);
            b->cap[s] = (size_t)need[s] + (size_t)need[s] / 8 + 4096;
            b->p[s] = (char *)dwgsim_hip_host_alloc(b->cap[s]);
            if (!b->p[s]) { fprintf(stderr, "dwgsim-hip: cannot allocate %zu bytes of page-locked host memory\n", b->cap[s]); b->cap[s] = 0; failed_ = true; return nullptr; }
        }
        return b;
    }
    // the text in `b` (b->n[s] bytes per stream) is the next output of this lane
    void submit(int lane, TextBuf *b)
    {
        const size_t CH = (size_t)1 << 20;
        std::vector<std::shared_ptr<Chunk>> cs;
        for (int s = 0; s < 3; ++s)
            for (size_t off = 0; off < b->n[s]; off += b->raw ? b->n[s] : CH) {
                if (b->raw) {          // already gzip members: written as they are, the buffer is released by the writer
                    auto c = std::make_shared<Chunk>();
                    c->stream = s; c->src = b->p[s]; c->n = b->n[s]; c->owner = b; c->raw = true; c->done = true; c->text_n = b->text_n[s];
                    cs.push_back(std::move(c));
                    continue;
                }
                auto c = std::make_shared<Chunk>();
                c->stream = s; c->src = b->p[s] + off; c->n = b->n[s] - off < CH ? b->n[s] - off : CH; c->owner = b;
                cs.push_back(std::move(c));
            }
        std::unique_lock<std::mutex> lk(m_);
        if (cs.empty()) { lanes_[(size_t)lane].free_bufs.push_back(b); cv_buf_.notify_all(); return; }
        b->left.store((int)cs.size());
        b_lane_[b] = lane;
        for (auto &c : cs) { lanes_[(size_t)lane].q.push_back(c); if (!c->raw) todo_.push_back(c); }
        cv_work_.notify_all(); cv_write_.notify_all();
    }
    // this lane has nothing more for the current contig: an end mark IN the lane's queue (the lane may already be filling in the next
    // contig's text behind it while the writer is still busy with other lanes)
    void end_lane(int lane) { std::unique_lock<std::mutex> lk(m_); lanes_[(size_t)lane].q.push_back(nullptr); cv_write_.notify_all(); }
    void finish()
    {
        { std::unique_lock<std::mutex> lk(m_); if (stop_) return; stop_ = true; cv_work_.notify_all(); cv_write_.notify_all(); }
        writer_.join();
        { std::unique_lock<std::mutex> lk(fm_); files_stop_ = true; cv_file_.notify_all(); }
        for (int s = 0; s < 3; ++s) if (file_thr_[s].joinable()) file_thr_[s].join();
        for (auto &w : workers_) w.join();
    }
    uint64_t bytes_in() const { return bytes_in_; }
    uint64_t bytes_out() const { return bytes_out_; }

private:
    struct Lane { std::deque<std::shared_ptr<Chunk>> q; std::vector<TextBuf *> free_bufs; };       // q: chunks in order, nullptr = end of the lane's share of a contig
    void work()
    {
        for (;;) {
            std::shared_ptr<Chunk> c;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&]() { return !todo_.empty() || stop_; });
                if (todo_.empty()) return;
                c = todo_.front(); todo_.pop_front();
            }
            const bool ok = deflate_member(c->src, c->n, level_, c->gz);
            std::unique_lock<std::mutex> lk(m_);
            c->ok = ok; c->done = true;
            if (c->owner->left.fetch_sub(1) == 1) { lanes_[(size_t)b_lane_[c->owner]].free_bufs.push_back(c->owner); cv_buf_.notify_all(); }
            cv_write_.notify_all();
        }
    }
    void write_loop()
    {
        size_t lane = 0;                           // lanes are drained in order, contig after contig
        for (;;) {
            std::shared_ptr<Chunk> c;
            {
                std::unique_lock<std::mutex> lk(m_);
                for (;;) {
                    Lane &L = lanes_[lane];
                    if (!L.q.empty()) {
                        if (!L.q.front()) { L.q.pop_front(); lane = (lane + 1) % lanes_.size(); continue; }      // end mark: on to the next lane
                        if (L.q.front()->done) { c = L.q.front(); L.q.pop_front(); break; }
                    } else if (stop_) {            // shutting down (possibly after an error that left a lane without its end mark): drain what there is
                        bool any = false; for (auto &x : lanes_) if (!x.q.empty()) any = true;
                        if (!any) return;
                        lane = (lane + 1) % lanes_.size(); continue;
                    }
                    cv_write_.wait(lk);
                }
            }
            // the chunk is next in its file: hand it to that file's writer (the three files are written side by side)
            std::unique_lock<std::mutex> lk(fm_);
            fq_[c->stream].push_back(c);
            cv_file_.notify_all();
        }
    }
    void file_loop(int s)
    {
        for (;;) {
            std::shared_ptr<Chunk> c;
            {
                std::unique_lock<std::mutex> lk(fm_);
                cv_file_.wait(lk, [&]() { return !fq_[s].empty() || files_stop_; });
                if (fq_[s].empty()) return;
                c = fq_[s].front(); fq_[s].pop_front();
            }
            const void *data = c->raw ? (const void *)c->src : (const void *)c->gz.data(); const size_t nb = c->raw ? c->n : c->gz.size();
            if (!c->ok || fwrite(data, 1, nb, f_[s]) != nb) { std::unique_lock<std::mutex> lk(m_); failed_ = true; cv_buf_.notify_all(); }
            bytes_in_ += c->raw ? c->text_n : c->n; bytes_out_ += nb;
            if (c->raw) {
                std::unique_lock<std::mutex> lk(m_);
                if (c->owner->left.fetch_sub(1) == 1) { lanes_[(size_t)b_lane_[c->owner]].free_bufs.push_back(c->owner); cv_buf_.notify_all(); }
            } else c->gz = std::vector<unsigned char>();
        }
    }
    FILE *f_[3]; int level_;
    std::mutex m_; std::condition_variable cv_work_, cv_write_, cv_buf_;
    std::vector<Lane> lanes_; std::deque<std::shared_ptr<Chunk>> todo_;
    std::vector<std::unique_ptr<TextBuf>> bufs_;
    std::vector<std::thread> workers_; std::thread writer_, file_thr_[3];
    std::mutex fm_; std::condition_variable cv_file_; std::deque<std::shared_ptr<Chunk>> fq_[3]; bool files_stop_ = false;
    struct PtrMap {              // TextBuf -> lane (a handful of entries)
        std::vector<std::pair<TextBuf *, int>> v;
        int &operator[](TextBuf *b) { for (auto &e : v) if (e.first == b) return e.second; v.emplace_back(b, 0); return v.back().second; }
    } b_lane_;
    bool stop_ = false; std::atomic<bool> failed_{false};
    std::atomic<uint64_t> bytes_in_{0}, bytes_out_{0};
};

static void close_gz(FILE *f, int level)
{
    if (!f) return;
    if (ftell(f) == 0) { std::vector<unsigned char> e; deflate_member("", 0, level, e); fwrite(e.data(), 1, e.size(), f); }   // empty stream: still a valid .gz
    fclose(f);
}

// run fn(d) for d = 0 .. n-1 on n host threads (one per device) and wait
static void on_all(int n, const std::function<void(int)> &fn)
{
    if (n == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int d = 0; d < n; ++d) th.emplace_back(fn, d);
    for (auto &t : th) t.join();
}

// host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (containers)
static unsigned usable_cores()
{
    unsigned n = std::thread::hardware_concurrency(); if (n == 0) n = 4;
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (unsigned)c < n) n = (unsigned)c; }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long long p = 0;
        if (fscanf(f, "%63s %lld", q, &p) == 2 && strcmp(q, "max") && p > 0) { const unsigned lim = (unsigned)((atof(q) / (double)p) + 0.5); if (lim >= 1 && lim < n) n = lim; }
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long long q = 0, p = 0; if (fscanf(g, "%lld", &q) != 1) q = 0; fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &p) != 1) p = 0; fclose(h); }
        if (q > 0 && p > 0) { const unsigned lim = (unsigned)((double)q / (double)p + 0.5); if (lim >= 1 && lim < n) n = lim; }
    }
    return n;
}

static double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

int main(int argc, char **argv)
{
    const double t_start = now_s();
    const bool timing = getenv("DWGSIM_HIP_TIMING") != nullptr;      // stage times on stderr
    dwgsim_hip_params_t o; dwgsim_hip_params_default(&o);
    std::string prefix_s, fixedq_s, flow_s, regions_fn, muts_fn; int muts_type = -1, muts_flags = 0;
    int c;
    while ((c = getopt(argc, argv, "id:s:N:C:1:2:e:E:r:F:R:X:I:c:S:A:n:y:BHf:z:M:m:b:v:x:P:q:Q:o:ah")) >= 0) {
        switch (c) {
        case 'i': o.is_inner = 1; break;
        case 'd': o.dist = xatoi(optarg, 'd', 0); break;
        case 's': o.std_dev = atof(optarg); break;
        case 'N': o.N = xatoi(optarg, 'N', 1); o.C = -1; break;
        case 'C': o.C = atof(optarg); o.N = -1; break;
        case '1': o.length[0] = xatoi(optarg, '1', 0); break;
        case '2': o.length[1] = xatoi(optarg, '2', 0); break;
        case 'e': get_error_rate(optarg, &o.e_start[0], &o.e_end[0]); break;
        case 'E': get_error_rate(optarg, &o.e_start[1], &o.e_end[1]); break;
        case 'r': o.mut_rate = atof(optarg); break;
        case 'F': o.mut_freq = atof(optarg); break;
        case 'R': o.indel_frac = atof(optarg); break;
        case 'X': o.indel_extend = atof(optarg); break;
        case 'I': o.indel_min = xatoi(optarg, 'I', 0); break;
        case 'c': o.data_type = xatoi(optarg, 'c', 0); break;
        case 'S': o.strandedness = xatoi(optarg, 'S', 0); break;
        case 'A': o.read_one_strand = xatoi(optarg, 'A', 0); break;
        case 'n': o.max_n = xatoi(optarg, 'n', 0); break;
        case 'y': o.rand_read = atof(optarg); break;
        case 'H': o.is_hap = 1; break;
        case 'h': return usage(&o);
        case 'z': o.seed = xatoi(optarg, 'z', 1); break;
        case 'M': o.output_type = xatoi(optarg, 'M', 0); break;
        case 'P': prefix_s = optarg; o.read_prefix = prefix_s.c_str(); break;
        case 'q': fixedq_s = optarg; if (fixedq_s.size() != 1) { fprintf(stderr, "Error: command line option -q requires one character\n"); return usage(&o); } o.fixed_quality = (unsigned char)fixedq_s[0]; break;
        case 'Q': o.quality_std = atof(optarg); break;
        case 'o': o.reads_output_type = atoi(optarg); break;
        case 'a': o.amplicons = 1; break;
        case 'f': flow_s = optarg; o.flow_order = flow_s.c_str(); break;
        case 'm': muts_fn = optarg; muts_type = 1; muts_flags |= 1; break;
        case 'b': muts_fn = optarg; muts_type = 0; muts_flags |= 2; break;
        case 'v': muts_fn = optarg; muts_type = 2; muts_flags |= 4; break;
        case 'x': regions_fn = optarg; break;
        case 'B': o.use_base_error = 1; break;
        default: fprintf(stderr, "Unrecognized option: -%c\n", c); return usage(&o);
        }
    }
    if (argc - optind < 2) return usage(&o);
    if (muts_flags != 0 && muts_flags != 1 && muts_flags != 2 && muts_flags != 4) { fprintf(stderr, "Error: -m/-b/-v cannot be used together\n"); return usage(&o); }
    if (o.read_prefix) fprintf(stderr, "Warning: remember to use the -P option with dwgsim_eval\n");
    if (o.seed == -1) o.seed = (int32_t)(time(0) & 0x7fffffff);
    if (o.seed < 0) o.seed &= 0x7fffffff;
    char msg[512];
    if (dwgsim_hip_params_check(&o, msg, sizeof msg) != DWGSIM_HIP_OK) { fprintf(stderr, "%s", msg); return usage(&o); }
    if (o.output_type == 1) fprintf(stderr, "[dwgsim_core] note: the reference dereferences a NULL VCF handle with -M 1; dwgsim-hip simply writes no mutation files\n");

    const bool want_mut = o.output_type != 1, want_reads = o.output_type != 2;
    // devices
    std::vector<int> devs;
    if (const char *e = getenv("DWGSIM_HIP_DEVICES")) {
        if (strchr(e, ',')) { for (const char *q = e; *q;) { devs.push_back(atoi(q)); const char *k = strchr(q, ','); if (!k) break; q = k + 1; } }
        else { const int n = atoi(e); for (int d = 0; d < n; ++d) devs.push_back(d); }
    }
    if (devs.empty()) devs.push_back(getenv("DWGSIM_HIP_DEVICE") ? atoi(getenv("DWGSIM_HIP_DEVICE")) : 0);
    const int ND = (int)devs.size();
    unsigned nthreads = usable_cores();
    if (const char *e = getenv("DWGSIM_HIP_THREADS")) nthreads = (unsigned)atoi(e);
    bool gpu_gzip = true;
    if (const char *e = getenv("DWGSIM_HIP_GZIP")) {
        if (!strcmp(e, "cpu")) gpu_gzip = false;
        else if (strcmp(e, "gpu")) { fprintf(stderr, "dwgsim-hip: DWGSIM_HIP_GZIP must be gpu or cpu\n"); return 1; }
    }
    int gz_level = 1;
    if (const char *e = getenv("DWGSIM_HIP_GZIP_LEVEL")) { gz_level = atoi(e); if (gz_level < 0 || gz_level > 9) gz_level = 1; }
    uint64_t min_share = 65536;
    if (const char *e = getenv("DWGSIM_HIP_MIN_SHARE")) min_share = (uint64_t)atoll(e);       // (tests: force tiny contigs onto several devices)
    uint64_t BATCH = 1u << 20;
    if (const char *e = getenv("DWGSIM_HIP_BATCH")) { const long long v = atoll(e); if (v > 0) BATCH = (uint64_t)v; }

    const char *fn_fa = argv[optind], *out_prefix = argv[optind + 1];
    Fasta fa;
    if (!read_fasta(fn_fa, fa)) return 1;
    const double t_fasta = now_s();
    // the contig table -- names, lengths, their number and sum -- comes from <in.ref.fa>.fai when that file exists (dwgsim.c:465-478:
    // the VCF header, tot_len, n_ref and the table the mutation / region files are checked against), else from the FASTA itself
    std::vector<std::string> tab_names; std::vector<int64_t> tab_lens;
    if (FILE *fai = fopen((std::string(fn_fa) + ".fai").c_str(), "r")) {
        char nmbuf[4096]; int ll, d0, d1, d2;
        while (0 < fscanf(fai, "%4095s\t%d\t%d\t%d\t%d", nmbuf, &ll, &d0, &d1, &d2)) { tab_names.push_back(nmbuf); tab_lens.push_back(ll); }
        fclose(fai);
    } else for (size_t i = 0; i < fa.seqs.size(); ++i) { tab_names.push_back(fa.names[i]); tab_lens.push_back((int64_t)fa.seqs[i].size()); }
    uint64_t tot_len = 0;
    for (size_t i = 0; i < tab_names.size(); ++i) { fprintf(stderr, "[dwgsim_core] %s length: %d\n", tab_names[i].c_str(), (int)tab_lens[i]); tot_len += (uint64_t)tab_lens[i]; }
    fprintf(stderr, "[dwgsim_core] %d sequences, total length: %llu\n", (int)tab_names.size(), (unsigned long long)tot_len);

    const bool has_bfast = want_reads && o.reads_output_type != 1, has_bwa = want_reads && o.reads_output_type != 2;
    FILE *fp_txt = nullptr, *fp_vcf = nullptr, *fgz[3] = {nullptr, nullptr, nullptr};
    std::string p = out_prefix;
    if (want_mut) {
        fp_txt = fopen((p + ".mutations.txt").c_str(), "w"); fp_vcf = fopen((p + ".mutations.vcf").c_str(), "w");
        if (!fp_txt || !fp_vcf) { fprintf(stderr, "[main] fail to open mutation files for '%s'. Abort!\n", out_prefix); return 1; }
        fprintf(fp_vcf, "##fileformat=VCFv4.1\n");
        for (size_t i = 0; i < tab_names.size(); ++i) fprintf(fp_vcf, "##contig=<ID=%s,length=%d>\n", tab_names[i].c_str(), (int)tab_lens[i]);
        fprintf(fp_vcf, "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n"
                        "##INFO=<ID=pl,Number=1,Type=Integer,Description=\"Phasing: 1 - HET contig 1, #2 - HET contig #2, 3 - HOM both contigs\">\n"
                        "##INFO=<ID=mt,Number=1,Type=String,Description=\"Variant Type: SUBSTITUTE/INSERT/DELETE\">\n"
                        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");
    }
    if (has_bwa) { fgz[0] = fopen((p + ".bwa.read1.fastq.gz").c_str(), "wb"); fgz[1] = fopen((p + ".bwa.read2.fastq.gz").c_str(), "wb"); if (!fgz[0] || !fgz[1]) { fprintf(stderr, "fail to open FASTQ outputs\n"); return 1; } }
    if (has_bfast) { fgz[2] = fopen((p + ".bfast.fastq.gz").c_str(), "wb"); if (!fgz[2]) { fprintf(stderr, "fail to open FASTQ outputs\n"); return 1; } }

    // one context per device
    std::vector<dwgsim_hip_ctx_t *> ctx((size_t)ND, nullptr);
    std::vector<const char *> nm; std::vector<int64_t> ln;
    for (size_t i = 0; i < tab_names.size(); ++i) { nm.push_back(tab_names[i].c_str()); ln.push_back(tab_lens[i]); }
    auto destroy_all = [&]() { for (auto *x : ctx) if (x) dwgsim_hip_destroy(x); };
    for (int d = 0; d < ND; ++d) {
        int err = 0;
        ctx[(size_t)d] = dwgsim_hip_create(&o, devs[(size_t)d], &err);
        if (!ctx[(size_t)d]) { fprintf(stderr, "dwgsim-hip: cannot create a GPU context on device %d (error %d)\n", devs[(size_t)d], err); destroy_all(); return 1; }
        if (!regions_fn.empty()) {   // dwgsim.c:499-506
            uint64_t tl = 0;
            if (dwgsim_hip_set_regions(ctx[(size_t)d], regions_fn.c_str(), nm.data(), ln.data(), (int)nm.size(), &tl) < 0) { fprintf(stderr, "%s", dwgsim_hip_last_error(ctx[(size_t)d])); destroy_all(); return 1; }
            tot_len = tl;
        }
        if (gpu_gzip && want_reads && dwgsim_hip_set_gzip(ctx[(size_t)d], 1) < 0) { fprintf(stderr, "%s", dwgsim_hip_last_error(ctx[(size_t)d])); destroy_all(); return 1; }
        if (muts_type >= 0 && dwgsim_hip_set_mutation_input(ctx[(size_t)d], muts_type, muts_fn.c_str(), nm.data(), ln.data(), (int)nm.size()) < 0) {     // dwgsim.c:494-497
            fprintf(stderr, "%s", dwgsim_hip_last_error(ctx[(size_t)d])); destroy_all(); return 1;
        }
    }

    // host side of the output pipeline: per device three page-locked buffer sets sized for one batch
    std::unique_ptr<Output> out;
    if (want_reads) {
        out = std::make_unique<Output>(fgz[0], fgz[1], fgz[2], ND, nthreads, gz_level, 3);
        if (out->failed()) { destroy_all(); return 1; }
    }

    const double t_ctx = now_s();
    double t_walk = 0, t_sim = 0;
    int64_t n_sim = 0; uint64_t rand_ii = 0, ctr = 0; int n_ref = (int)tab_names.size(), prev_skip = 0;
    std::atomic<int> rc{0};
    std::mutex err_m;
    auto fail = [&](const char *what) { std::lock_guard<std::mutex> g(err_m); if (rc.exchange(1) == 0) fprintf(stderr, "%s%s", what, (what[0] && what[strlen(what) - 1] != '\n') ? "\n" : ""); };
    for (size_t ci = 0; ci < fa.seqs.size() && rc == 0; ++ci) {
        const int64_t l = (int64_t)fa.seqs[ci].size(); const char *name = fa.names[ci].c_str();
        --n_ref;
        int64_t n_pairs = 0, l_eff = l;
        if (want_reads) {
            const bool last_takes_rest = n_ref == 0 && o.C < 0;     // dwgsim.c:535-537
            if (!regions_fn.empty() && !last_takes_rest) {
                l_eff = dwgsim_hip_contig_region_length(ctx[0], (uint32_t)ci, fa.seqs[ci].data(), l);
                if (l_eff == -10) { fprintf(stderr, "[dwgsim_core] #0 skip sequence '%s' as it is not in the targeted region\n", name); continue; }
                if (l_eff == -11) { fprintf(stderr, "[dwgsim_core] #1 skip sequence '%s' as more than 95%% of its targeted bases are non-ACGT\n", name); continue; }
            }
            n_pairs = dwgsim_hip_pairs_for_contig(&o, l_eff, tot_len, n_ref == 0, n_sim);
            if (n_pairs < 0) {
                if (!prev_skip) fprintf(stderr, "\n");
                prev_skip = 1;
                if (n_pairs == -2) fprintf(stderr, "[dwgsim_core] #2 skip sequence '%s' as it is shorter than the read length %d < %d!\n", name, (int)l, o.length[0] > o.length[1] ? o.length[0] : o.length[1]);
                else if (n_pairs == -3) fprintf(stderr, "[dwgsim_core] #3 skip sequence '%s' as it is shorter than %f!\n", name, o.dist + 3 * o.std_dev);
                else if (n_pairs == -4) fprintf(stderr, "[dwgsim_core] #4 skip sequence '%s' as it is shorter than %d!\n", name, (l < o.length[0]) ? o.length[0] : o.length[1]);
                else fprintf(stderr, "[dwgsim_core] #5 skip sequence '%s' as not enough pairs found\n", name);
                continue;
            }
            prev_skip = 0;
        }
        // devices that take part in this contig: a share of less than 2^16 pairs per device is not worth a second copy of the contig
        int nd = ND;
        while (nd > 1 && (uint64_t)n_pairs / (uint64_t)nd < min_share) --nd;
        std::vector<int> cid((size_t)nd, -1);
        const double t_c0 = now_s();
        on_all(nd, [&](int d) {
            dwgsim_hip_ctx_t *x = ctx[(size_t)d];
            cid[(size_t)d] = dwgsim_hip_add_contig(x, name, fa.seqs[ci].data(), l, (uint32_t)ci);
            if (cid[(size_t)d] >= 0 && !regions_fn.empty()) dwgsim_hip_contig_set_placement_length(x, cid[(size_t)d], l_eff);
            if (cid[(size_t)d] < 0 || dwgsim_hip_mutate_contig(x, cid[(size_t)d]) < 0) fail((std::string("dwgsim-hip: ") + dwgsim_hip_last_error(x)).c_str());
        });
        if (rc == 0 && want_mut) {
            const char *t, *v; size_t tl, vl;
            if (dwgsim_hip_mutations_text(ctx[0], cid[0], &t, &tl, &v, &vl) < 0) fail((std::string("dwgsim-hip: ") + dwgsim_hip_last_error(ctx[0])).c_str());
            else { fwrite(t, 1, tl, fp_txt); fwrite(v, 1, vl, fp_vcf); }
        }
        const double t_c1 = now_s(); t_walk += t_c1 - t_c0;
        if (rc == 0 && want_reads) {
            // read-index ranges, in order; rand_ii offsets from one integer per range (dwgsim.c:1042,1096)
            std::vector<uint64_t> first((size_t)nd), cnt((size_t)nd), rnd((size_t)nd, 0), rbase((size_t)nd, rand_ii);
            for (int d = 0; d < nd; ++d) dwgsim_hip_shard_range((uint64_t)n_pairs, d, nd, &first[(size_t)d], &cnt[(size_t)d]);
            if (nd > 1) {
                on_all(nd - 1, [&](int d) { if (dwgsim_hip_count_random(ctx[(size_t)d], cid[(size_t)d], first[(size_t)d], cnt[(size_t)d], &rnd[(size_t)d]) < 0) fail(dwgsim_hip_last_error(ctx[(size_t)d])); });
                for (int d = 1; d < nd; ++d) rbase[(size_t)d] = rbase[(size_t)d - 1] + rnd[(size_t)d - 1];
            }
            // per range: the abort-rule summaries of its batches (joined across ranges below) and its random reads
            std::vector<std::vector<uint64_t>> segs((size_t)nd);
            std::vector<uint64_t> got_rand((size_t)nd, 0);
            std::atomic<uint64_t> progress{ctr};
            on_all(nd, [&](int d) {
                dwgsim_hip_ctx_t *x = ctx[(size_t)d];
                const uint64_t f0 = first[(size_t)d], n_all = cnt[(size_t)d];
                struct Pending { bool live = false; int slot = 0; uint64_t n = 0; } prev;
                auto finish_batch = [&](Pending &pb) {       // wait for the kernels, start and await the copies, hand the text to the deflate stage
                    if (!pb.live) return;
                    pb.live = false;
                    dwgsim_hip_batch_t b;
                    if (dwgsim_hip_wait(x, pb.slot, &b) < 0) { fail(dwgsim_hip_last_error(x)); return; }
                    for (int q = 0; q < 4; ++q) segs[(size_t)d].push_back(b.fail_seg[q]);
                    got_rand[(size_t)d] += b.n_random;
                    TextBuf *tb = out->acquire(d, gpu_gzip ? b.gz_bytes : b.bytes);
                    if (!tb) { fail("dwgsim-hip: writing FASTQ failed"); return; }
                    tb->raw = gpu_gzip;
                    for (int s = 0; s < 3; ++s) {
                        tb->n[s] = gpu_gzip ? b.gz_bytes[s] : b.bytes[s]; tb->text_n[s] = b.bytes[s];
                        if (tb->n[s] && (gpu_gzip ? dwgsim_hip_fetch_gz_async(x, pb.slot, s, tb->p[s], tb->cap[s]) : dwgsim_hip_fetch_async(x, pb.slot, s, tb->p[s], tb->cap[s])) < 0) { fail(dwgsim_hip_last_error(x)); tb->n[0] = tb->n[1] = tb->n[2] = 0; out->submit(d, tb); return; }
                    }
                    if (dwgsim_hip_fetch_wait(x, pb.slot) < 0) { fail(dwgsim_hip_last_error(x)); tb->n[0] = tb->n[1] = tb->n[2] = 0; }
                    out->submit(d, tb);
                    const uint64_t done = progress.fetch_add(pb.n) + pb.n;
                    if (d == 0) fprintf(stderr, "\r[dwgsim_core] %llu", (unsigned long long)done);
                };
                int k = 0;
                for (uint64_t off = 0; off < n_all && rc == 0; off += BATCH, ++k) {
                    const uint64_t n = n_all - off < BATCH ? n_all - off : BATCH;
                    const int slot = k & 1;
                    if (dwgsim_hip_simulate_async(x, cid[(size_t)d], f0 + off, n, k == 0 ? rbase[(size_t)d] : DWGSIM_HIP_RAND_CHAIN, slot) < 0) { fail(dwgsim_hip_last_error(x)); break; }
                    finish_batch(prev);                  // batch k-1 is copied out while batch k runs
                    prev.live = true; prev.slot = slot; prev.n = n;
                }
                if (rc == 0) finish_batch(prev);
                else if (prev.live) { dwgsim_hip_batch_t b; (void)dwgsim_hip_wait(x, prev.slot, &b); }
                out->end_lane(d);
            });
            for (int d = nd; d < ND; ++d) out->end_lane(d);      // lanes that sat this contig out
            // the reference's failure counter over the whole contig, ranges joined in read-index order (dwgsim.c:635, :833-843)
            uint64_t acc[4] = {0, 0, 0, 0}; bool aborted = false;
            for (int d = 0; d < nd && !aborted; ++d)
                for (size_t q = 0; q + 4 <= segs[(size_t)d].size(); q += 4) if (dwgsim_hip_failseg_join(acc, &segs[(size_t)d][q])) { aborted = true; break; }
            if (aborted && rc == 0) fail("\r[dwgsim_core] failed to generate a read after 10001 trials\n");
            for (int d = 0; d < nd; ++d) rand_ii += got_rand[(size_t)d];
            n_sim += n_pairs; ctr += (uint64_t)n_pairs;
            fprintf(stderr, "\r[dwgsim_core] %llu", (unsigned long long)ctr);
        }
        t_sim += now_s() - t_c1;
        for (int d = 0; d < nd; ++d) if (cid[(size_t)d] >= 0) dwgsim_hip_drop_contig(ctx[(size_t)d], cid[(size_t)d]);
    }
    const double t_gpu_done = now_s();
    if (out) { out->finish(); if (out->failed() && rc == 0) { fprintf(stderr, "dwgsim-hip: writing FASTQ failed\n"); rc = 1; } }
    const double t_out_done = now_s();
    fprintf(stderr, "\n[dwgsim_core] Complete!\n");
    if (timing) fprintf(stderr, "[dwgsim-hip] read FASTA %.2f s | contexts, files, inputs %.2f s | upload + walk + mutation text %.2f s | simulate + copy (deflate running behind) %.2f s | "
                                "drain deflate + write %.2f s | total %.2f s; text %.2f GB -> gz %.2f GB, %d device(s), %s\n",
                        t_fasta - t_start, t_ctx - t_fasta, t_walk, t_sim, t_out_done - t_gpu_done, t_out_done - t_start,
                        out ? out->bytes_in() / 1e9 : 0.0, out ? out->bytes_out() / 1e9 : 0.0, ND,
                        gpu_gzip ? "gzip members made on the GPU" : (std::string("zlib level ") + std::to_string(gz_level) + " on " + std::to_string(nthreads) + " host threads").c_str());
    destroy_all();
    out.reset();
    if (fp_txt) fclose(fp_txt);
    if (fp_vcf) fclose(fp_vcf);
    for (int s = 0; s < 3; ++s) close_gz(fgz[s], gz_level);
    return rc.load();
}
