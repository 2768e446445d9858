// dwgsim_cli.cpp -- `dwgsim-hip`: the dwgsim command line (reference src/dwgsim.c:1123-1184 main(),
// src/dwgsim_opt.c:204-472 option surface) driving the MI355X hot path through the job level of the C-ABI
// (include/dwgsim_hip.h, dwgsim_hip_job_*).  Same usage, same defaults, same output file names:
//     dwgsim-hip [options] <in.ref.fa> <out.prefix>
//     <prefix>.mutations.txt / .mutations.vcf / .bfast.fastq.gz / .bwa.read1.fastq.gz / .bwa.read2.fastq.gz
// What is left here: option parsing, the FASTA reader (mut.c:49-87), the five files.  Everything between -- contig scheduling, grouping of
// small contigs, the pipelines of all GPUs, sharding, ordering -- is the library's (dw_job.cpp); the sink below writes what it delivers:
// gzip members made on the GPU as they are, or (DWGSIM_HIP_GZIP=cpu) text deflated here on the host cores.  A multi-member .gz
// decompresses to exactly the concatenated text, which is what the reference's own test compares (testdata/test.sh:23-25).
// Environment (all optional):
//     DWGSIM_HIP_DEVICES   "0,1,2,3" or a count "4" (default: every HIP device the process sees)
//     DWGSIM_HIP_GZIP      "gpu" (default): 32 KiB members made by k_gzip (Huffman codes + matches on the name lines), ~0.44 of the text; "cpu": zlib on the host cores at
//                          DWGSIM_HIP_GZIP_LEVEL (default 1; smaller files, deflate-bound) with DWGSIM_HIP_THREADS threads (default: all cores)
//     DWGSIM_HIP_BATCH     read pairs per GPU launch (default 2^20)
//     DWGSIM_HIP_GROUP_BP  consecutive contigs are resident together up to this many bases (default 32 Mi)
//     DWGSIM_HIP_MIN_SHARE a group is spread over fewer devices while a device's share would be below this many pairs (default 65536)
//     DWGSIM_HIP_READ_THREADS threads that index, verify and copy the mapped FASTA (default: all cores); DWGSIM_HIP_READ_CHUNK bytes of text per task (4 Mi)
//     DWGSIM_HIP_TIMING    print the stage times
//     DWGSIM_HIP_TEARDOWN  free every buffer and context before returning from main (by default the process ends as soon as the files are closed)
//     DWGSIM_HIP_SINK      "null": measurement aid -- the FASTQ deliveries are counted, not written (the .gz files stay empty); "memcpy": every delivery is copied
//                          once into a scratch buffer by the delivering thread (the least any consumer does with it): with DWGSIM_HIP_TIMING the line reports the time
//                          the per-stream delivery threads spent inside the sink, i.e. the rate ONE delivery thread can sustain (the job level's ceiling per stream)
//     DWGSIM_HIP_SINK_ORDERED  the FASTQ pieces through the ordered sink (one delivering thread per file, write() in file order) instead of the offset sink (one
//                          thread per device and file, pwrite() at the piece's place: the default when the members are made on the GPU)
//     DWGSIM_HIP_SOLO      "r/W": measurement aid (dw_job.cpp) -- this process's one device does what device r of a W-device job does, nothing else
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sched.h>
#include <limits.h>
#include <zlib.h>
#include <string>
#include <algorithm>
#include <vector>
#include <deque>
#include <thread>
#include <future>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <memory>
#include <functional>
#include "../../include/dwgsim_hip.h"

#define PACKAGE_VERSION "0.1.17-hip"

// The option summary of the reference (dwgsim_opt.c:93-160), line for line -- it is the interface a drop-in shows -- behind this program's own
// banner; the bracketed values are the current settings, printed as the reference prints them (an error-rate step that is only computed
// after parsing, the inverted "not using" of -b / -v, "(null)" for strings that were not given).
struct UsageExtras { const char *flow, *muts_fn, *regions_fn, *prefix, *fixed_quality; int muts_type; };      // muts_type: 0 bed, 1 txt, 2 vcf (mut_input.h:33-37), -1 none
static const char *str_or_null(const char *s) { return s ? s : "(null)"; }
static int usage(const dwgsim_hip_params_t *o, const UsageExtras &x)
{
    auto yes = [](int v) { return v == 1 ? "True" : "False"; };
    const int mt = x.muts_type;      // (-1 while none was given: dwgsim_opt.c:77)
    FILE *f = stderr;
    fprintf(f, "\nProgram: dwgsim-hip (short read simulator, MI355X hot path of dwgsim)\nVersion: %s\n\n", PACKAGE_VERSION);
    fprintf(f, "Usage:   dwgsim-hip [options] <in.ref.fa> <out.prefix>\n\nOptions:\n");
    const char *P = "         ";
    fprintf(f, "%s-e FLOAT      per base/color/flow error rate of the first read [from %.3f to %.3f by %.3f]\n", P, o->e_start[0], o->e_end[0], 0.0);
    fprintf(f, "%s-E FLOAT      per base/color/flow error rate of the second read [from %.3f to %.3f by %.3f]\n", P, o->e_start[1], o->e_end[1], 0.0);
    fprintf(f, "%s-i            use the inner distance instead of the outer distance for pairs [%s]\n", P, yes(o->is_inner));
    fprintf(f, "%s-d INT        %s distance between the two ends for pairs [%d]\n", P, o->is_inner ? "inner" : "outer", o->dist);
    fprintf(f, "%s-s INT        standard deviation of the distance for pairs [%.3f]\n", P, o->std_dev);
    fprintf(f, "%s-N INT        number of read pairs (-1 to disable) [%lld]\n", P, (long long)o->N);
    fprintf(f, "%s-C FLOAT      mean coverage across available positions (-1 to disable) [%.2lf]\n", P, o->C);
    fprintf(f, "%s-1 INT        length of the first read [%d]\n", P, o->length[0]);
    fprintf(f, "%s-2 INT        length of the second read [%d]\n", P, o->length[1]);
    fprintf(f, "%s-r FLOAT      rate of mutations [%.4f]\n", P, o->mut_rate);
    fprintf(f, "%s-F FLOAT      frequency of given mutation to simulate low fequency somatic mutations [%.4f]\n", P, o->mut_freq);
    fprintf(f, "%s                  NB: freqeuncy F refers to the first strand of mutation, therefore mutations \n", P);
    fprintf(f, "%s                  on the second strand occur with a frequency of 1-F \n", P);
    fprintf(f, "%s-R FLOAT      fraction of mutations that are indels [%.2f]\n", P, o->indel_frac);
    fprintf(f, "%s-X FLOAT      probability an indel is extended [%.2f]\n", P, o->indel_extend);
    fprintf(f, "%s-I INT        the minimum length indel [%d]\n", P, o->indel_min);
    fprintf(f, "%s-y FLOAT      probability of a random DNA read [%.2f]\n", P, o->rand_read);
    fprintf(f, "%s-n INT        maximum number of Ns allowed in a given read [%d]\n", P, o->max_n);
    fprintf(f, "%s-c INT        generate reads for [%d]:\n", P, o->data_type);
    for (const char *l : {"0: Illumina", "1: SOLiD", "2: Ion Torrent"}) fprintf(f, "%s                  %s\n", P, l);
    fprintf(f, "%s-S INT        generate paired end reads with orientation [%d]:\n", P, o->strandedness);
    for (const char *l : {"0: default (opposite strand for Illumina, same strand for SOLiD/Ion Torrent)", "1: same strand (mate pair)", "2: opposite strand (paired end)"}) fprintf(f, "%s                  %s\n", P, l);
    fprintf(f, "%s-A INT        generate paired end reads with read one [%d]:\n", P, o->read_one_strand);
    for (const char *l : {"0: default (both, random)", "1: forward genomic strand", "2: reverse genomic strand"}) fprintf(f, "%s                  %s\n", P, l);
    fprintf(f, "%s-f STRING     the flow order for Ion Torrent data [%s]\n", P, str_or_null(x.flow));
    fprintf(f, "%s-B            use a per-base error rate for Ion Torrent data [%s]\n", P, yes(o->use_base_error));
    fprintf(f, "%s-H            haploid mode [%s]\n", P, yes(o->is_hap));
    fprintf(f, "%s-z INT        random seed (-1 uses the current time) [%d]\n", P, o->seed);
    fprintf(f, "%s-M            output files to generate [%d]:\n", P, o->output_type);
    for (const char *l : {"0: both reads and mutation files", "1: reads only", "2: mutations only"}) fprintf(f, "%s                  %s\n", P, l);
    fprintf(f, "%s-m FILE       the mutations txt file to re-create [%s]\n", P, mt != 1 ? "not using" : str_or_null(x.muts_fn));
    fprintf(f, "%s-b FILE       the bed-like file set of candidate mutations [%s]\n", P, mt == 0 ? "not using" : str_or_null(x.muts_fn));
    fprintf(f, "%s-v FILE       the vcf file set of candidate mutations (use pl tag for strand) [%s]\n", P, mt == 2 ? "not using" : str_or_null(x.muts_fn));
    fprintf(f, "%s-x FILE       the bed of regions to cover [%s]\n", P, x.regions_fn ? x.regions_fn : "not using");
    fprintf(f, "%s-P STRING     a read prefix to prepend to each read name [%s]\n", P, x.prefix ? x.prefix : "not using");
    fprintf(f, "%s-q STRING     a fixed base quality to apply (single character) [%s]\n", P, x.fixed_quality ? x.fixed_quality : "not using");
    fprintf(f, "%s-Q FLOAT      standard deviation of the base quality scores [%.2lf]\n", P, x.fixed_quality ? 0.0 : o->quality_std);
    fprintf(f, "%s-o INT        output type for the FASTQ files [%d]:\n", P, o->reads_output_type);
    for (const char *l : {"0: interleaved (bfast) and per-read-end (bwa)", "1: per-read-end (bwa) only", "2: interleaved (bfast) only"}) fprintf(f, "%s                  %s\n", P, l);
    fprintf(f, "%s-a            assume each contig is an amplicon, so read pairs will sequence the full amplicon [%s]\n", P, yes(o->amplicons));
    fprintf(f, "%s-h            print this message\n\n", P);
    fprintf(f, "Note: For SOLiD mate pair reads and BFAST, the first read is F3 and the second is R3. For SOLiD mate pair reads\n"
               "and BWA, the reads in the first file are R3 the reads annotated as the first read etc.\n\n");
    fprintf(f, "Note: The longest supported insertion is %u.\n\n", UINT32_MAX);
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

// mut.c:49-87 seq_read_fasta: the name is the header up to the first blank, the sequence keeps isalpha, '-' and '.'; a '>' opens a new
// record wherever it stands.  Every finished record goes to `on_record` at once, so the first contigs are on the GPU while the rest of
// the file is still being read.  The reader is what a whole-genome run waits for (3.1 GB of text): a regular file is mapped instead of
// copied through stdio, lines are found with memchr, and a line of plain letters -- all but the headers of a usual FASTA -- is verified
// eight bytes at a time and appended with one copy; anything else goes character by character as the reference does.
// `expect(k)`: the length of the k-th record when the caller knows it (the .fai), so that its bases are stored without re-allocations.
struct FastaParser {
    const std::function<bool(const std::string &, std::vector<uint8_t> &)> &on_record;
    const std::function<int64_t(size_t)> &expect;
    std::string name; std::vector<uint8_t> seq; int state = 0;   // 0 before first '>', 1 in name, 2 rest of header line, 3 sequence
    bool have = false, go = true; size_t n_rec = 0;
    bool keep[256];
    FastaParser(const std::function<bool(const std::string &, std::vector<uint8_t> &)> &f, const std::function<int64_t(size_t)> &e) : on_record(f), expect(e)
    {
        for (int c = 0; c < 256; ++c) keep[c] = isalpha(c) || c == '-' || c == '.';
    }
    static bool all_letters(const char *p, size_t n)
    {
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t x; memcpy(&x, p + i, 8);
            if (x & 0x8080808080808080ull) return false;
            const uint64_t y = x | 0x2020202020202020ull;                                  // 7-bit values, lower case
            const uint64_t ge_a = y + 0x1F1F1F1F1F1F1F1Full, gt_z = y + 0x0505050505050505ull;   // bit 7 of a byte: y >= 'a', y > 'z' (no carries: y <= 0x7F)
            if ((ge_a & ~gt_z & 0x8080808080808080ull) != 0x8080808080808080ull) return false;
        }
        for (; i < n; ++i) if (!isalpha((unsigned char)p[i])) return false;
        return true;
    }
    void open_record() { state = 1; name.clear(); seq.clear(); const int64_t e = expect ? expect(n_rec) : 0; if (e > 0) seq.reserve((size_t)e); ++n_rec; }
    void feed(const char *buf, size_t n)
    {
        size_t i = 0;
        while (i < n && go) {
            const int c = (unsigned char)buf[i];
            if (state == 3) {
                if (c != '>' ) {
                    const char *nl = (const char *)memchr(buf + i, '\n', n - i);
                    const size_t e = nl ? (size_t)(nl - buf) : n;
                    if (all_letters(buf + i, e - i)) { seq.insert(seq.end(), (const uint8_t *)buf + i, (const uint8_t *)buf + e); i = nl ? e + 1 : e; continue; }
                    // the run of sequence characters from here, then whatever stops it
                    size_t j = i;
                    while (j < n && keep[(unsigned char)buf[j]]) ++j;
                    if (j > i) { seq.insert(seq.end(), (const uint8_t *)buf + i, (const uint8_t *)buf + j); i = j; continue; }
                    ++i;
                    continue;
                }
                go = on_record(name, seq); open_record();
                ++i;
                continue;
            }
            if (state == 0) { if (c == '>') { open_record(); have = true; } }
            else if (state == 1) { if (c == ' ' || c == '\t') state = 2; else if (c == '\n') state = 3; else if (c != '\r') name.push_back((char)c); }
            else if (state == 2) { if (c == '\n') state = 3; }
            ++i;
        }
    }
    bool finish() { if (have && go) go = on_record(name, seq); return go; }
};
static bool read_fasta(const char *fn, const std::function<bool(const std::string &, std::vector<uint8_t> &)> &on_record, const std::function<int64_t(size_t)> &expect = nullptr)
{
    FastaParser ps(on_record, expect);
    std::vector<char> buf((size_t)1 << 24);
    if (strcmp(fn, "-")) {
        const int fd = open(fn, O_RDONLY);
        if (fd < 0) { fprintf(stderr, "[main] fail to open file '%s'. Abort!\n", fn); return false; }      // (xopen in main, dwgsim.c:225, :1139)
        ssize_t n;
        while (ps.go && (n = read(fd, buf.data(), buf.size())) > 0) ps.feed(buf.data(), (size_t)n);
        close(fd);
        return ps.finish();
    }
    size_t n;
    while (ps.go && (n = fread(buf.data(), 1, buf.size(), stdin)) > 0) ps.feed(buf.data(), n);
    return ps.finish();
}

// ---- the reader of a regular file: mapped, indexed, and parsed by all host cores ----
// The reference reads the FASTA with one fgetc per character on one thread (mut.c:49-87) in front of a loop that needs minutes per chromosome;
// here the GPUs take a chromosome in a tenth of a second, and one thread parsing 3 GB of text was what a whole-genome run waited for.  So:
//   1. every '>' of the mapped file is found by memchr, all cores side by side; a '>' opens a record unless it stands in a header line
//      (FastaParser above: state 1 / 2), which a sequential pass over the (few) candidates settles -> records [header | body);
//   2. a body is REGULAR when every line holds the same number L of letters followed by '\n' and the last one 1 .. L letters with or without
//      '\n' -- what every FASTA writer produces.  Then the sequence length and the place of every base follow from the byte count, and the lines
//      are verified (letters only, '\n' where it must be) and copied by all cores straight into the job's page-locked staging
//      (dwgsim_hip_job_begin_contig / commit_contig): the sequence is touched once;
//   3. a body that is not (blank lines, '\r', '-' or '.', ragged lines: the verification says so) goes through FastaParser, the
//      reference's own state machine -- same bytes as before, on one thread.
class Pool {
public:
    explicit Pool(unsigned n) { for (unsigned t = 1; t < (n ? n : 1); ++t) th_.emplace_back([this]() { work(); }); }
    ~Pool() { { std::lock_guard<std::mutex> g(m_); stop_ = true; } cv_.notify_all(); for (auto &t : th_) t.join(); }
    unsigned size() const { return (unsigned)th_.size() + 1; }
    void run(size_t n_tasks, const std::function<void(size_t)> &fn)      // fn(0 .. n_tasks-1), the caller works too; returns when all are done
    {
        if (n_tasks == 0) return;
        { std::lock_guard<std::mutex> g(m_); fn_ = &fn; next_ = 0; n_ = n_tasks; left_ = n_tasks; ++gen_; }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&]() { return left_ == 0; });
        fn_ = nullptr;
    }
private:
    void drain()
    {
        for (;;) {
            size_t i; const std::function<void(size_t)> *f;
            { std::lock_guard<std::mutex> g(m_); if (!fn_ || next_ >= n_) return; i = next_++; f = fn_; }
            (*f)(i);
            { std::lock_guard<std::mutex> g(m_); if (--left_ == 0) done_.notify_all(); }
        }
    }
    void work()
    {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&]() { return stop_ || gen_ != seen; }); if (stop_) return; seen = gen_; }
            drain();
        }
    }
    std::mutex m_; std::condition_variable cv_, done_; std::vector<std::thread> th_;
    const std::function<void(size_t)> *fn_ = nullptr; size_t next_ = 0, n_ = 0, left_ = 0; uint64_t gen_ = 0; bool stop_ = false;
};

struct FastaRecord { size_t gt = 0, body = 0, end = 0; std::string name; };      // '>' | first byte behind the header line | one past the record

struct MappedFasta {
    const char *p = nullptr; size_t n = 0; int fd = -1;
    size_t chunk_bytes = (size_t)4 << 20;      // text a thread verifies and copies at a time (DWGSIM_HIP_READ_CHUNK: tests make it tiny)
    std::vector<FastaRecord> rec;
    ~MappedFasta() { if (p) munmap((void *)p, n); if (fd >= 0) close(fd); }
    bool open_file(const char *fn)
    {
        fd = open(fn, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return false;
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char *)m; n = (size_t)st.st_size;
        return true;
    }
    // a record that has been handed over is not read again: its pages leave the address space now, piece by piece while the job runs, instead of all
    // 3 GB of a genome when the process ends (0.06 s per GB at exit: tools/ubench_exit.hip)
    void release(const FastaRecord &r) const
    {
        const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
        const size_t lo = (r.gt + pg - 1) / pg * pg, hi = r.end / pg * pg;
        if (hi > lo) (void)madvise((void *)(p + lo), hi - lo, MADV_DONTNEED);
    }
    void index(Pool &pool)
    {
        const size_t piece = 2 * chunk_bytes, np = (n + piece - 1) / piece;
        std::vector<std::vector<size_t>> found(np);
        pool.run(np, [&](size_t k) {
            const size_t lo = k * piece, hi = std::min(n, lo + piece);
            for (const char *q = p + lo; q < p + hi;) { const char *g = (const char *)memchr(q, '>', (size_t)(p + hi - q)); if (!g) break; found[k].push_back((size_t)(g - p)); q = g + 1; }
        });
        size_t header_end = 0;      // one past the header line of the record that is open
        for (const auto &v : found) for (size_t g : v) {
            if (!rec.empty() && g < header_end) continue;      // inside a header line: part of the name, or ignored (FastaParser states 1 / 2)
            if (!rec.empty()) rec.back().end = g;
            FastaRecord r; r.gt = g;
            const char *nl = (const char *)memchr(p + g, '\n', n - g);
            header_end = nl ? (size_t)(nl - p) + 1 : n;
            r.body = header_end; r.end = n;
            for (size_t q = g + 1; q < header_end; ++q) { const char c = p[q]; if (c == ' ' || c == '\t' || c == '\n') break; if (c != '\r') r.name.push_back(c); }
            rec.push_back(std::move(r));
        }
    }
    // a regular body: L letters per line, `len` letters in all; false: not of that shape on the face of it (the lines themselves are verified by fill)
    bool geometry(const FastaRecord &r, size_t *L, int64_t *len) const
    {
        const size_t nb = r.end - r.body;
        if (nb == 0) { *L = 1; *len = 0; return true; }
        const char *b = p + r.body;
        const char *nl = (const char *)memchr(b, '\n', nb);
        const size_t l0 = nl ? (size_t)(nl - b) : nb;
        if (l0 == 0) return false;
        const size_t q = nb / (l0 + 1), rem = nb % (l0 + 1);
        size_t last = rem;                                      // bytes of a last, shorter line
        if (rem && b[nb - 1] == '\n') { if (rem == 1) return false; last = rem - 1; }
        *L = l0; *len = (int64_t)(q * l0 + last);
        return true;
    }
    // verifies and copies a regular body into dst[0 .. len); false: some line is not what the geometry says (dst is then garbage)
    bool fill(const FastaRecord &r, size_t L, int64_t len, uint8_t *dst, Pool &pool) const
    {
        if (len == 0) return true;
        const char *b = p + r.body;
        const size_t nb = r.end - r.body, full = nb / (L + 1), rem = nb % (L + 1), last = (size_t)len - full * L;
        const size_t per = std::max<size_t>(1, chunk_bytes / (L + 1)), nt = (full + per - 1) / per;
        std::atomic<bool> ok{true};
        pool.run(nt, [&](size_t k) {
            const size_t i0 = k * per, i1 = std::min(full, i0 + per);
            for (size_t i = i0; i < i1 && ok.load(std::memory_order_relaxed); ++i) {
                const char *ln = b + i * (L + 1);
                if (ln[L] != '\n' || !FastaParser::all_letters(ln, L)) { ok = false; return; }
                memcpy(dst + i * L, ln, L);
            }
        });
        if (!ok) return false;
        if (last) {
            const char *ln = b + full * (L + 1);
            if (!FastaParser::all_letters(ln, last) || (rem == last + 1 && ln[last] != '\n')) return false;
            memcpy(dst + full * L, ln, last);
        }
        return true;
    }
    // the reference's state machine over one record (irregular bodies)
    void parse_generic(const FastaRecord &r, std::vector<uint8_t> &seq) const
    {
        std::string nm;
        const std::function<bool(const std::string &, std::vector<uint8_t> &)> take = [&](const std::string &, std::vector<uint8_t> &s) { seq.swap(s); return true; };
        const std::function<int64_t(size_t)> none;
        FastaParser ps(take, none);
        ps.feed(p + r.gt, r.end - r.gt);
        ps.finish();
    }
};

static void parallel_copy(uint8_t *dst, const uint8_t *src, size_t n, Pool &pool)
{
    const size_t piece = (size_t)8 << 20, np = (n + piece - 1) / piece;
    if (np <= 1) { if (n) memcpy(dst, src, n); return; }
    pool.run(np, [&](size_t k) { const size_t lo = k * piece; memcpy(dst + lo, src + lo, std::min(piece, n - lo)); });
}

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

// DWGSIM_HIP_GZIP=cpu: the text of a delivery is cut into 1 MiB pieces, deflated as independent gzip members by a pool shared by the three
// streams, and written in order (the reference itself feeds zlib one byte at a time, dwgsim.c:930-931: ~80 % of its wall time)
class DeflatePool {
public:
    DeflatePool(unsigned n_threads, int level) : level_(level) { for (unsigned t = 0; t < (n_threads ? n_threads : 1); ++t) th_.emplace_back([this]() { work(); }); }
    ~DeflatePool() { { std::lock_guard<std::mutex> g(m_); stop_ = true; } cv_.notify_all(); for (auto &t : th_) t.join(); }
    bool write(FILE *f, const char *text, size_t n)
    {
        const size_t CH = (size_t)1 << 20, nc = (n + CH - 1) / CH;
        std::vector<Piece> pc(nc);
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t k = 0; k < nc; ++k) { pc[k].src = text + k * CH; pc[k].n = n - k * CH < CH ? n - k * CH : CH; todo_.push_back(&pc[k]); }
        }
        cv_.notify_all();
        bool ok = true;
        for (size_t k = 0; k < nc; ++k) {
            { std::unique_lock<std::mutex> lk(m_); done_.wait(lk, [&]() { return pc[k].done; }); }
            if (!pc[k].ok || fwrite(pc[k].gz.data(), 1, pc[k].gz.size(), f) != pc[k].gz.size()) ok = false;
            std::vector<unsigned char>().swap(pc[k].gz);
        }
        return ok;
    }
private:
    struct Piece { const char *src = nullptr; size_t n = 0; std::vector<unsigned char> gz; bool done = false, ok = true; };
    void work()
    {
        for (;;) {
            Piece *p;
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&]() { return stop_ || !todo_.empty(); }); if (todo_.empty()) return; p = todo_.front(); todo_.pop_front(); }
            const bool ok = deflate_member(p->src, p->n, level_, p->gz);
            { std::lock_guard<std::mutex> g(m_); p->ok = ok; p->done = true; }
            done_.notify_all();
        }
    }
    int level_; std::mutex m_; std::condition_variable cv_, done_; std::deque<Piece *> todo_; std::vector<std::thread> th_; bool stop_ = false;
};

static void close_gz(FILE *f, int level, uint64_t written_elsewhere = 0)
{
    if (!f) return;
    if (ftell(f) == 0 && written_elsewhere == 0) { std::vector<unsigned char> e; deflate_member("", 0, level, e); fwrite(e.data(), 1, e.size(), f); }   // empty stream: still a valid .gz
    fclose(f);
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

// what the job delivers, written as the reference writes it (dwgsim.c:919-981 -> the three .gz files, mut.c:781-893 -> the two mutation files)
struct FileSink {
    FILE *fp_txt = nullptr, *fp_vcf = nullptr, *fgz[3] = {nullptr, nullptr, nullptr};
    DeflatePool *pool = nullptr; bool null_sink = false, memcpy_sink = false;
    std::atomic<uint64_t> bytes_in{0}, bytes_out{0}, sink_ns[3] = {{0}, {0}, {0}}, sink_bytes[3] = {{0}, {0}, {0}};
    std::vector<char> scratch[3];
    static int mutations(void *u, const char *, const char *txt, size_t tl, const char *vcf, size_t vl)
    {
        FileSink *s = (FileSink *)u;
        if (s->fp_txt && fwrite(txt, 1, tl, s->fp_txt) != tl) return 1;
        if (s->fp_vcf && fwrite(vcf, 1, vl, s->fp_vcf) != vl) return 1;
        return 0;
    }
    std::atomic<uint64_t> written[3] = {{0}, {0}, {0}};      // bytes that went to file `stream` through reads_at
    // pieces with their place (dwgsim_hip_job_sink_t::reads_at): several threads per file, pwrite
    static int reads_at(void *u, int stream, uint64_t offset, const void *data, size_t len, size_t text_len, int)
    {
        FileSink *s = (FileSink *)u;
        FILE *f = s->fgz[stream];
        if (!f) return 1;
        s->bytes_in += text_len; s->bytes_out += len;
        if (s->null_sink) return 0;
        if (s->memcpy_sink) {
            static thread_local std::vector<char> mine;
            timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
            if (mine.size() < len) mine.resize(len);
            memcpy(mine.data(), data, len);
            clock_gettime(CLOCK_MONOTONIC, &b);
            s->sink_ns[stream] += (uint64_t)((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec)); s->sink_bytes[stream] += len;
            return 0;
        }
        const char *p = (const char *)data; size_t left = len; off_t at = (off_t)offset;
        while (left) { const ssize_t w = pwrite(fileno(f), p, left, at); if (w <= 0) return 1; p += w; left -= (size_t)w; at += w; }
        s->written[stream] += len;
        return 0;
    }
    static int reads(void *u, int stream, const void *data, size_t len, size_t text_len, int gz)
    {
        FileSink *s = (FileSink *)u;
        FILE *f = s->fgz[stream];
        if (!f) return 1;
        s->bytes_in += text_len;
        if (s->null_sink) { s->bytes_out += len; return 0; }
        if (s->memcpy_sink) {      // (one thread per stream calls this: scratch[stream] is that thread's)
            timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
            if (s->scratch[stream].size() < len) s->scratch[stream].resize(len);
            memcpy(s->scratch[stream].data(), data, len);
            clock_gettime(CLOCK_MONOTONIC, &b);
            s->sink_ns[stream] += (uint64_t)((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec)); s->sink_bytes[stream] += len;
            s->bytes_out += len; return 0;
        }
        if (gz) { s->bytes_out += len; return fwrite(data, 1, len, f) == len ? 0 : 1; }
        const long before = ftell(f);
        if (!s->pool->write(f, (const char *)data, len)) return 1;
        s->bytes_out += (uint64_t)(ftell(f) - before);
        return 0;
    }
};

int main(int argc, char **argv)
{
    const double t_start = now_s();
    const bool timing = getenv("DWGSIM_HIP_TIMING") != nullptr;      // stage times on stderr
    dwgsim_hip_params_t o; dwgsim_hip_params_default(&o);
    std::string prefix_s, fixedq_s, flow_s, regions_fn, muts_fn; int muts_type = -1, muts_flags = 0; bool have_q = false;
    int c;
    auto usage_now = [&]() {
        const UsageExtras x{flow_s.empty() ? nullptr : flow_s.c_str(), muts_fn.empty() ? nullptr : muts_fn.c_str(), regions_fn.empty() ? nullptr : regions_fn.c_str(),
                            o.read_prefix ? prefix_s.c_str() : nullptr, have_q ? fixedq_s.c_str() : nullptr, muts_type};
        return usage(&o, x);
    };
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
        case 'h': return usage_now();
        case 'z': o.seed = xatoi(optarg, 'z', 1); break;
        case 'M': o.output_type = xatoi(optarg, 'M', 0); break;
        case 'P': prefix_s = optarg; o.read_prefix = prefix_s.c_str(); break;
        case 'q': fixedq_s = optarg; have_q = true; o.fixed_quality = fixedq_s.size() == 1 ? (unsigned char)fixedq_s[0] : -2; break;      // (checked behind the other options, as the reference does: dwgsim_opt.c:364-367)
        case 'Q': o.quality_std = atof(optarg); break;
        case 'o': o.reads_output_type = atoi(optarg); break;
        case 'a': o.amplicons = 1; break;
        case 'f': flow_s = optarg; o.flow_order = flow_s.c_str(); break;
        case 'm': muts_fn = optarg; muts_type = 1; muts_flags |= 1; break;
        case 'b': muts_fn = optarg; muts_type = 0; muts_flags |= 2; break;
        case 'v': muts_fn = optarg; muts_type = 2; muts_flags |= 4; break;
        case 'x': regions_fn = optarg; break;
        case 'B': o.use_base_error = 1; break;
        default: fprintf(stderr, "Unrecognized option: -%c\n", c); return usage_now();
        }
    }
    if (argc - optind < 2) return usage_now();
    const int seed_given = o.seed;      // (the usage text shows what was given)
    if (o.seed == -1) o.seed = (int32_t)(time(0) & 0x7fffffff);
    if (o.seed < 0) o.seed &= 0x7fffffff;
    char msg[512];
    // the checks in the reference's order (dwgsim_opt.c:307-391): ranges up to -Q (dwgsim_hip_params_check follows it), the -P warning, -o, then -m / -b / -v together
    {
        auto refuse = [&](const char *m) { fprintf(stderr, "%s", m); o.seed = seed_given; return usage_now(); };
        dwgsim_hip_params_t chk = o;
        chk.reads_output_type = 0;
        if (chk.output_type < 0 || chk.output_type > 2) chk.output_type = 0;      // (the reference does not check -M: anything but 1 and 2 writes everything, dwgsim.c:1143-1159)
        if (const int rc = dwgsim_hip_params_check(&chk, msg, sizeof msg); rc != DWGSIM_HIP_OK && rc != DWGSIM_HIP_ERR_UNSUP) return refuse(msg);
        if (o.read_prefix) fprintf(stderr, "Warning: remember to use the -P option with dwgsim_eval\n");
        if (o.reads_output_type < 0 || o.reads_output_type > 2) return refuse("Error: command line option -o was out of range\n");
        if (muts_flags != 0 && muts_flags != 1 && muts_flags != 2 && muts_flags != 4) return refuse("Error: -m/-b/-v cannot be used together\n");
        if (o.output_type < 0 || o.output_type > 2) o.output_type = 0;
        if (dwgsim_hip_params_check(&o, msg, sizeof msg) != DWGSIM_HIP_OK) return refuse(msg);
    }
    if (o.output_type == 1) fprintf(stderr, "[dwgsim_core] note: the reference dereferences a NULL VCF handle with -M 1; dwgsim-hip simply writes no mutation files\n");

    const bool want_mut = o.output_type != 1, want_reads = o.output_type != 2;
    // devices: every one the process sees, unless told otherwise
    std::vector<int> devs;
    if (const char *e = getenv("DWGSIM_HIP_DEVICES")) {
        if (strchr(e, ',')) { for (const char *q = e; *q;) { devs.push_back(atoi(q)); const char *k = strchr(q, ','); if (!k) break; q = k + 1; } }
        else { const int n = atoi(e); for (int d = 0; d < n; ++d) devs.push_back(d); }
    } else if (const char *e1 = getenv("DWGSIM_HIP_DEVICE")) devs.push_back(atoi(e1));
    unsigned nthreads = usable_cores();
    if (const char *e = getenv("DWGSIM_HIP_THREADS")) nthreads = (unsigned)atoi(e);
    bool gpu_gzip = true;
    if (const char *e = getenv("DWGSIM_HIP_GZIP")) {
        if (!strcmp(e, "cpu")) gpu_gzip = false;
        else if (strcmp(e, "gpu")) { fprintf(stderr, "dwgsim-hip: DWGSIM_HIP_GZIP must be gpu or cpu\n"); return 1; }
    }
    int gz_level = 1;
    if (const char *e = getenv("DWGSIM_HIP_GZIP_LEVEL")) { gz_level = atoi(e); if (gz_level < 0 || gz_level > 9) gz_level = 1; }
    dwgsim_hip_job_options_t jo; memset(&jo, 0, sizeof jo);
    jo.gzip = gpu_gzip ? 1 : 0;
    if (const char *e = getenv("DWGSIM_HIP_MIN_SHARE")) jo.min_share = (uint64_t)atoll(e);       // (tests: force tiny contigs onto several devices)
    if (const char *e = getenv("DWGSIM_HIP_BATCH")) { const long long v = atoll(e); if (v > 0) jo.batch_pairs = (uint64_t)v; }
    if (const char *e = getenv("DWGSIM_HIP_GROUP_BP")) { const long long v = atoll(e); if (v > 0) jo.group_bp = (uint64_t)v; }

    const char *fn_fa = argv[optind], *out_prefix = argv[optind + 1];
    // the job (runtime set-up, one context per device: about 0.1 s) is made on a thread of its own while this one reads the contig table
    FileSink fs;
    dwgsim_hip_job_sink_t sink; memset(&sink, 0, sizeof sink);
    sink.user = &fs; sink.mutations = want_mut ? FileSink::mutations : nullptr; sink.reads = want_reads ? FileSink::reads : nullptr;
    // members made on the GPU are written where they belong by one thread per device and file (pwrite); text that the host still has to deflate goes through the ordered sink
    if (want_reads && gpu_gzip && !getenv("DWGSIM_HIP_SINK_ORDERED")) sink.reads_at = FileSink::reads_at;
    int job_err = 0;
    std::future<dwgsim_hip_job_t *> job_made = std::async(std::launch::async, [&]() { return dwgsim_hip_job_create(&o, devs.empty() ? nullptr : devs.data(), (int)devs.size(), &sink, &jo, &job_err); });
    auto give_up = [&](int code) { if (dwgsim_hip_job_t *jb = job_made.get()) dwgsim_hip_job_destroy(jb); return code; };
    // the contig table -- names, lengths, their number and sum -- comes from <in.ref.fa>.fai when that file exists (dwgsim.c:465-478:
    // the VCF header, tot_len, n_ref and the table the mutation / region files are checked against): the FASTA is then read once, contig
    // after contig, each one handed to the GPUs as soon as it is complete.  Without an index the reference reads the FASTA twice; here it
    // is read once into memory and handed over from there.
    // A regular file is mapped, its records are found by all cores, and their lines are verified and copied by all cores (MappedFasta);
    // anything else (stdin, a pipe) goes through the sequential parser.
    std::vector<std::string> tab_names; std::vector<int64_t> tab_lens;
    std::vector<std::pair<std::string, std::vector<uint8_t>>> held; bool streaming = false;
    unsigned read_threads = nthreads;
    if (const char *e = getenv("DWGSIM_HIP_READ_THREADS")) { const int v = atoi(e); if (v >= 1) read_threads = (unsigned)v; }
    Pool rpool(read_threads);
    MappedFasta mf;
    if (const char *e = getenv("DWGSIM_HIP_READ_CHUNK")) { const long long v = atoll(e); if (v >= 1) mf.chunk_bytes = (size_t)v; }
    // (the reference opens the FASTA in main, dwgsim.c:1139, and the regions file at the top of dwgsim_core, :460-463, before it reads the contig table)
    if (strcmp(fn_fa, "-") != 0 && access(fn_fa, R_OK) != 0) { fprintf(stderr, "[main] fail to open file '%s'. Abort!\n", fn_fa); return give_up(1); }
    if (!regions_fn.empty() && regions_fn != "-" && access(regions_fn.c_str(), R_OK) != 0) { fprintf(stderr, "[dwgsim_core] fail to open file '%s'. Abort!\n", regions_fn.c_str()); return give_up(1); }
    const bool mapped = strcmp(fn_fa, "-") != 0 && mf.open_file(fn_fa);
    if (mapped) mf.index(rpool);
    if (FILE *fai = fopen((std::string(fn_fa) + ".fai").c_str(), "r")) {
        char nmbuf[4096]; int ll, d0, d1, d2;
        while (0 < fscanf(fai, "%4095s\t%d\t%d\t%d\t%d", nmbuf, &ll, &d0, &d1, &d2)) { tab_names.push_back(nmbuf); tab_lens.push_back(ll); }
        fclose(fai);
        streaming = true;
    } else if (mapped) {
        for (const FastaRecord &r : mf.rec) {
            std::vector<uint8_t> seq; size_t L = 0; int64_t len = 0;
            bool done = false;
            if (mf.geometry(r, &L, &len)) { seq.resize((size_t)len); done = mf.fill(r, L, len, seq.data(), rpool); }
            if (!done) mf.parse_generic(r, seq);
            held.emplace_back(r.name, std::move(seq));
        }
        for (auto &r : held) { tab_names.push_back(r.first); tab_lens.push_back((int64_t)r.second.size()); }
    } else {
        if (!read_fasta(fn_fa, [&](const std::string &nm, std::vector<uint8_t> &seq) { held.emplace_back(nm, std::move(seq)); seq = std::vector<uint8_t>(); return true; })) return give_up(1);
        for (auto &r : held) { tab_names.push_back(r.first); tab_lens.push_back((int64_t)r.second.size()); }
    }
    const double t_fasta = now_s();
    uint64_t tot_len = 0;
    for (size_t i = 0; i < tab_names.size(); ++i) { fprintf(stderr, "[dwgsim_core] %s length: %d\n", tab_names[i].c_str(), (int)tab_lens[i]); tot_len += (uint64_t)tab_lens[i]; }
    fprintf(stderr, "[dwgsim_core] %d sequences, total length: %llu\n", (int)tab_names.size(), (unsigned long long)tot_len);

    const bool has_bfast = want_reads && o.reads_output_type != 1, has_bwa = want_reads && o.reads_output_type != 2;
    if (const char *e = getenv("DWGSIM_HIP_SINK")) { fs.null_sink = !strcmp(e, "null"); fs.memcpy_sink = !strcmp(e, "memcpy"); }
    std::string p = out_prefix;
    if (want_mut) {
        fs.fp_txt = fopen((p + ".mutations.txt").c_str(), "w"); fs.fp_vcf = fopen((p + ".mutations.vcf").c_str(), "w");
        if (!fs.fp_txt || !fs.fp_vcf) { fprintf(stderr, "[main] fail to open mutation files for '%s'. Abort!\n", out_prefix); return give_up(1); }
        fprintf(fs.fp_vcf, "##fileformat=VCFv4.1\n");
        for (size_t i = 0; i < tab_names.size(); ++i) fprintf(fs.fp_vcf, "##contig=<ID=%s,length=%d>\n", tab_names[i].c_str(), (int)tab_lens[i]);
        fprintf(fs.fp_vcf, "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n"
                           "##INFO=<ID=pl,Number=1,Type=Integer,Description=\"Phasing: 1 - HET contig 1, #2 - HET contig #2, 3 - HOM both contigs\">\n"
                           "##INFO=<ID=mt,Number=1,Type=String,Description=\"Variant Type: SUBSTITUTE/INSERT/DELETE\">\n"
                           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");
    }
    if (has_bwa) { fs.fgz[0] = fopen((p + ".bwa.read1.fastq.gz").c_str(), "wb"); fs.fgz[1] = fopen((p + ".bwa.read2.fastq.gz").c_str(), "wb"); if (!fs.fgz[0] || !fs.fgz[1]) { fprintf(stderr, "fail to open FASTQ outputs\n"); return give_up(1); } }
    if (has_bfast) { fs.fgz[2] = fopen((p + ".bfast.fastq.gz").c_str(), "wb"); if (!fs.fgz[2]) { fprintf(stderr, "fail to open FASTQ outputs\n"); return give_up(1); } }
    for (int s = 0; s < 3; ++s) if (fs.fgz[s]) setvbuf(fs.fgz[s], nullptr, _IONBF, 0);      // deliveries are megabytes: no second copy through stdio
    std::unique_ptr<DeflatePool> pool;
    if (!gpu_gzip && want_reads) { pool = std::make_unique<DeflatePool>(nthreads, gz_level); fs.pool = pool.get(); }

    dwgsim_hip_job_t *job = job_made.get();
    if (!job) { fprintf(stderr, "dwgsim-hip: cannot set the job up (error %d)\n", job_err); return 1; }
    int rc = 0;
    auto job_error = [&]() { const char *e = dwgsim_hip_job_last_error(job); if (rc == 0) fprintf(stderr, "%s%s", e, (e[0] && e[strlen(e) - 1] != '\n') ? "\n" : ""); rc = 1; };
    {
        std::vector<const char *> nm; for (auto &s : tab_names) nm.push_back(s.c_str());
        if (dwgsim_hip_job_set_contig_table(job, nm.data(), tab_lens.data(), (int)nm.size()) < 0) job_error();
        if (rc == 0 && !regions_fn.empty() && dwgsim_hip_job_set_regions(job, regions_fn.c_str()) < 0) job_error();             // dwgsim.c:499-506
        if (rc == 0 && muts_type >= 0 && dwgsim_hip_job_set_mutation_input(job, muts_type, muts_fn.c_str()) < 0) job_error();   // dwgsim.c:494-497
        if (rc == 0 && dwgsim_hip_job_prepare(job, nullptr) < 0) job_error();
    }
    const double t_ctx = now_s();
    auto taken = [&](int64_t r) -> bool { if (r < 0 && !DWGSIM_HIP_IS_SKIP(r)) { job_error(); return false; } return true; };      // scheduled or skipped (a note was printed): go on; an error: stop feeding
    auto feed = [&](const std::string &nm, std::vector<uint8_t> &seq) -> bool {      // a sequence that sits in memory: into the staging with all cores
        int64_t st = 0;
        uint8_t *dst = dwgsim_hip_job_begin_contig(job, nm.c_str(), (int64_t)seq.size(), &st);
        if (st < 0 || !dst) { if (st >= 0) st = DWGSIM_HIP_ERR_STATE; return taken(st); }      // the status decides; no place without an error is an error too
        parallel_copy(dst, seq.data(), seq.size(), rpool);
        return taken(dwgsim_hip_job_commit_contig(job));
    };
    if (rc == 0) {
        if (streaming && mapped) {
            for (const FastaRecord &r : mf.rec) {      // every record parsed where the upload will read it
                size_t L = 0; int64_t len = 0; bool done = false, go = true;
                if (mf.geometry(r, &L, &len)) {
                    int64_t st = 0;
                    uint8_t *dst = dwgsim_hip_job_begin_contig(job, r.name.c_str(), len, &st);
                    if (st < 0 || !dst) { if (st >= 0 || !taken(st)) rc = 1; break; }
                    if (mf.fill(r, L, len, dst, rpool)) { done = true; go = taken(dwgsim_hip_job_commit_contig(job)); mf.release(r); }
                    else (void)dwgsim_hip_job_cancel_contig(job);
                }
                if (!done) { std::vector<uint8_t> seq; mf.parse_generic(r, seq); go = feed(r.name, seq); }
                if (!go) break;
            }
        }
        else if (streaming) { if (!read_fasta(fn_fa, feed, [&](size_t k) -> int64_t { return k < tab_lens.size() ? tab_lens[k] : 0; }) && rc == 0) rc = 1; }
        else for (auto &r : held) { if (!feed(r.first, r.second)) break; std::vector<uint8_t>().swap(r.second); }
    }
    const double t_fed = now_s();
    if (dwgsim_hip_job_finish(job) < 0) job_error();
    const double t_out_done = now_s();
    if (rc == 0) fprintf(stderr, "\n[dwgsim_core] Complete!\n");
    if (timing) fprintf(stderr, "[dwgsim-hip] contig table%s %.2f s | contexts, files, inputs %.2f s | %scontigs handed to the GPUs %.2f s | remaining simulate + copy + write %.2f s | "
                                "total %.2f s; text %.2f GB -> gz %.2f GB, %s\n",
                        streaming ? " (.fai)" : " (FASTA read into memory)", t_fasta - t_start, t_ctx - t_fasta, streaming ? "FASTA read, " : "", t_fed - t_ctx, t_out_done - t_fed, t_out_done - t_start,
                        fs.bytes_in.load() / 1e9, fs.bytes_out.load() / 1e9,
                        gpu_gzip ? "gzip members made on the GPU" : (std::string("zlib level ") + std::to_string(gz_level) + " on " + std::to_string(nthreads) + " host threads").c_str());
    if (timing && fs.memcpy_sink) for (int st = 0; st < 3; ++st) if (fs.sink_bytes[st].load())
        fprintf(stderr, "[dwgsim-hip-sink] stream %d: %.2f GB copied by its delivery thread(s) in %.3f thread-seconds inside the sink = %.1f GB/s per thread while delivering (the run took %.2f s)\n", st, fs.sink_bytes[st].load() / 1e9,
                fs.sink_ns[st].load() / 1e9, fs.sink_bytes[st].load() / (double)std::max<uint64_t>(fs.sink_ns[st].load(), 1), t_out_done - t_start);
    if (timing) { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); const double e = ts.tv_sec + ts.tv_nsec * 1e-9; fprintf(stderr, "[dwgsim-hip-clock] main entered at %.3f, output complete at %.3f (seconds since the epoch)\n", e - (t_out_done - t_start) - (now_s() - t_out_done), e - (now_s() - t_out_done)); }
    if (fs.fp_txt) fclose(fs.fp_txt);
    if (fs.fp_vcf) fclose(fs.fp_vcf);
    for (int s = 0; s < 3; ++s) close_gz(fs.fgz[s], gz_level, fs.written[s].load());
    if (getenv("DWGSIM_HIP_TEARDOWN")) { dwgsim_hip_job_destroy(job); return rc; }
    // everything has been delivered and the files are closed: the process ends here.  Handing back device memory, page-locked buffers and the
    // runtime piece by piece costs 0.2 s after a chromosome-sized job; the driver reclaims them with the process.
    fflush(nullptr);
    _exit(rc);
}
