// dwgsim_cli.cpp -- `dwgsim-hip`: the dwgsim command line (reference src/dwgsim.c:1123-1184 main(),
// src/dwgsim_opt.c:204-472 option surface) driving the MI355X hot path through the C-ABI of
// include/dwgsim_hip.h.  Same usage, same defaults, same output file names:
//     dwgsim-hip [options] <in.ref.fa> <out.prefix>
//     <prefix>.mutations.txt / .mutations.vcf / .bfast.fastq.gz / .bwa.read1.fastq.gz / .bwa.read2.fastq.gz
// Host-only work here: option parsing, FASTA reading (mut.c:49-87), contig scheduling, file I/O and
// gzip (multi-member gzip: the reference's own test compares decompressed bytes, testdata/test.sh:23-25).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <time.h>
#include <unistd.h>
#include <limits.h>
#include <zlib.h>
#include <string>
#include <vector>
#include <thread>
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

static bool read_fasta(const char *fn, Fasta &fa)     // mut.c:49-87 seq_read_fasta
{
    FILE *fp = strcmp(fn, "-") ? fopen(fn, "r") : stdin;
    if (!fp) { fprintf(stderr, "[dwgsim_core] fail to open file '%s'. Abort!\n", fn); return false; }
    std::vector<char> buf(1 << 22);
    std::string name; std::vector<uint8_t> seq; int state = 0;   // 0 before first '>', 1 in name, 2 rest of header line, 3 sequence
    bool have = false;
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), fp)) > 0) {
        for (size_t i = 0; i < n; ++i) {
            const int c = (unsigned char)buf[i];
            if (state == 0) { if (c == '>') { state = 1; name.clear(); seq.clear(); have = true; } }
            else if (state == 1) { if (c == ' ' || c == '\t') state = 2; else if (c == '\n') state = 3; else if (c != '\r') name.push_back((char)c); }
            else if (state == 2) { if (c == '\n') state = 3; }
            else {
                if (c == '>') { fa.names.push_back(name); fa.seqs.push_back(seq); state = 1; name.clear(); seq.clear(); }
                else if (isalpha(c) || c == '-' || c == '.') seq.push_back((uint8_t)c);
            }
        }
    }
    if (have) { fa.names.push_back(name); fa.seqs.push_back(seq); }
    if (fp != stdin) fclose(fp);
    return true;
}

// gzip writer: every batch of FASTQ text is cut into 4 MiB chunks, each chunk is deflated as an independent
// gzip member by a pool of host threads, and the members are written in order.  A multi-member .gz
// decompresses to exactly the concatenated text (the reference's own test compares decompressed bytes,
// testdata/test.sh:23-25); the reference itself feeds zlib one byte at a time (dwgsim.c:930-931), which
// is ~80 % of its wall time.
struct GzOut {
    FILE *f = nullptr;
    bool open(const std::string &fn) { f = fopen(fn.c_str(), "wb"); return f != nullptr; }
    static bool deflate_member(const char *src, size_t n, std::vector<unsigned char> &out)
    {
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        out.resize(deflateBound(&zs, (uLong)n) + 64);
        zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n; zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
        const int rc = deflate(&zs, Z_FINISH);
        out.resize(zs.total_out);
        deflateEnd(&zs);
        return rc == Z_STREAM_END;
    }
    bool write(const char *p, size_t n, unsigned nthreads)
    {
        const size_t CH = (size_t)4 << 20;
        const size_t nch = (n + CH - 1) / CH;
        std::vector<std::vector<unsigned char>> parts(nch);
        std::vector<char> ok(nch, 1);
        if (nthreads < 1) nthreads = 1;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads && t < nch; ++t)
            th.emplace_back([&, t]() { for (size_t k = t; k < nch; k += nthreads) ok[k] = deflate_member(p + k * CH, (k + 1) * CH <= n ? CH : n - k * CH, parts[k]) ? 1 : 0; });
        for (auto &x : th) x.join();
        for (size_t k = 0; k < nch; ++k) { if (!ok[k]) return false; if (fwrite(parts[k].data(), 1, parts[k].size(), f) != parts[k].size()) return false; }
        return true;
    }
    void close()
    {
        if (!f) return;
        if (ftell(f) == 0) { std::vector<unsigned char> e; deflate_member("", 0, e); fwrite(e.data(), 1, e.size(), f); }   // empty stream: still a valid .gz
        fclose(f); f = nullptr;
    }
};

int main(int argc, char **argv)
{
    dwgsim_hip_params_t o; dwgsim_hip_params_default(&o);
    std::string prefix_s, fixedq_s, flow_s, regions_fn, muts_fn; int muts_type = -1, muts_flags = 0;
    int c, device = 0;
    if (const char *d = getenv("DWGSIM_HIP_DEVICE")) device = atoi(d);
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

    const char *fn_fa = argv[optind], *out_prefix = argv[optind + 1];
    Fasta fa;
    if (!read_fasta(fn_fa, fa)) return 1;
    uint64_t tot_len = 0;
    for (size_t i = 0; i < fa.seqs.size(); ++i) { fprintf(stderr, "[dwgsim_core] %s length: %d\n", fa.names[i].c_str(), (int)fa.seqs[i].size()); tot_len += fa.seqs[i].size(); }
    fprintf(stderr, "[dwgsim_core] %d sequences, total length: %llu\n", (int)fa.seqs.size(), (unsigned long long)tot_len);

    const bool want_mut = o.output_type != 1, want_reads = o.output_type != 2;
    const bool has_bfast = want_reads && o.reads_output_type != 1, has_bwa = want_reads && o.reads_output_type != 2;
    FILE *fp_txt = nullptr, *fp_vcf = nullptr; GzOut gz[3];
    std::string p = out_prefix;
    if (want_mut) {
        fp_txt = fopen((p + ".mutations.txt").c_str(), "w"); fp_vcf = fopen((p + ".mutations.vcf").c_str(), "w");
        if (!fp_txt || !fp_vcf) { fprintf(stderr, "[main] fail to open mutation files for '%s'. Abort!\n", out_prefix); return 1; }
        fprintf(fp_vcf, "##fileformat=VCFv4.1\n");
        for (size_t i = 0; i < fa.seqs.size(); ++i) fprintf(fp_vcf, "##contig=<ID=%s,length=%d>\n", fa.names[i].c_str(), (int)fa.seqs[i].size());
        fprintf(fp_vcf, "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n"
                        "##INFO=<ID=pl,Number=1,Type=Integer,Description=\"Phasing: 1 - HET contig 1, #2 - HET contig #2, 3 - HOM both contigs\">\n"
                        "##INFO=<ID=mt,Number=1,Type=String,Description=\"Variant Type: SUBSTITUTE/INSERT/DELETE\">\n"
                        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");
    }
    if (has_bwa && (!gz[0].open(p + ".bwa.read1.fastq.gz") || !gz[1].open(p + ".bwa.read2.fastq.gz"))) { fprintf(stderr, "fail to open FASTQ outputs\n"); return 1; }
    if (has_bfast && !gz[2].open(p + ".bfast.fastq.gz")) { fprintf(stderr, "fail to open FASTQ outputs\n"); return 1; }

    int err = 0;
    dwgsim_hip_ctx_t *ctx = dwgsim_hip_create(&o, device, &err);
    if (!ctx) { fprintf(stderr, "dwgsim-hip: cannot create a GPU context (error %d)\n", err); return 1; }
    if (!regions_fn.empty()) {   // dwgsim.c:499-506
        std::vector<const char *> nm; std::vector<int64_t> ln;
        for (size_t i = 0; i < fa.seqs.size(); ++i) { nm.push_back(fa.names[i].c_str()); ln.push_back((int64_t)fa.seqs[i].size()); }
        if (dwgsim_hip_set_regions(ctx, regions_fn.c_str(), nm.data(), ln.data(), (int)nm.size(), &tot_len) < 0) { fprintf(stderr, "%s", dwgsim_hip_last_error(ctx)); dwgsim_hip_destroy(ctx); return 1; }
    }
    if (muts_type >= 0) {     // dwgsim.c:494-497
        std::vector<const char *> nm; std::vector<int64_t> ln;
        for (size_t i = 0; i < fa.seqs.size(); ++i) { nm.push_back(fa.names[i].c_str()); ln.push_back((int64_t)fa.seqs[i].size()); }
        if (dwgsim_hip_set_mutation_input(ctx, muts_type, muts_fn.c_str(), nm.data(), ln.data(), (int)nm.size()) < 0) { fprintf(stderr, "%s", dwgsim_hip_last_error(ctx)); dwgsim_hip_destroy(ctx); return 1; }
    }
    const uint64_t BATCH = 1u << 22;
    unsigned nthreads = std::thread::hardware_concurrency(); if (nthreads == 0) nthreads = 4;
    if (const char *e = getenv("DWGSIM_HIP_THREADS")) nthreads = (unsigned)atoi(e);
    std::vector<char> host[3];
    int64_t n_sim = 0; uint64_t rand_ii = 0, ctr = 0; int n_ref = (int)fa.seqs.size(), prev_skip = 0, rc = 0;
    for (size_t ci = 0; ci < fa.seqs.size() && rc == 0; ++ci) {
        const int64_t l = (int64_t)fa.seqs[ci].size(); const char *name = fa.names[ci].c_str();
        --n_ref;
        int64_t n_pairs = 0, l_eff = l;
        if (want_reads) {
            const bool last_takes_rest = n_ref == 0 && o.C < 0;     // dwgsim.c:535-537
            if (!regions_fn.empty() && !last_takes_rest) {
                l_eff = dwgsim_hip_contig_region_length(ctx, (uint32_t)ci, fa.seqs[ci].data(), l);
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
        const int cid = dwgsim_hip_add_contig(ctx, name, fa.seqs[ci].data(), l, (uint32_t)ci);
        if (cid >= 0 && !regions_fn.empty()) dwgsim_hip_contig_set_placement_length(ctx, cid, l_eff);
        if (cid < 0 || dwgsim_hip_mutate_contig(ctx, cid) < 0) { fprintf(stderr, "dwgsim-hip: %s\n", dwgsim_hip_last_error(ctx)); rc = 1; break; }
        if (want_mut) {
            const char *t, *v; size_t tl, vl;
            if (dwgsim_hip_mutations_text(ctx, cid, &t, &tl, &v, &vl) < 0) { fprintf(stderr, "dwgsim-hip: %s\n", dwgsim_hip_last_error(ctx)); rc = 1; break; }
            fwrite(t, 1, tl, fp_txt); fwrite(v, 1, vl, fp_vcf);
        }
        for (uint64_t first = 0; want_reads && first < (uint64_t)n_pairs; first += BATCH) {
            const uint64_t n = (uint64_t)n_pairs - first < BATCH ? (uint64_t)n_pairs - first : BATCH;
            dwgsim_hip_batch_t b;
            if (dwgsim_hip_simulate(ctx, cid, first, n, rand_ii, 0, &b) < 0) { fprintf(stderr, "%s", dwgsim_hip_last_error(ctx)); rc = 1; break; }
            for (int s = 0; s < 3; ++s) if (b.bytes[s]) {
                host[s].resize(b.bytes[s]);
                if (dwgsim_hip_fetch(ctx, 0, s, host[s].data(), host[s].size()) < 0) { fprintf(stderr, "dwgsim-hip: %s\n", dwgsim_hip_last_error(ctx)); rc = 1; break; }
            }
            // deflate with all host cores (independent gzip members), write in stream order
            for (int s = 0; s < 3; ++s) if (b.bytes[s] && gz[s].f && !gz[s].write(host[s].data(), host[s].size(), nthreads)) { fprintf(stderr, "dwgsim-hip: writing FASTQ failed\n"); rc = 1; break; }
            rand_ii += b.n_random; n_sim += (int64_t)n; ctr += n;
            fprintf(stderr, "\r[dwgsim_core] %llu", (unsigned long long)ctr);
        }
        dwgsim_hip_drop_contig(ctx, cid);
    }
    fprintf(stderr, "\n[dwgsim_core] Complete!\n");
    dwgsim_hip_destroy(ctx);
    if (fp_txt) fclose(fp_txt);
    if (fp_vcf) fclose(fp_vcf);
    for (int s = 0; s < 3; ++s) gz[s].close();
    return rc;
}
