/*
 * dwgsim_hip.h -- C-ABI of the MI355X-native dwgsim hot path (libdwgsim_hip.so).
 *
 * The reference (nh13/DWGSIM) has no plugin/FFI layer; its seam is the single call
 * dwgsim_core(dwgsim_opt_t*) (src/dwgsim.c:419, called at :1163) and inside it
 * mut_diref() (src/mut.c:591, called at dwgsim.c:629) followed by the inlined per-pair loop
 * (dwgsim.c:636-1099).  The entry points below are what a host-side binding for that seam
 * binds: plain pointers and sizes, no C++/torch types, error codes
 * instead of exit(1) (the reference's messages are available through dwgsim_hip_last_error).
 *
 * Random numbers: Philox4x32-10 keyed by (seed, contig) and counter
 * (index, retry, domain|attempt, block) -- see DESIGN.md "RNG layout".  Any read-index range
 * of any contig can be generated independently and in any order.
 *
 * Threading: a context is single-owner; different contexts (one per GPU) may be used
 * concurrently.  There is no global state.
 */
#ifndef DWGSIM_HIP_H
#define DWGSIM_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWGSIM_HIP_ABI_VERSION 5

/* error codes (negative) */
#define DWGSIM_HIP_OK            0
#define DWGSIM_HIP_ERR_ARG      -1   /* bad argument / option out of range (dwgsim_opt.c:307-371) */
#define DWGSIM_HIP_ERR_DEVICE   -2   /* no usable HIP device / HIP runtime error */
#define DWGSIM_HIP_ERR_NOMEM    -3
#define DWGSIM_HIP_ERR_UNSUP    -4   /* option outside the accelerated path (see DESIGN.md "out of scope") */
#define DWGSIM_HIP_ERR_FAILED   -5   /* "failed to generate a read after %d trials" (dwgsim.c:837-840) */
#define DWGSIM_HIP_SLOTS 4          /* output sets of a context: batches that can be in flight (kernels | copy-out issued, twice | copy-out landing) */
#define DWGSIM_HIP_ERR_STATE    -6   /* call order violated (e.g. simulate before mutate) */

/* Why dwgsim_core() passes over a contig (dwgsim.c:539-625, the "[dwgsim_core] #k skip sequence" notes): returned in place of a pair count by
 * dwgsim_hip_pairs_for_contig, dwgsim_hip_contig_region_length and dwgsim_hip_job_add_contig.  A range of their own, disjoint from the error
 * codes above: a skipped contig is not a failure, and a failure must never be mistaken for a skipped contig. */
#define DWGSIM_HIP_SKIP_NO_REGION    -100  /* #0: the contig is not in the targeted region (-x) */
#define DWGSIM_HIP_SKIP_NON_ACGT     -101  /* #1: more than 95 % of its targeted bases are non-ACGT */
#define DWGSIM_HIP_SKIP_AMPLICON     -102  /* #2: shorter than the read length (-a) */
#define DWGSIM_HIP_SKIP_SHORT_INSERT -103  /* #3: shorter than dist + 3 std_dev */
#define DWGSIM_HIP_SKIP_SHORT_READ   -104  /* #4: shorter than a read */
#define DWGSIM_HIP_SKIP_NO_PAIRS     -105  /* #5: a negative pair count */
#define DWGSIM_HIP_IS_SKIP(r) ((r) <= -100 && (r) >= -105)

/* POD mirror of the dwgsim_opt_t fields read on the path (src/dwgsim_opt.h:21-60);
 * defaults are those of dwgsim_opt_init() (src/dwgsim_opt.c:40-80). */
typedef struct dwgsim_hip_params {
    double  e_start[2], e_end[2];   /* -e / -E  per-base error rate ramp (start..end) of read 1 / 2   */
    int32_t is_inner;               /* -i */
    int32_t dist;                   /* -d */
    double  std_dev;                /* -s */
    int64_t N;                      /* -N (-1: use C) */
    double  C;                      /* -C (-1: use N) */
    int32_t length[2];              /* -1 / -2 */
    double  mut_rate;               /* -r */
    double  mut_freq;               /* -F */
    double  indel_frac;             /* -R */
    double  indel_extend;           /* -X */
    int32_t indel_min;              /* -I */
    double  rand_read;              /* -y */
    int32_t max_n;                  /* -n */
    int32_t data_type;              /* -c 0 Illumina, 1 SOLiD, 2 Ion Torrent */
    int32_t strandedness;           /* -S */
    int32_t read_one_strand;        /* -A */
    int32_t is_hap;                 /* -H */
    int32_t seed;                   /* -z (must be >= 0 here: the CLI resolves -1 to time(0)) */
    int32_t fixed_quality;          /* -q as a character code, or -1 */
    double  quality_std;            /* -Q */
    int32_t reads_output_type;      /* -o 0 all, 1 bwa only, 2 bfast only */
    int32_t output_type;            /* -M 0 all, 1 reads only, 2 mutations only */
    int32_t amplicons;              /* -a */
    const char *read_prefix;        /* -P or NULL */
    const char *flow_order;         /* -f or NULL */
    int32_t use_base_error;         /* -B */
} dwgsim_hip_params_t;

typedef struct dwgsim_hip_ctx dwgsim_hip_ctx_t;

/* output streams of one simulate() batch */
enum { DWGSIM_HIP_STREAM_BWA1 = 0, DWGSIM_HIP_STREAM_BWA2 = 1, DWGSIM_HIP_STREAM_BFAST = 2 };

typedef struct dwgsim_hip_batch {
    uint64_t n_pairs;          /* pairs generated (== requested) */
    uint64_t n_random;         /* how many of them are random reads ("rand_ii" increment, dwgsim.c:1096) */
    uint64_t n_retries;        /* rejected attempts (N filter / walk off the contig, dwgsim.c:833-842) */
    uint64_t bytes[3];         /* finished FASTQ text bytes per stream (0 if the stream is disabled) */
    const void *dev_ptr[3];    /* device addresses of the packed text (valid until the slot is reused) */
    float    kernel_ms;        /* HIP-event time of the batch's kernels on the context's stream (simulate_pairs, the abort-rule epilogue, the gzip kernels) */
    float    sim_kernel_ms;    /* ... of the dominant kernel (simulate_pairs) alone */
    /* The reference's abort rule (one counter of failed attempts over the pairs of a contig, reset by every genomic read, fatal above
     * 10 000: dwgsim.c:635, :833-843) as a mergeable summary of THIS batch alone: {fails before its first reset (all of them if it has
     * none), fails after its last reset, has a reset, a run between two of its resets passed the limit}.  A host that shards one contig
     * over several contexts joins the summaries in read-index order (dwgsim_hip_failseg_join) to get the exact verdict. */
    uint64_t fail_seg[4];
    uint64_t fail_carry;       /* the counter after this batch, given the carry it started from */
    uint64_t gz_bytes[3];      /* with dwgsim_hip_set_gzip(ctx, 1): bytes of the .gz form of each stream (a sequence of complete gzip members) */
} dwgsim_hip_batch_t;

/* rand_base value meaning "continue the running count after the previous batch of this context" (kept on the device, so successive
 * batches can be enqueued without a host round trip) */
#define DWGSIM_HIP_RAND_CHAIN UINT64_MAX

/* dwgsim_opt_init() defaults (dwgsim_opt.c:40-80) */
void dwgsim_hip_params_default(dwgsim_hip_params_t *p);

/* Range/consistency checks of dwgsim_opt_parse() (dwgsim_opt.c:307-371, :396-413, :463-469).
 * Returns DWGSIM_HIP_OK or DWGSIM_HIP_ERR_ARG / _UNSUP; msg (optional, cap bytes) receives the
 * reference's message text. */
int dwgsim_hip_params_check(const dwgsim_hip_params_t *p, char *msg, size_t cap);

/* Pairs to simulate on a contig of length l: dwgsim.c:535-537, :582-590 and the skip rules
 * #2-#5 (:595-623).  Returns n_pairs (>= 0), or DWGSIM_HIP_SKIP_AMPLICON / _SHORT_INSERT / _SHORT_READ / _NO_PAIRS for skip rule #2 .. #5
 * (skipped contigs get no mutations either, dwgsim.c:605-611). */
int64_t dwgsim_hip_pairs_for_contig(const dwgsim_hip_params_t *p, int64_t l, uint64_t tot_len,
                                    int is_last_contig, int64_t n_sim_so_far);

/* Replaces the allocation/teardown half of dwgsim_core (dwgsim.c:442-453, :1103-1120).
 * device = HIP device ordinal.  Fails (NULL, *err set) when no GPU is present. */
dwgsim_hip_ctx_t *dwgsim_hip_create(const dwgsim_hip_params_t *p, int device, int *err);
void dwgsim_hip_destroy(dwgsim_hip_ctx_t *ctx);
/* The parameters the context works with: those it was created from, with the error rates -B calibrated (dwgsim_opt.c:452-454: e.start = e.end =
 * the scaled rate).  read_prefix / flow_order come back NULL (the context keeps its own copies). */
int dwgsim_hip_get_params(const dwgsim_hip_ctx_t *ctx, dwgsim_hip_params_t *out);
const char *dwgsim_hip_last_error(const dwgsim_hip_ctx_t *ctx);

/* Replaces seq_read_fasta()'s result + nst_nt4_table lookup (mut.c:49-87, dwgsim.c:56-73):
 * ascii[0..len) are the contig's sequence characters (caller keeps ownership); the packed bases
 * stay resident in HBM.  contig_index = 0-based ordinal of the contig in the FASTA (RNG key).
 * Returns a handle >= 0 or an error code. */
int dwgsim_hip_add_contig(dwgsim_hip_ctx_t *ctx, const char *name, const uint8_t *ascii, int64_t len,
                          uint32_t contig_index);
/* Releases the contig's group.  Every batch that reads the group must have been waited for (dwgsim_hip_wait): DWGSIM_HIP_ERR_STATE otherwise. */
int dwgsim_hip_drop_contig(dwgsim_hip_ctx_t *ctx, int contig);

/* Several contigs at once -- a GROUP.  The reference's contig loop (dwgsim.c:519-625) costs nothing per contig beyond a calloc; here a contig
 * that is added alone costs a chain of walk kernels, a launch and a host synchronisation of its own, which a scaffold-level assembly (10^3-10^5
 * contigs) cannot afford.  The contigs of one call are resident together: dwgsim_hip_mutate_contig on any of them walks all of them with ONE
 * chain of kernels, and read-index ranges of several of them can be simulated by ONE launch (dwgsim_hip_simulate_ranges_async).  Returns the
 * handle of the first contig; contig k of the call has handle + k.  dwgsim_hip_drop_contig on any of them releases the whole group.
 * Upload: if the n buffers already are the group layout inside one page-locked allocation (dwgsim_hip_host_alloc; ascii[k] == ascii[0] +
 * starts[k] of dwgsim_hip_group_layout, zero bytes between the contigs) the copy is asynchronous and the buffers must stay unchanged until
 * dwgsim_hip_mutate_wait returned for the group; any other buffers may be released when the call returns. */
int dwgsim_hip_add_contigs(dwgsim_hip_ctx_t *ctx, int n, const char *const *names, const uint8_t *const *ascii, const int64_t *lens,
                           const uint32_t *contig_index);
/* Where the contigs of a group lie in its coordinate space: starts[k] (optional) = first byte of contig k; returns the total size. */
int64_t dwgsim_hip_group_layout(const int64_t *lens, int n, int64_t *starts);

/* Replaces regions_bed_init() (src/regions_bed.c:38-125, dwgsim.c:499-506): target regions (-x).  names/lens as for
 * dwgsim_hip_set_mutation_input.  *total_len receives the summed region length (the reference's tot_len, dwgsim.c:502-505).
 * Call before add_contig. */
int dwgsim_hip_set_regions(dwgsim_hip_ctx_t *ctx, const char *path, const char *const *names, const int64_t *lens,
                           int n_contigs, uint64_t *total_len);

/* dwgsim.c:539-581: the region length of a contig -- the `l` the reference then uses for pairs-per-contig, the skip rules
 * and fragment placement -- or DWGSIM_HIP_SKIP_NO_REGION (skip #0) / DWGSIM_HIP_SKIP_NON_ACGT (skip #1: > 95 % non-ACGT).  non_acgt / region_bases
 * (optional) receive the two numbers the reference prints with skip #1 (num_n and the region length, dwgsim.c:574-576). */
int64_t dwgsim_hip_contig_region_length(dwgsim_hip_ctx_t *ctx, uint32_t contig_index, const uint8_t *ascii, int64_t len, int64_t *non_acgt, int64_t *region_bases);

/* The `l` that sizes fragment placement on this contig (dwgsim.c:659-671): defaults to the contig length, or to its region
 * length once regions are set.  The reference's last contig in -N mode keeps the full length (dwgsim.c:535-537 bypasses the
 * region bookkeeping): a caller mirroring that passes len here. */
int dwgsim_hip_contig_set_placement_length(dwgsim_hip_ctx_t *ctx, int contig, int64_t l);

/* Replaces muts_input_init() (src/mut_input.c:47-67, called at dwgsim.c:494-497): read a mutation file that then drives
 * mutate_contig instead of the random walk (mut.c:644-745).  type: 0 = bed (-b), 1 = txt (-m), 2 = vcf (-v).
 * names/lens = every contig of the FASTA in file order (the reference's contigs_add table, dwgsim.c:474-476). */
int dwgsim_hip_set_mutation_input(dwgsim_hip_ctx_t *ctx, int type, const char *path, const char *const *names,
                                  const int64_t *lens, int n_contigs);

/* Replaces mut_diref() random branch + mut_left_justify() (mut.c:591-643, :481-589): builds the
 * two mutated haplotypes of the contig in HBM. */
int dwgsim_hip_mutate_contig(dwgsim_hip_ctx_t *ctx, int contig);
/* The same in two halves (the walk runs on a stream of its own: a group can be uploaded and walked while batches of another group are being
 * simulated): mutate_async enqueues, mutate_wait blocks until the haplotypes are finished and reports the walk's errors.  `contig` = any
 * contig of the group; the whole group is walked. */
int dwgsim_hip_mutate_async(dwgsim_hip_ctx_t *ctx, int contig);
int dwgsim_hip_mutate_wait(dwgsim_hip_ctx_t *ctx, int contig);
/* 1: the enqueued walk has finished on the device (mutate_wait will not block; a capacity re-run, rare, may still follow inside it), 0: not yet */
int dwgsim_hip_mutate_poll(dwgsim_hip_ctx_t *ctx, int contig);

/* Replaces mut_print() (mut.c:781-893): the mutations.txt and mutations.vcf BODY lines of this
 * contig.  Buffers are owned by the context and valid until the next call for any contig. */
int dwgsim_hip_mutations_text(dwgsim_hip_ctx_t *ctx, int contig, const char **txt, size_t *txt_len,
                              const char **vcf, size_t *vcf_len);
/* mut_print() in two halves, for a caller that keeps the device busy meanwhile (the job level does): `take` fetches the list of mutated cells of
 * the contig's whole GROUP (device work, on the calling thread) and returns it as an object of its own (NULL: see last_error; *n_contigs =
 * contigs of the group, in the order they were added); `mutlist_text` makes the mutations.txt / .vcf body lines of the group's k-th contig
 * from it -- it touches neither the context nor the device, so any thread may call it while the context simulates the group's reads.  The
 * buffers belong to the list and are valid until the next call on the same list. */
typedef struct dwgsim_hip_mutlist dwgsim_hip_mutlist_t;
dwgsim_hip_mutlist_t *dwgsim_hip_mutations_take(dwgsim_hip_ctx_t *ctx, int contig, int *n_contigs);
int dwgsim_hip_mutlist_text(dwgsim_hip_mutlist_t *list, int k, const char **txt, size_t *txt_len, const char **vcf, size_t *vcf_len);
void dwgsim_hip_mutlist_free(dwgsim_hip_mutlist_t *list);

/* Number of random reads among pairs [first_ii, first_ii + n_pairs) of the contig (for sharding:
 * rand_ii is a running count over all earlier pairs, dwgsim.c:1042,1096). */
int dwgsim_hip_count_random(dwgsim_hip_ctx_t *ctx, int contig, uint64_t first_ii, uint64_t n_pairs,
                            uint64_t *n_random);

/* A read-index range of one contig.  A call that takes several of them covers them in the order given (= file order); they must belong to
 * contigs of one group. */
typedef struct dwgsim_hip_range { int32_t contig; int32_t reserved; uint64_t first_ii, n_pairs; } dwgsim_hip_range_t;
/* random reads among the pairs of the ranges: their total, and (per_range != NULL) one count per range */
int dwgsim_hip_count_random_ranges(dwgsim_hip_ctx_t *ctx, const dwgsim_hip_range_t *ranges, int n_ranges, uint64_t *n_random, uint64_t *per_range);

/* Replaces the loop body dwgsim.c:636-1099 for the read-index range [first_ii, first_ii+n_pairs)
 * of one contig.  rand_base = number of random reads emitted before first_ii (over all contigs).
 * slot in 0 .. DWGSIM_HIP_SLOTS - 1: which of the context's output sets to fill.  Blocking form:
 * returns after the kernels completed; FASTQ text stays in HBM (out->dev_ptr) until a fetch copies it out.
 * Replaces, together with the fetch calls, the gzprintf / gzputc stream of dwgsim.c:919-981. */
int dwgsim_hip_simulate(dwgsim_hip_ctx_t *ctx, int contig, uint64_t first_ii, uint64_t n_pairs,
                        uint64_t rand_base, int slot, dwgsim_hip_batch_t *out);

/* The same in two halves, so that several batches can be in flight (one per slot): simulate_async only enqueues -- kernels, the abort-rule
 * epilogue and the read-back of the batch's counters on the context's compute stream -- and returns; wait blocks until that batch has
 * finished, reports its errors and fills *out.  Batch k+1 may be enqueued (other slot, rand_base = DWGSIM_HIP_RAND_CHAIN) before
 * batch k was waited for; a slot is reused only after its wait(), and its kernels wait on the device for any fetch still reading it. */
int dwgsim_hip_simulate_async(dwgsim_hip_ctx_t *ctx, int contig, uint64_t first_ii, uint64_t n_pairs,
                              uint64_t rand_base, int slot);
int dwgsim_hip_wait(dwgsim_hip_ctx_t *ctx, int slot, dwgsim_hip_batch_t *out);
/* ... for several ranges at once: one launch, one contiguous piece of every output stream (the loop dwgsim.c:519-1099 over several
 * contigs).  The abort rule's counter starts from zero wherever a range begins its contig (first_ii == 0), as `int num_failed = 0` does at
 * dwgsim.c:635; rand_base counts the random reads in front of the first range.
 * ONE call takes at most 2^31 blocks of pairs; Illumina / SOLiD: and only as many pairs as leave the sum of their random reads and the bytes of their first output
 * stream in 62 bits together (the kernels' one look-back word: at 2 x 150 bp 2^26 pairs, 50 GB of text per stream) -- beyond either: DWGSIM_HIP_ERR_ARG
 * "too many pairs in one call"; a job is cut into calls long before (dwgsim_hip_job_*: 2^18 pairs each). */
int dwgsim_hip_simulate_ranges_async(dwgsim_hip_ctx_t *ctx, const dwgsim_hip_range_t *ranges, int n_ranges, uint64_t rand_base, int slot);

/* The failure counter carried into the next simulate call (default: the previous batch's counter when the call continues the
 * same contig at the next read index, else 0).  For sharded jobs: the carry out of the preceding shard. */
int dwgsim_hip_set_fail_carry(dwgsim_hip_ctx_t *ctx, uint64_t carry);

/* acc = acc . next in read-index order (both as dwgsim_hip_batch_t::fail_seg).  Returns 1 when the joined run passes the limit
 * (the reference would have aborted), else 0.  Start from {carry, carry, 0, 0}. */
int dwgsim_hip_failseg_join(uint64_t acc[4], const uint64_t next[4]);

/* Contiguous near-equal read-index ranges, in order: shard `rank` of `world` over [0, n_pairs). */
void dwgsim_hip_shard_range(uint64_t n_pairs, int rank, int world, uint64_t *first, uint64_t *n);

/* Page-locked host memory for fetch destinations (hipHostMalloc / hipHostFree for callers that do not link the HIP runtime). */
void *dwgsim_hip_host_alloc(size_t bytes);
void dwgsim_hip_host_free(void *p);

/* Asynchronous copy of one finished stream of a waited-for slot into PAGE-LOCKED host memory, on the context's copy stream (it
 * overlaps with the kernels of the other slot); fetch_wait blocks until every copy enqueued for the slot has landed. */
int dwgsim_hip_fetch_async(dwgsim_hip_ctx_t *ctx, int slot, int stream, void *host_dst, size_t cap);
int dwgsim_hip_fetch_wait(dwgsim_hip_ctx_t *ctx, int slot);

/* gzip on the GPU (replaces the gzopen / gzprintf / gzputc output of dwgsim.c:919-981, :1150-1158): once switched on, every simulate call also
 * leaves each finished stream in HBM as a sequence of complete gzip members (one per 32 KiB of text, dynamic Huffman codes; concatenating the
 * members of successive batches gives a valid .gz whose decompressed bytes are exactly the text), and fetch_gz_async copies THAT to page-locked
 * host memory: half of the bytes cross PCIe and the host only writes them. */
int dwgsim_hip_set_gzip(dwgsim_hip_ctx_t *ctx, int on);
int dwgsim_hip_fetch_gz_async(dwgsim_hip_ctx_t *ctx, int slot, int stream, void *host_dst, size_t cap);

/* Copy one finished stream of a slot to host memory.  A page-locked destination (hipHostMalloc / hipHostRegister) takes one direct
 * hipMemcpyAsync at link speed; pageable memory goes through double-buffered pinned staging inside. */
int dwgsim_hip_fetch(dwgsim_hip_ctx_t *ctx, int slot, int stream, void *host_dst, size_t cap);

/* Library / device info for logs: returns the ABI version; name gets the HIP device name. */
/* The NUMA node of the host the device hangs off (/sys/bus/pci/devices/<bus id>/numa_node), -1 when unknown.  The job level runs each device's worker
 * thread -- which allocates and fills that device's page-locked buffers -- on the cores of this node.  (No counterpart in the reference: it has one thread.) */
int dwgsim_hip_device_numa_node(int device);
int dwgsim_hip_device_info(int device, char *name, size_t cap, int *n_cu, size_t *hbm_bytes);
/* HIP devices visible to the process (0 without a GPU). */
int dwgsim_hip_device_count(void);

/* ------------------------------------------------------------------------------------------------------------------------------------
 * Job level: dwgsim_core() (dwgsim.c:419-1121) as a whole, on any number of GPUs.  The calls above drive ONE device and leave the
 * choreography to the caller; these own it: contig scheduling (dwgsim.c:519-625), grouping of small contigs, upload / walk / simulate /
 * copy-out pipelines of every device, read-index sharding (batch b of a group's pairs belongs to device b mod n; every device walks the
 * group itself; the random-read count in front of a batch and the abort rule's summaries are the only things that cross devices, as host
 * integers -- no collective), and delivery of the output in file order.  A binding at the reference's seam (dwgsim.c:1163) is: create, set the
 * contig table, add every contig of the FASTA in order, finish.
 * ------------------------------------------------------------------------------------------------------------------------------------ */
typedef struct dwgsim_hip_job dwgsim_hip_job_t;

typedef struct dwgsim_hip_job_sink {
    void *user;
    /* mut_print() (mut.c:781-893): the mutations.txt / mutations.vcf body lines of the next contig, contigs in FASTA order.  One call at a
     * time, from a thread of the job.  Non-zero return stops the job. */
    int (*mutations)(void *user, const char *contig, const char *txt, size_t txt_len, const char *vcf, size_t vcf_len);
    /* The gzprintf / gzputc stream of dwgsim.c:919-981: the next piece of output stream `stream` (DWGSIM_HIP_STREAM_*), pieces in file
     * order.  gz != 0: `len` bytes of complete gzip members that hold text_len bytes of text; gz == 0: len == text_len bytes of text.
     * One thread per stream, so the three files can be written side by side; `data` is page-locked memory of the job, reused once the call
     * has returned.  NULL: the reads are simulated and left on the device (benchmarks).  Non-zero return stops the job. */
    int (*reads)(void *user, int stream, const void *data, size_t len, size_t text_len, int gz);
    /* dwgsim_core's stderr lines (skip notes, the running pair count); NULL: written to stderr unless options.quiet */
    void (*message)(void *user, const char *text);
    /* (ABI 5) Instead of reads, for a sink that can take pieces OUT OF ORDER (pwrite): `offset` = where this piece goes in output stream `stream` as it is
     * delivered (the members back to back, or the text).  Called CONCURRENTLY, by one thread per device and stream -- every device hands over what it made
     * itself -- each piece exactly once, in any order; together the pieces tile [0, total) of each stream.  The ordered form has one thread per stream, and
     * a sink that touches the bytes takes 21-26 GB/s from it (measured, memcpy: profiles/r06_solo_rank_entry.txt): less than one device's link delivers, so at
     * N > 1 devices it would be the job's ceiling.  Used when reads_at != NULL (reads is then ignored).  Non-zero return stops the job. */
    int (*reads_at)(void *user, int stream, uint64_t offset, const void *data, size_t len, size_t text_len, int gz);
} dwgsim_hip_job_sink_t;

typedef struct dwgsim_hip_job_options {
    int32_t  gzip;           /* 1: reads() gets gzip members made on the GPU (dwgsim_hip_set_gzip), 0: text */
    int32_t  quiet;
    uint64_t batch_pairs;    /* pairs per launch (0: 2^18) */
    uint64_t group_bp;       /* consecutive contigs are resident, walked and simulated together up to this many bases (0: 32 Mi) */
    uint64_t min_share;      /* a group is spread over fewer devices while a device's share would stay below this many pairs (0: 65536) */
} dwgsim_hip_job_options_t;

/* devices == NULL or n_devices <= 0: every HIP device the process sees.  options == NULL: GPU gzip, defaults. */
dwgsim_hip_job_t *dwgsim_hip_job_create(const dwgsim_hip_params_t *p, const int *devices, int n_devices, const dwgsim_hip_job_sink_t *sink,
                                        const dwgsim_hip_job_options_t *options, int *err);
/* contigs_add() (dwgsim.c:465-478): every contig of the FASTA (or of its .fai) -- fixes tot_len and n_ref, and is what -m/-b/-v/-x files are
 * checked against.  Then, optionally, regions_bed_init() / muts_input_init() (dwgsim.c:494-506). */
int dwgsim_hip_job_set_contig_table(dwgsim_hip_job_t *job, const char *const *names, const int64_t *lens, int n);
int dwgsim_hip_job_set_regions(dwgsim_hip_job_t *job, const char *path);
int dwgsim_hip_job_set_mutation_input(dwgsim_hip_job_t *job, int type, const char *path);
/* optional: parse the files above and start the device threads now (errors of -x / -m / -b / -v surface here instead of at the first contig) */
int dwgsim_hip_job_prepare(dwgsim_hip_job_t *job, uint64_t *total_len);
/* The body of the contig loop (dwgsim.c:519-625 and everything below it) for the next contig of the FASTA.  Returns the pairs scheduled for
 * it (>= 0), or why it is skipped (DWGSIM_HIP_SKIP_*: test with DWGSIM_HIP_IS_SKIP), or an error code (DWGSIM_HIP_ERR_*: the job has failed, stop
 * feeding it and call finish / last_error).
 * The sequence is copied: `ascii` is the caller's again when the call returns.  Blocks only when the devices are two groups behind. */
int64_t dwgsim_hip_job_add_contig(dwgsim_hip_job_t *job, const char *name, const uint8_t *ascii, int64_t len);
/* The same in two halves, for a caller that produces the sequence itself (seq_read_fasta, mut.c:49-87, with as many threads as it likes):
 * begin returns where the contig's `len` sequence characters go -- page-locked staging of the job, already in the layout the upload needs, so
 * nothing is copied again -- or NULL with *status = an error code; any threads of the caller fill [p, p + len); commit schedules the contig and
 * returns what dwgsim_hip_job_add_contig returns; cancel hands the reservation back (e.g. the record turned out to have another length).  One
 * contig can be open at a time; the pointer is valid until commit / cancel.  begin blocks only when the devices are two groups behind. */
uint8_t *dwgsim_hip_job_begin_contig(dwgsim_hip_job_t *job, const char *name, int64_t len, int64_t *status);
int64_t dwgsim_hip_job_commit_contig(dwgsim_hip_job_t *job);
int dwgsim_hip_job_cancel_contig(dwgsim_hip_job_t *job);
/* no more contigs: returns when everything has been delivered to the sink (DWGSIM_HIP_OK) or the job failed */
int dwgsim_hip_job_finish(dwgsim_hip_job_t *job);
const char *dwgsim_hip_job_last_error(const dwgsim_hip_job_t *job);
void dwgsim_hip_job_destroy(dwgsim_hip_job_t *job);

/* ====================================================================================================================================
 * TEST / ANALYSIS HOOKS -- everything below this line is NOT part of the drop-in surface.  A binding of the reference needs none of it; the
 * parity tests and the profiling scripts do.  (All of it is read-only with respect to what the product computes: the options select between
 * code paths that produce the same bytes, the self-tests and counters only report.)
 * ==================================================================================================================================== */
/* "justify_seq" = 1: left-justification from one thread (cross-check of the cluster-parallel form); "walk_cap" = n: start the mutation walk with
 * room for n candidates (exercises the exact re-run); "walk_seg_min" = n: segmented form of the walk's serial scans from n candidates on;
 * "writer" = 0 / 1: force the register / LDS-FIFO record writer; "sim_threads" = 64 / 256: force the one-wave blocks of the long-read variant / the 256-lane blocks (where their reads fit LDS);
 * "place_cap" = n: room for n undecided pairs per list in dwgsim_hip_count_random* (exercises its second run); "split" = 0 / 1: the Illumina
 * read kernel as one kernel with look-backs / as two kernels with the offsets computed in between (default: two for reads of up to 100 bases); "flow_slots" = n:
 * n scratch slots per XCD for the Ion Torrent read buffers / the long reads of the one-wave blocks (blocks wait for slots); "flow_cap" = n: the Ion Torrent
 * read capacity a job starts from (a read that outgrows it makes the batch run again with twice the room); "phases" = 1: print the phase
 * split of the -DDW_PHASE_TIMING analysis build. */
int dwgsim_hip_debug_option(dwgsim_hip_ctx_t *ctx, const char *key, int64_t value);
/* "place_open": pairs the last dwgsim_hip_count_random* call could not settle from the coarse haplotype summaries; "flow_cap_mult": how often (as a
 * power of two) the Ion Torrent read capacity has been doubled so far; "walk_us" / "count_us":
 * HIP-event time (microseconds, accumulated) of the context's walk chains / random-read counts on the walk stream */
int dwgsim_hip_debug_get(dwgsim_hip_ctx_t *ctx, const char *key, int64_t *value);
/* the gzip kernel on arbitrary host bytes (the product only ever feeds it FASTQ text) */
int dwgsim_hip_debug_gzip(dwgsim_hip_ctx_t *ctx, const void *text, size_t n, void *out, size_t cap, size_t *out_n);
/* occurrences of `byte` in one finished stream of a slot, counted on the device (whole-output checks without a copy-out) */
int dwgsim_hip_debug_count_byte(dwgsim_hip_ctx_t *ctx, int slot, int stream, int byte, uint64_t *count);
/* device self-tests of the arithmetic shortcuts (dw_common.hpp / dw_simulate.hip): the range-restricted fp64 division / sqrt / log against the
 * compiler's general forms on n operand sets; the lazy fp32 quality normals against the exact fp64 form on n tries from `first` (n = 2^32: every
 * try) and, with exhaustive != 0, the hardware log2 / rcp / sqrt on every float of their operand ranges.  out: see dw_host.cpp. */
int dwgsim_hip_selftest_fp64(int device, uint32_t seed, uint64_t n, uint64_t *out);
int dwgsim_hip_selftest_lazy(int device, uint32_t first, uint64_t n, double sigma, int exhaustive, uint64_t *out);
/* the number formatters of the name line (decimal positions and counts, the hexadecimal read index: dw_read.hpp put_dec / put_hex) against one
 * division per digit on the values first + i * stride, i < n.  out[0] / out[1]: decimal / hexadecimal texts that differ, out[2]: values compared. */
int dwgsim_hip_selftest_text(int device, uint64_t first, uint64_t n, uint64_t stride, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
