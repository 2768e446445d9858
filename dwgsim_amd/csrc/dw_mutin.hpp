// dw_mutin.hpp -- user-supplied mutations (-m txt, -b bed, -v vcf): host-side parsing and resolution.
//
// Replaces muts_input_init() (src/mut_input.c:47-67 -> mut_txt.c:40-133, mut_bed.c:37-137,
// mut_vcf.c:42-280) and the file-driven branches of mut_diref() (src/mut.c:644-745).  The entries of
// a file are applied to a contig strictly in file order with read-modify-write semantics on single
// cells (`|=` for deletions, `=` for substitutions, last insertion wins), so the host resolves them
// into the FINAL value of every touched cell plus the insertion payloads; the GPU scatters those
// patches into the resident haplotypes and then left-justifies as usual.  O(#entries) host work.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace dw {

struct ContigName { std::string name; int64_t len; };

struct MutEntry {            // one parsed line (txt / vcf: a single position; bed: a half-open range)
    uint32_t contig = 0;     // FASTA ordinal
    uint32_t pos = 0;        // txt / vcf: 1-based position;  bed: 0-based start
    uint32_t end = 0;        // bed only: end (exclusive)
    uint8_t type = 0;        // T_SUB / T_INS / T_DEL
    uint8_t is_hap = 0;      // txt / vcf: haplotype mask as read (may be 4: applies to neither, mut_vcf.c:93,116-120)
    std::string bases;       // alternative / inserted bases ("*" in bed = random)
};

struct MutInput {
    int type = -1;           // 0 bed, 1 txt, 2 vcf (mut_input.h:29-33)
    std::vector<MutEntry> e;
};

// Returns false and fills `err` with the reference's message on a malformed file.
bool parse_mutation_input(int type, const char *path, const std::vector<ContigName> &contigs, MutInput &out, std::string &err);

struct InsPayload { int32_t pos; std::vector<uint8_t> bases; };       // printed order P[0..n)
struct ResolvedContig {
    std::vector<int32_t> pos;          // touched positions, ascending
    std::vector<uint16_t> cells;       // final cell of haplotype 1 | haplotype 2 << 8
    std::vector<InsPayload> ins[2];    // per haplotype, ascending position
};

// Applies the entries of `contig` to the reference bases `ascii[0..l)`; random decisions (bed ploidy, '*' bases,
// N bases inside given insertions) come from Philox (seed, contig) with the domains of DESIGN.md.
void resolve_mutation_input(const MutInput &in, uint32_t contig, const uint8_t *ascii, int64_t l, uint32_t seed, bool is_hap_mode,
                            ResolvedContig &out);

// ---- target regions (-x): regions_bed_init(), src/regions_bed.c:38-125 ----
struct Regions { std::vector<uint32_t> contig, start, end; };      // sorted, overlapping / touching intervals merged
bool parse_regions(const char *path, const std::vector<ContigName> &contigs, Regions &out, std::string &err);

} // namespace dw
