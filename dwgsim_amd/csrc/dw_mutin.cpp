// dw_mutin.cpp -- see dw_mutin.hpp.  Pure host C++ (no device code).
#include "dw_mutin.hpp"
#include "dw_kernels.hpp"
#include <stdio.h>
#include <string.h>
#include <ctype.h>
#include <map>
#include <algorithm>

namespace dw {
namespace {

uint8_t code_of(int ch)       // dwgsim.c:56-73 nst_nt4_table
{
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

// ---- host copy of the counter RNG (DESIGN.md "RNG layout"): wide uniforms only ----
void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
double wide_uniform(uint32_t seed, uint32_t contig, uint32_t dom, uint64_t idx, uint32_t slot)
{
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)((idx >> 32) & 0xFFFFu), dom << 24, slot >> 1};
    philox4x32_10(c, seed, contig);
    const uint32_t hi = (slot & 1) ? c[2] : c[0], lo = (slot & 1) ? c[3] : c[1];
    return (double)(((uint64_t)hi << 21) | (uint64_t)(lo >> 11)) * 0x1p-53;
}
constexpr uint32_t D_MUTIN = 18, D_MUTIN_BASE = 19;

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
}  // namespace
}  // namespace dw
#include <stdarg.h>
namespace dw {
namespace {
std::string fmt(const char *f, ...)
{
    char b[2048]; va_list ap; va_start(ap, f); vsnprintf(b, sizeof b, f, ap); va_end(ap); return b;
}

// contig lookup shared by the three readers: the cursor only moves forward (files must follow the FASTA order)
struct ContigCursor {
    const std::vector<ContigName> &c; size_t i = 0;
    explicit ContigCursor(const std::vector<ContigName> &cc) : c(cc) {}
    bool seek(const std::string &name, bool *moved) { *moved = false; while (i < c.size() && c[i].name != name) { ++i; *moved = true; } return i < c.size(); }
};

char iupac_to_alt(char iupac, char base)      // dwgsim.c:202-213 iupac_and_base_to_mut
{
    static const char *codes = "XACMGRSVTWYHKDBN";
    const int b = code_of(base);
    for (int i = 0; i < 4; ++i) if (codes[(1 << (b & 3)) | (1 << i)] == iupac) return "ACGTN"[i];
    return 'X';
}
int mutation_type(std::string s)              // dwgsim.c:183-200 get_muttype
{
    for (auto &ch : s) ch = (char)tolower((unsigned char)ch);
    if (s == "snp" || s == "substitute" || s == "sub" || s == "s") return T_SUB;
    if (s == "insertion" || s == "insert" || s == "ins" || s == "i") return T_INS;
    if (s == "deletion" || s == "delet" || s == "del" || s == "d") return T_DEL;
    return -1;
}

bool read_txt(FILE *fp, const std::vector<ContigName> &contigs, MutInput &out, std::string &err)     // mut_txt.c:40-133
{
    ContigCursor cur(contigs);
    char name[1024], mut[1024], ref; uint32_t pos, is_hap;
    while (0 < fscanf(fp, "%1023s\t%u\t%c\t%1023s\t%d", name, &pos, &ref, mut, &is_hap)) {
        bool moved;
        if (!cur.seek(name, &moved)) { err = fmt("Error: mutation contig not found or out of order [%s]\n", name); return false; }
        if (pos <= 0 || contigs[cur.i].len < (int64_t)pos) { err = fmt("Error: start out of range [%s,%u]\n", name, pos); return false; }
        MutEntry e; e.contig = (uint32_t)cur.i; e.pos = pos; e.is_hap = (uint8_t)is_hap;
        const bool ref_gap = ref == '-', alt_gap = mut[0] == '-';
        if (ref_gap && !alt_gap) e.type = T_INS;
        else if (!ref_gap && alt_gap) e.type = T_DEL;
        else if (!ref_gap && !alt_gap) {
            e.type = T_SUB;
            if (is_hap < 3) {                       // heterozygous substitutions are written as IUPAC codes (mut.c:830)
                if (code_of(mut[0]) < 4) { err = "Error: heterozygous bases must be in IUPAC form\n"; return false; }
                mut[0] = iupac_to_alt(mut[0], ref);
                if (mut[0] == 'X') { err = "Error: out of range\n"; return false; }
                mut[1] = 0;
            }
        } else { err = "Error: out of range\n"; return false; }
        e.bases = mut;
        out.e.push_back(e);
    }
    return true;
}

bool read_bed(FILE *fp, const std::vector<ContigName> &contigs, MutInput &out, std::string &err)     // mut_bed.c:37-137
{
    ContigCursor cur(contigs);
    char name[1024], type[1024], bases[1024]; uint32_t start, end, prev_contig = 0, max_end = 0;
    while (0 < fscanf(fp, "%1023s\t%u\t%u\t%1023s\t%1023s", name, &start, &end, bases, type)) {
        bool moved;
        if (!cur.seek(name, &moved)) { err = fmt("Error: contig not found [%s]\n", name); return false; }
        const int64_t len = contigs[cur.i].len;
        if (len <= (int64_t)start) { err = fmt("Error: start out of range [%s,%u]\n", name, start); return false; }
        if (len < (int64_t)end) { err = fmt("Error: end out of range [%s,%u]\n", name, end); return false; }
        if (end <= start) { err = fmt("Error: end <= start [%s,%u,%u]\n", name, start, end); return false; }
        if (strcmp("*", bases) != 0 && (size_t)(end - start) != strlen(bases)) { err = fmt("Error: bases did not match start and end [%s,%u,%u,%s]\n", name, start, end, bases); return false; }
        if (prev_contig == (uint32_t)cur.i && start + 1 <= max_end) { fprintf(stderr, "Warning: overlapping entries, ignoring entry [%s\t%u\t%u\t%s\t%s]\n", name, start, end, bases, type); continue; }
        if (prev_contig != (uint32_t)cur.i || max_end < end) { prev_contig = (uint32_t)cur.i; max_end = end; }
        const int ty = mutation_type(type);
        if (ty == T_INS && end - start > 26) { err = fmt("Error: insertion of length %d exceeded the maximum supported length of %d\n", (int)(end - start), 26); return false; }
        if (ty < 0) { err = fmt("Error: mutation type unrecognized [%s]\n", type); return false; }
        MutEntry e; e.contig = (uint32_t)cur.i; e.pos = start; e.end = end; e.type = (uint8_t)ty; e.bases = bases;
        out.e.push_back(e);
    }
    return true;
}

bool read_vcf(FILE *fp, const std::vector<ContigName> &contigs, MutInput &out, std::string &err)     // mut_vcf.c:42-280
{
    std::string all; { char buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, fp)) > 0) all.append(buf, n); }
    ContigCursor cur(contigs);
    bool warned = false; uint32_t prev_pos = 0;
    size_t s = 0;
    while (s < all.size()) {
        size_t n = all.find_first_of("\n\r", s);
        if (n == std::string::npos) n = all.size();
        if (n == s) { ++s; continue; }
        if (all[s] == '#') { s = n; continue; }
        const std::string line = all.substr(s, n - s);
        char name[1024], id[1024], ref[1024], alt[1025]; uint32_t pos = 0;
        if (EOF == sscanf(line.c_str(), "%1023s\t%u\t%1023s\t%1023s\t%1024s", name, &pos, id, ref, alt)) { err = "Error: VCF parsing error\n"; return false; }
        // ploidy: the first "[\t;]pl=<digit>" of the line; without it the very first such record is homozygous and every
        // later one keeps mask 4, i.e. is applied to neither haplotype (mut_vcf.c:93,116-120)
        uint32_t is_hap = 4;
        for (size_t q = 0; q + 4 < line.size(); ++q)
            if ((line[q] == '\t' || line[q] == ';') && line[q + 1] == 'p' && line[q + 2] == 'l' && line[q + 3] == '=') {
                const char d = line[q + 4];
                if (d < '1' || d > '3') { err = "Error: Could not determine the strand of the mutation from the 'pl' tag.\n"; return false; }
                is_hap = (uint32_t)(d - '0');
                break;
            }
        if (is_hap == 4 && !warned) { fprintf(stderr, "Warning: strand of the mutation not found; please use the 'pl' tag.\n"); warned = true; is_hap = 3; }
        bool moved;
        if (!cur.seek(name, &moved)) { err = fmt("Error: contig not found [%s]\n", name); return false; }
        if (moved) prev_pos = 0;
        if (pos <= 0 || contigs[cur.i].len < (int64_t)pos) { err = fmt("Error: start out of range [%s,%u]\n", name, pos); return false; }
        if (pos < prev_pos) { err = fmt("Error: out of order [%s,%u]\n", name, pos); return false; }
        std::string r = ref, a = alt;
        if (r == ".") r.clear();
        if (a == ".") a.clear();
        // (the reference indexes past the contig here -- undefined behaviour; a library call must not: same message as for a start out of range)
        if (contigs[cur.i].len < (int64_t)pos + (int64_t)r.size() - 1) { err = fmt("Error: start out of range [%s,%u]\n", name, pos + (uint32_t)r.size() - 1); return false; }
        if (r.empty() && a.empty()) { err = "Error: empty alleles\n"; return false; }
        if (a.find(',') != std::string::npos) { err = "Error: multiple alleles are not supported\n"; return false; }
        for (auto *str : {&r, &a}) for (auto &ch : *str) { ch = "ACGTNN"[code_of(ch)]; if (ch == 'N') { err = "Error: non-ACGT base found\n"; return false; } }
        MutEntry e; e.contig = (uint32_t)cur.i; e.is_hap = (uint8_t)is_hap;
        if (r.size() == a.size()) {                                   // SNP / MNP: one substitution per base
            for (size_t j = 0; j < r.size(); ++j) { e.pos = pos + (uint32_t)j; e.type = T_SUB; e.bases = std::string(1, a[j]); out.e.push_back(e); }
        } else if (r.size() < a.size()) {                             // insertion after the shared prefix
            size_t j = 0; for (; j < r.size(); ++j, ++pos) if (r[j] != a[j]) break;
            e.pos = pos; e.type = T_INS; e.bases = a.substr(j); out.e.push_back(e);
        } else {                                                      // deletion of everything after the shared prefix
            size_t j = 0; for (; j < a.size(); ++j, ++pos) if (r[j] != a[j]) break;
            if (j == r.size()) { err = "Error: no deleted bases\n"; return false; }
            for (; j < r.size(); ++j, ++pos) { e.pos = pos; e.type = T_DEL; e.bases.clear(); out.e.push_back(e); }
        }
        prev_pos = pos;
        s = n;
    }
    return true;
}

}  // namespace

bool parse_mutation_input(int type, const char *path, const std::vector<ContigName> &contigs, MutInput &out, std::string &err)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { err = fmt("[dwgsim_core] fail to open file '%s'. Abort!\n", path); return false; }
    out.type = type; out.e.clear();
    const bool ok = type == 1 ? read_txt(fp, contigs, out, err) : type == 0 ? read_bed(fp, contigs, out, err) : type == 2 ? read_vcf(fp, contigs, out, err) : false;
    fclose(fp);
    if (!ok && err.empty()) err = "Error: mutation input type unrecognized!\n";
    return ok;
}

bool parse_regions(const char *path, const std::vector<ContigName> &contigs, Regions &out, std::string &err)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { err = fmt("[dwgsim_core] fail to open file '%s'. Abort!\n", path); return false; }
    out = Regions();
    ContigCursor cur(contigs);
    char name[1024]; uint32_t start, end; long prev_contig = -1; uint32_t prev_start = 0, prev_end = 0;
    bool ok = true;
    while (ok && 0 < fscanf(fp, "%1023s\t%u\t%u", name, &start, &end)) {
        bool moved;
        if (!cur.seek(name, &moved)) { err = fmt("Error: contig not found [%s].  Are you sure your BED is coordinate sorted?\n", name); ok = false; break; }
        const int64_t len = contigs[cur.i].len; const long ci = (long)cur.i;
        if (len < (int64_t)start) { err = fmt("Error: start out of range [%s,%u]\n", name, start); ok = false; break; }
        if (len < (int64_t)end) { err = fmt("Error: end out of range [%s,%u]\n", name, end); ok = false; break; }
        if (end < start) { err = fmt("Error: end < start [%s,%u,%u]\n", name, start, end); ok = false; break; }
        if (prev_contig == ci && start < prev_start) { err = fmt("Error: the input was not sorted [%s,%u,%u,%u]\n", name, start, end, end - start); ok = false; break; }
        if (prev_contig == ci && start <= prev_end && prev_start <= start) { if (prev_end < end) { out.end.back() = end; prev_end = end; } }   // merge
        else { prev_contig = ci; prev_start = start; prev_end = end; out.contig.push_back((uint32_t)ci); out.start.push_back(start); out.end.push_back(end); }
        int b; while (EOF != (b = fgetc(fp))) if (b == '\n' || b == '\r') break;     // the rest of the line is ignored
    }
    fclose(fp);
    return ok;
}

void resolve_mutation_input(const MutInput &in, uint32_t contig, const uint8_t *ascii, int64_t l, uint32_t seed, bool is_hap_mode, ResolvedContig &out)
{
    std::map<int32_t, uint16_t> cell;                         // touched cells, both haplotypes
    std::map<int32_t, std::vector<uint8_t>> ins[2];           // last insertion at a position wins (the cell is overwritten)
    auto get = [&](int64_t p) -> uint16_t { auto it = cell.find((int32_t)p); if (it != cell.end()) return it->second; const uint8_t c = code_of(ascii[p]); return (uint16_t)(c | (c << 8)); };
    auto set_hap = [&](int64_t p, int h, uint8_t v) { uint16_t w = get(p); w = h ? (uint16_t)((w & 0x00ff) | (v << 8)) : (uint16_t)((w & 0xff00) | v); cell[(int32_t)p] = w; };
    auto hap_cell = [&](int64_t p, int h) -> uint8_t { const uint16_t w = get(p); return (uint8_t)(h ? w >> 8 : w & 0xff); };
    auto add_insertion = [&](int64_t p, uint8_t c, int hapmask, const std::string *bases, uint32_t random_len, uint64_t entry) {   // mut.c:282-377
        std::vector<uint8_t> P;
        if (!bases) { P.resize(random_len); for (uint32_t j = 0; j < random_len; ++j) P[random_len - 1 - j] = (uint8_t)(uint64_t)(wide_uniform(seed, contig, D_MUTIN_BASE, entry, j) * 4.0); }
        else { P.resize(bases->size()); for (size_t j = 0; j < bases->size(); ++j) { int b = code_of((*bases)[j]); if (b >= 4) b = (int)(wide_uniform(seed, contig, D_MUTIN_BASE, entry, (uint32_t)j) * 4.0); P[j] = (uint8_t)b; } }
        for (int h = 0; h < 2; ++h) if (hapmask & (1 << h)) { ins[h][(int32_t)p] = P; set_hap(p, h, (uint8_t)(T_INS | c)); }
    };
    (void)l;
    for (size_t k = 0; k < in.e.size(); ++k) {
        const MutEntry &e = in.e[k];
        if (e.contig != contig) { if (in.type == 0 && contig < e.contig) break; continue; }
        if (in.type == 0) {                                   // bed: ploidy per entry (mut.c:661-669)
            int hapmask, which = 0; bool hom = false;
            if (is_hap_mode || wide_uniform(seed, contig, D_MUTIN, k, 0) < 0.333333) { hom = true; hapmask = 3; }
            else { which = wide_uniform(seed, contig, D_MUTIN, k, 1) < 0.5 ? 0 : 1; hapmask = 1 << which; }
            const bool random = e.bases == "*";
            if (e.type == T_SUB || e.type == T_DEL) {
                for (uint32_t j = e.pos; j < e.end; ++j) {
                    uint8_t c = code_of(ascii[j]);
                    if (e.type == T_SUB) {
                        if (random) c = (uint8_t)((c + (uint64_t)(wide_uniform(seed, contig, D_MUTIN_BASE, k, j - e.pos) * 3.0 + 1)) & 3);
                        else c = code_of(e.bases[j - e.pos]);
                    }
                    const uint8_t v = (uint8_t)(e.type | c);
                    if (hom) { set_hap(j, 0, v); set_hap(j, 1, v); } else set_hap(j, which, v);
                }
            } else add_insertion(e.pos, code_of(ascii[e.pos]), hapmask, random ? nullptr : &e.bases, e.end - e.pos, k);
        } else {                                              // txt / vcf (mut.c:725-744)
            const int64_t p = (int64_t)e.pos - 1;
            const uint8_t c = code_of(ascii[p]);
            if (e.type == T_DEL) { for (int h = 0; h < 2; ++h) if (e.is_hap & (1 << h)) set_hap(p, h, (uint8_t)(hap_cell(p, h) | T_DEL | c)); }
            else if (e.type == T_SUB) { for (int h = 0; h < 2; ++h) if (e.is_hap & (1 << h)) set_hap(p, h, (uint8_t)(T_SUB | code_of(e.bases[0]))); }
            else add_insertion(p, c, e.is_hap, &e.bases, 0, k);
        }
    }
    out.pos.clear(); out.cells.clear(); out.ins[0].clear(); out.ins[1].clear();
    for (auto &kv : cell) { out.pos.push_back(kv.first); out.cells.push_back(kv.second); }
    for (int h = 0; h < 2; ++h) for (auto &kv : ins[h]) {
        // an insertion whose cell was later overwritten by another type is unreachable; keep only live ones
        const uint16_t w = get(kv.first); const uint8_t v = (uint8_t)(h ? w >> 8 : w & 0xff);
        if ((v & TMASK) == T_INS) out.ins[h].push_back(InsPayload{kv.first, kv.second});
    }
}

}  // namespace dw
