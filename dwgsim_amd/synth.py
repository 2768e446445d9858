"""Deterministic synthetic reference genomes (real genomes are not available offline).

splitmix64-seeded uniform ACGT contigs with optional N runs; the same bytes on every machine, so
tests, bench.py and the CPU baseline all see identical inputs.
"""
from __future__ import annotations
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(n: int, seed: int) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (vectorised: state_k = seed + (k+1)*gamma)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


_LUT4 = None


def _lut4() -> np.ndarray:
    """byte (four 2-bit codes, lowest bits first) -> the four ASCII bases as one little-endian uint32"""
    global _LUT4
    if _LUT4 is None:
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8).astype(np.uint32)
        b = np.arange(256, dtype=np.uint32)
        _LUT4 = acgt[b & 3] | (acgt[(b >> 2) & 3] << 8) | (acgt[(b >> 4) & 3] << 16) | (acgt[(b >> 6) & 3] << 24)
    return _LUT4


def random_contig(length: int, seed: int, n_runs=()) -> np.ndarray:
    """uint8 ASCII array of `length` bases; n_runs = iterable of (start, end) half-open N blocks.
    Base i = "ACGT"[(w[i // 32] >> 2 * (i % 32)) & 3] with w = splitmix64(seed); generated in chunks through a byte -> 4-base
    table, so a 250 Mb contig takes seconds and no 32x temporary."""
    nwords = (length + 31) // 32
    out = np.empty(nwords * 32, dtype=np.uint8)
    out32 = out.view(np.uint32)
    lut = _lut4()
    CH = 1 << 20
    with np.errstate(over="ignore"):
        for a in range(0, nwords, CH):
            b = min(nwords, a + CH)
            k = np.arange(a + 1, b + 1, dtype=np.uint64)
            z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            out32[a * 8:b * 8] = lut[z.view(np.uint8)]       # little-endian: byte j of word k holds bases 32k + 4j .. +3
    out = out[:length]
    for a, b in n_runs:
        out[a:b] = ord("N")
    return out


def write_fasta(path: str, contigs, width: int = 60) -> None:
    """contigs: iterable of (name, uint8 array)."""
    with open(path, "wb") as f:
        for name, arr in contigs:
            f.write(b">" + name.encode() + b"\n")
            n = len(arr)
            full = (n // width) * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = arr[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if n > full:
                f.write(arr[full:].tobytes() + b"\n")


def repeat_rich_contig(length: int, seed: int) -> np.ndarray:
    """Random bases with homopolymers, short tandem repeats and N blocks sprinkled in: the worst case
    for indel left-justification (long shifts, interacting neighbours)."""
    out = random_contig(length, seed)
    r = _splitmix64(4 * (length // 400 + 8), seed ^ 0xABCDEF)
    k = 0
    pos = 50
    units = [b"A", b"T", b"C", b"G", b"AT", b"CA", b"CAG", b"TTAGGG", b"AAAT", b"GC"]
    while pos < length - 400:
        kind = int(r[k] % np.uint64(12)); reps = int(r[k + 1] % np.uint64(40)) + 3; gap = int(r[k + 2] % np.uint64(500)) + 20
        k += 3
        if kind < len(units):
            u = np.frombuffer(units[kind], dtype=np.uint8)
            seg = np.tile(u, reps)[: min(len(u) * reps, length - pos - 1)]
            out[pos:pos + len(seg)] = seg
            pos += len(seg)
        elif kind == 10:
            n = reps * 3
            out[pos:pos + n] = ord("N")
            pos += n
        pos += gap
    return out


def genome_like_contig(length: int, seed: int, n_runs=(), gc: float = 0.41, sine_frac: float = 0.11, line_frac: float = 0.17, sat_every: int = 9000,
                       homopolymer_every: int = 2500, soft_mask: bool = True) -> np.ndarray:
    """A contig with a genome's COMPOSITION instead of i.i.d. uniform bases (round-5 verdict, item 7): the Ion Torrent flow model does one event test per
    homopolymer (dwgsim.c:281-364) and left-justification walks through repeats (mut.c:482-589), so uniform bases flatter both.
      * background: independent bases with the given GC content (41 %: human; 50.8 %: E. coli);
      * an interspersed SINE-like family: copies of ONE 300-base consensus, each with ~10 % of its bases substituted, a poly-A tail of 12-30 bases, either
        strand, `sine_frac` of the contig (Alu: 11 % of the human genome); a LINE-like family of 1 kb at ~15 % divergence, 5'-truncated copies, `line_frac` (L1: 17 %);
      * microsatellites: a tandem repeat of a 1-6 base unit, 8-45 copies, about every `sat_every` bases; homopolymer runs of 6-24 bases about every
        `homopolymer_every` bases;
      * repeats are SOFT-MASKED (lower case) as RepeatMasker / UCSC FASTA files have them (nst_nt4_table maps both cases alike: dwgsim.c:56-73);
      * N blocks as given.
    Everything is drawn from splitmix64 streams of `seed`: the same bytes on every machine."""
    with np.errstate(over="ignore"):
        # background with the GC content: one byte of the stream per base through a 256-entry table
        n_gc = int(round(gc * 128)) * 2                      # table entries that are C or G
        tab = np.empty(256, dtype=np.uint8)
        half_at, half_gc = (256 - n_gc) // 2, n_gc // 2
        tab[:half_at] = ord("A"); tab[half_at:half_at + half_gc] = ord("C"); tab[half_at + half_gc:half_at + n_gc] = ord("G"); tab[half_at + n_gc:] = ord("T")
        out = np.empty((length + 7) // 8 * 8, dtype=np.uint8)
        CH = 1 << 21
        for a in range(0, len(out) // 8, CH):
            b = min(len(out) // 8, a + CH)
            k = np.arange(a + 1, b + 1, dtype=np.uint64)
            z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            out[a * 8:b * 8] = tab[z.view(np.uint8)]
        out = out[:length]
    lower = np.zeros(length, dtype=bool)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def family(cons_len, frac, div, fam_seed, tail, truncate):
        if frac <= 0 or length < 4 * cons_len:
            return
        cons = acgt[(_splitmix64(cons_len, fam_seed) >> np.uint64(40)).astype(np.int64) & 3]
        n = int(frac * length / (cons_len * (0.6 if truncate else 1.0)))
        r = _splitmix64(4 * n, fam_seed ^ 0x5151)
        pos = np.sort((r[0::4] % np.uint64(max(1, length - cons_len - 64))).astype(np.int64))
        strand = (r[1::4] & np.uint64(1)).astype(bool)
        cut = ((r[2::4] % np.uint64(cons_len * 4 // 5)).astype(np.int64) if truncate else np.zeros(n, dtype=np.int64))      # 5' truncation (LINE-like)
        tails = (r[3::4] % np.uint64(19)).astype(np.int64) + 12 if tail else np.zeros(n, dtype=np.int64)
        mut = _splitmix64(n * cons_len, fam_seed ^ 0xA11).reshape(n, cons_len)
        hit = (mut & np.uint64(0xFFFF)).astype(np.float64) < div * 65536.0
        shift = ((mut >> np.uint64(20)) % np.uint64(3)).astype(np.int64) + 1
        idx = np.searchsorted(acgt, cons)[None, :].repeat(n, 0)
        copies = acgt[np.where(hit, (idx + shift) & 3, idx)]
        last_end = 0
        for i in range(n):
            c = copies[i, cut[i]:]
            if tails[i]:
                c = np.concatenate([c, np.full(tails[i], ord("A"), dtype=np.uint8)])
            if strand[i]:
                c = comp[c[::-1]]
            p = max(int(pos[i]), last_end)
            if p + len(c) >= length:
                break
            out[p:p + len(c)] = c; lower[p:p + len(c)] = True
            last_end = p + len(c)

    family(300, sine_frac, 0.10, seed ^ 0x51AE, True, False)
    family(1000, line_frac, 0.15, seed ^ 0x11AE, False, True)
    units = [b"A", b"T", b"C", b"G", b"CA", b"GT", b"AT", b"GA", b"AAT", b"CAG", b"AAAT", b"GATA", b"TTAGGG", b"AAAAC"]
    if sat_every > 0:
        n = length // sat_every
        r = _splitmix64(3 * n + 3, seed ^ 0x5A7)
        for i in range(n):
            p = i * sat_every + int(r[3 * i] % np.uint64(sat_every))
            u = np.frombuffer(units[int(r[3 * i + 1] % np.uint64(len(units)))], dtype=np.uint8)
            reps = int(r[3 * i + 2] % np.uint64(38)) + 8
            seg = np.tile(u, reps)
            if p + len(seg) < length:
                out[p:p + len(seg)] = seg; lower[p:p + len(seg)] = True
    if homopolymer_every > 0:
        n = length // homopolymer_every
        r = _splitmix64(3 * n + 3, seed ^ 0x40B0)
        for i in range(n):
            p = i * homopolymer_every + int(r[3 * i] % np.uint64(homopolymer_every))
            run = int(r[3 * i + 2] % np.uint64(19)) + 6
            if p + run < length:
                out[p:p + run] = acgt[int(r[3 * i + 1] % np.uint64(4)) if int(r[3 * i + 1] >> np.uint64(8)) % 10 >= 6 else (0 if int(r[3 * i + 1]) & 4 else 3)]      # mostly poly-A / poly-T
    if soft_mask:
        out[lower] |= 0x20
    for a, b in n_runs:
        out[a:b] = ord("N")
    return out


# named workloads
def workload_contigs(name: str):
    if name == "tiny":            # a few kb with an N run, for unit tests
        return [("t1", random_contig(6000, 11, [(2500, 2530)])), ("t2", random_contig(4000, 12)), ("short", random_contig(300, 13))]
    if name == "repeats":         # left-justification stress
        return [("rep1", repeat_rich_contig(1_500_000, 21)), ("rep2", repeat_rich_contig(300_000, 22))]
    if name == "ecoli":           # S2: one contig, E. coli K-12 MG1655 length
        return [("ecoli_synth", random_contig(4_641_652, 1))]
    if name == "chr20":           # S3: chr20-length with telomere / internal N blocks
        return [("chr20_synth", random_contig(64_444_167, 2, [(0, 60_000), (26_400_000, 26_900_000), (64_334_167, 64_444_167)]))]
    if name == "ecoli_like":      # S2 with a bacterial genome's composition: 50.8 % GC, a few insertion-sequence copies and rRNA operons (1 kb family, 1 %), no soft-masking
        return [("ecoli_like", genome_like_contig(4_641_652, 101, gc=0.508, sine_frac=0.0, line_frac=0.012, sat_every=60000, homopolymer_every=20000, soft_mask=False))]
    if name == "chr20_like":      # S3 with a human chromosome's composition: 41 % GC, SINE- / LINE-like families, microsatellites, homopolymers, soft-masked repeats
        return [("chr20_like", genome_like_contig(64_444_167, 102, [(0, 60_000), (26_400_000, 26_900_000), (64_334_167, 64_444_167)]))]
    if name == "assembly5k":      # a scaffold-level assembly: 5000 contigs, exponential lengths with a mean of 30 kb (N50 ~ 50 kb), 150 Mb in all
        u = (_splitmix64(5000, 77) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
        lens = np.maximum(1000, (-30000.0 * np.log(1.0 - u)).astype(np.int64))
        out = []
        for i, l in enumerate(lens.tolist()):
            runs = [(l // 3, l // 3 + 200)] if i % 9 == 4 and l > 3000 else []          # a gap of Ns in some scaffolds
            out.append((f"scaffold{i + 1}", random_contig(int(l), 5000 + i, runs)))
        return out
    if name in ("grch38", "grch38_mini"):      # S4 (and S5 with the Ion Torrent flags): 24 contigs with the GRCh38 primary-assembly lengths
        scale = 1 if name == "grch38" else 64   # grch38_mini: every length / 64 (48 Mb), same layout -- for CPU-side checks of the multi-contig plumbing
        return [(nm, random_contig(l // scale, 1000 + i, [(a // scale, b // scale) for a, b in grch38_n_runs(nm, l)])) for i, (nm, l) in enumerate(GRCH38_PRIMARY)]
    raise ValueError(name)


# GRCh38 primary assembly (chr1-22, X, Y) sequence lengths, GCA_000001405.15
GRCH38_PRIMARY = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979),
    ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
    ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285),
    ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
]


def grch38_n_runs(name: str, l: int):
    """Telomere / centromere-like N blocks (~5 % of the genome in all): 10 kb at both ends, a centromere block of
    3 % of the length at 40 %, the short arms of the acrocentric chromosomes (first 15 %), and the unresolved tail of chrY (last 30 %)."""
    runs = [(0, 10_000), (l - 10_000, l), (int(l * 0.40), int(l * 0.40) + int(l * 0.03))]
    if name in ("chr13", "chr14", "chr15", "chr21", "chr22"):
        runs.append((0, int(l * 0.15)))
    if name == "chrY":
        runs.append((int(l * 0.70), l))
    return runs


if __name__ == "__main__":
    import sys
    write_fasta(sys.argv[2], workload_contigs(sys.argv[1]))
