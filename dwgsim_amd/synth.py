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


def random_contig(length: int, seed: int, n_runs=()) -> np.ndarray:
    """uint8 ASCII array of `length` bases; n_runs = iterable of (start, end) half-open N blocks."""
    nwords = (length + 31) // 32
    w = _splitmix64(nwords, seed)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
    codes = ((w[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)[:length]
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
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
    raise ValueError(name)


if __name__ == "__main__":
    import sys
    write_fasta(sys.argv[2], workload_contigs(sys.argv[1]))
