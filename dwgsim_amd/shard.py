"""Read-index sharding across GPUs: pairs are independent given the mutated
haplotypes and the counter-based RNG, so rank r simulates a contiguous read-index range and no
data-path collective is needed.  The only cross-shard quantity is the running random-read count
`rand_ii` that appears in random reads' names (dwgsim.c:1042,1096): one integer per rank,
exchanged with all_gather; every rank then offsets its own range."""
from __future__ import annotations


def shard_range(n_pairs: int, rank: int, world: int):
    """Contiguous, ordered, near-equal ranges: concatenating shard outputs in rank order reproduces
    the single-process output byte for byte."""
    base, rem = divmod(n_pairs, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def exchange_rand_base(ctx, cid: int, first: int, n: int, rank: int, world: int, dist, device=None, rand_before_contig: int = 0) -> int:
    """rand_base of this rank's range = random reads before the contig + those in ranges of lower ranks."""
    if world == 1 or dist is None:
        return rand_before_contig
    import torch
    mine = torch.tensor([ctx.count_random(cid, first, n)], dtype=torch.int64, device=device)
    allc = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, mine)          # one int64 per rank; a single host read-back
    return rand_before_contig + int(allc[:rank].sum().item())


def simulate_shard(ctx, cid: int, n_pairs: int, rank: int, world: int, dist, device=None, rand_before_contig: int = 0, slot: int = 0):
    first, n = shard_range(n_pairs, rank, world)
    rand_base = exchange_rand_base(ctx, cid, first, n, rank, world, dist, device, rand_before_contig)
    return first, n, ctx.simulate(cid, first, n, rand_base, slot)
