"""world_size-2 `gloo` test of the multi-GPU path (CPU): each rank owns a disjoint read-index range of
the same contig, the ranks exchange one integer (their random-read counts) with all_gather, and the
rank-ordered concatenation of the shard outputs must equal the oracle's single-process output.
Each rank drives the CPU emulation build of the kernels (tests/emu) in place of its GPU."""
import os, subprocess, sys, tempfile
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from dwgsim_amd import api, shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = api.load(os.path.join(sys.argv[1], "tests", "emu", "libdwgsim_emu.so"))
flags, fasta, outdir = sys.argv[2], sys.argv[3], sys.argv[4]
params = api.parse_flags(flags, lib)
contigs = api.read_fasta(fasta)
tot = sum(len(a) for _, a in contigs)
out = []          # per contig: {stream: bytes} of this rank's shard
n_sim, rand_ii, n_ref = 0, 0, len(contigs)
with api.Context(params, 0, lib) as ctx:
    for ci, (name, arr) in enumerate(contigs):
        n_ref -= 1
        n_pairs = api.pairs_for_contig(params, len(arr), tot, n_ref == 0, n_sim, lib)
        if n_pairs < 0:
            continue
        cid = ctx.add_contig(name, arr, ci)
        ctx.mutate(cid)                       # every rank re-runs the (deterministic) walk: no broadcast
        first, n, b = shard.simulate_shard(ctx, cid, n_pairs, rank, world, dist, rand_before_contig=rand_ii)
        out.append({s: (ctx.fetch(0, s, b.bytes[s]) if b.bytes[s] else b"") for s in range(3)})
        # total random reads of the contig (all ranks need it for the next contig's offset)
        t = torch.tensor([b.n_random], dtype=torch.int64)
        dist.all_reduce(t)
        rand_ii += int(t.item()); n_sim += n_pairs
        ctx.drop_contig(cid)
pickle.dump(out, open(os.path.join(outdir, f"rank{rank}.pkl"), "wb"))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_sharding_reproduces_single_process_output(oracle_bin, golden_dir):
    from parity_common import run_oracle, STREAMS, first_diff
    import pickle
    subprocess.run([os.path.join(HERE, "emu", "build.sh")], check=True, stdout=subprocess.DEVNULL)
    flags = "-z 9 -N 1500 -y 0.25 -1 50 -2 50 -d 200 -s 20"
    fasta = os.path.join(golden_dir, "tiny.fa")
    with tempfile.TemporaryDirectory() as t:
        want = run_oracle(oracle_bin, fasta, flags, t)
        w = os.path.join(t, "worker.py")
        open(w, "w").write(WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, w, ROOT, flags, fasta, t], env=dict(env, RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        shards = [pickle.load(open(os.path.join(t, f"rank{r}.pkl"), "rb")) for r in range(2)]
    # file order is contig-major, rank-minor: [contig A rank 0][A rank 1][contig B rank 0][B rank 1]...
    for s in STREAMS:
        got = b"".join(shards[r][c][s] for c in range(len(shards[0])) for r in range(2))
        assert got == want[s], f"{STREAMS[s]}: " + first_diff(got, want[s])


def test_shard_ranges_partition_the_index_space():
    from dwgsim_amd.shard import shard_range
    for n in (0, 1, 7, 488595, 10 ** 9 + 7):
        for w in (1, 2, 3, 8):
            pos = 0
            for r in range(w):
                f, c = shard_range(n, r, w)
                assert f == pos and c >= 0
                pos += c
            assert pos == n
