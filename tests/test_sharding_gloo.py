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


# The flow of bench.py --mode strong / dw_job.cpp: all contigs in ONE group, its pairs cut into batches, batch b on rank b mod world,
# per-batch random-read counts exchanged by ONE all-gather per group.  argv: root flags fasta outdir library batch_pairs
WORKER_BATCHES = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from dwgsim_amd import api
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = api.load(sys.argv[5])
flags, fasta, outdir, batch = sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[6])
params = api.parse_flags(flags, lib)
contigs = api.read_fasta(fasta)
out = []
with api.Context(params, 0, lib) as ctx:
    sched = list(api.schedule_contigs(params, contigs, ctx, lib))
    h0 = ctx.add_contigs([(name, arr) for _, name, arr, _, _ in sched], indices=[ci for ci, _, _, _, _ in sched])
    ctx.mutate_async(h0); ctx.mutate_wait(h0)             # every rank walks every contig itself: no broadcast
    batches = list(api.split_ranges([(h0 + k, 0, e[3]) for k, e in enumerate(sched) if e[3] > 0], batch))
    mine = [b for b in range(len(batches)) if b % world == rank]
    flat = [r for b in mine for r in batches[b]]
    per = iter(ctx.count_random_ranges(flat, per_range=True) if flat else [])
    width = -(-len(batches) // world)
    vec = torch.zeros(width, dtype=torch.int64)
    for k, b in enumerate(mine):
        vec[k] = sum(next(per) for _ in batches[b])
    allv = torch.empty(world * width, dtype=torch.int64)
    dist.all_gather_into_tensor(allv, vec)               # one integer per batch; the only thing that crosses ranks
    allv = allv.view(world, width)
    run, bases = 0, {}
    for b in range(len(batches)):
        bases[b] = run
        run += int(allv[b % world, b // world])
    for k, b in enumerate(mine):
        bt = ctx.simulate_ranges(batches[b], bases[b], k & 1)
        out.append((0, b, {s: (ctx.fetch(k & 1, s, bt.bytes[s]) if bt.bytes[s] else b"") for s in range(3)}))
        assert int(bt.n_random) == int(allv[b % world, b // world])
pickle.dump(out, open(os.path.join(outdir, f"rank{rank}.pkl"), "wb"))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_with_round_robin_batches_reproduce_single_process_output(oracle_bin, golden_dir):
    """The sharding of bench.py --gpus N and of the job level (dw_job.cpp) on two gloo ranks, each driving the CPU emulation build."""
    from parity_common import run_oracle, STREAMS, first_diff
    import pickle
    subprocess.run([os.path.join(HERE, "emu", "build.sh")], check=True, stdout=subprocess.DEVNULL)
    flags = "-z 9 -N 1500 -y 0.25 -1 50 -2 50 -d 200 -s 20"
    fasta = os.path.join(golden_dir, "tiny.fa")
    with tempfile.TemporaryDirectory() as t:
        want = run_oracle(oracle_bin, fasta, flags, t)
        w = os.path.join(t, "worker.py")
        open(w, "w").write(WORKER_BATCHES)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, w, ROOT, flags, fasta, t, os.path.join(ROOT, "tests", "emu", "libdwgsim_emu.so"), "170"], env=dict(env, RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        parts = [pickle.load(open(os.path.join(t, f"rank{r}.pkl"), "rb")) for r in range(2)]
    merged = sorted(parts[0] + parts[1], key=lambda m: (m[0], m[1]))
    for s in STREAMS:
        got = b"".join(m[2][s] for m in merged)
        assert got == want[s], f"{STREAMS[s]}: " + first_diff(got, want[s])


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
