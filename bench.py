#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X dwgsim hot path on BASELINE.json's metric
("M read-pairs/sec, 2x150 bp PE").

A step = one pass of the hot path over one job held resident in HBM: the mutation walk of the
contig (replaces mut_diref) followed by the per-pair loop over the job's whole read-index range
(replaces dwgsim.c:636-1099), FASTQ text left packed in HBM.  Workload at N=1: BASELINE configs[1]
(E. coli-sized synthetic contig S2, 4 641 652 bp, `-z 13 -1 150 -2 150 -C 30 -o 1` => 488 595
pairs).  N>1: one process per GPU, each rank simulates its own disjoint read-index range of an
N-times larger job over the same contig (weak scaling, no data-path collective; the only exchange
is one integer per rank -- the random-read count that offsets rand_ii).

Prints ONE JSON line on rank 0.
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLAGS = "-z 13 -1 150 -2 150 -C 30 -o 1"
ALGO_BYTES_PER_PAIR = 863.0        # 713 B FASTQ written + 150 B haplotype bases read at 4 bit/base
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(contigs, sample_pairs=250000):      # ~12 s of single-thread CPU work
    """The unmodified reference (oracle/_ref/dwgsim, kind 'reference') -- or the oracle port in drand48
    mode if the prebuilt binary is absent -- timed on this box's host cores on a bounded sample of the
    same workload (same contig, same flags, -N sample instead of -C 30)."""
    from dwgsim_amd import synth
    with tempfile.TemporaryDirectory() as t:
        fa = os.path.join(t, "ref.fa")
        synth.write_fasta(fa, contigs)
        ref = os.path.join(ROOT, "oracle", "_ref", "dwgsim")
        flags = f"-z 13 -1 150 -2 150 -N {sample_pairs} -o 1".split()
        if os.path.exists(ref):
            kind, cmd = "reference", [ref] + flags + [fa, os.path.join(t, "out")]
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
            kind, cmd = "port", [os.path.join(ROOT, "oracle", "build", "dwgsim_oracle"), "--rng", "drand48"] + flags + [fa, os.path.join(t, "out")]
        t0 = time.time()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
    return {"value": round(sample_pairs / dt / 1e6, 6), "unit": "M read-pairs/s", "cores": 1, "kind": kind,
            "sample": f"{sample_pairs} pairs of the same workload (S2 contig, 2x150, -o 1, gzip FASTQ as the reference writes it), {dt:.1f} s wall, single thread (the reference is single-threaded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ecoli")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="analysis only: torch.distributed backend (gloo lets several ranks share one GPU on a 1-GPU box)")
    ap.add_argument("--flags", default=None, help="analysis only: override the dwgsim flags of the workload (the default is the BASELINE configuration)")
    args = ap.parse_args()

    import torch
    from dwgsim_amd import api, synth, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback of the hot path)")
    dev = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist = None
    coll_device = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)

    lib = api.load()
    flags = args.flags or FLAGS
    params = api.parse_flags(flags, lib)
    contigs = synth.workload_contigs(args.workload)
    name, arr = contigs[0]
    tot_len = len(arr)
    n_pairs = api.pairs_for_contig(params, tot_len, tot_len, False, 0, lib)   # pairs of the 30x job = per-GPU share

    ctx = api.Context(params, dev, lib)
    cid = ctx.add_contig(name, arr, 0)
    first_ii = rank * n_pairs

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stats = {"walk_ms": 0.0, "count_ms": 0.0, "kernel_ms": 0.0, "sim_kernel_ms": 0.0, "bytes": 0, "n_random": 0}

    def step(record):
        t0 = time.perf_counter()
        ctx.mutate(cid)                                        # mutation walk on the GPU
        t1 = time.perf_counter()
        # one integer per rank (no data-path collective): random reads in the ranges of lower ranks
        rand_base = shard.exchange_rand_base(ctx, cid, first_ii, n_pairs, rank, world, dist, device=coll_device)
        t2 = time.perf_counter()
        b = ctx.simulate(cid, first_ii, n_pairs, rand_base, 0)
        if record:
            stats["walk_ms"] += (t1 - t0) * 1e3; stats["count_ms"] += (t2 - t1) * 1e3
            stats["kernel_ms"] += b.kernel_ms; stats["sim_kernel_ms"] += b.sim_kernel_ms
            stats["bytes"] = int(b.bytes[0] + b.bytes[1] + b.bytes[2]); stats["n_random"] = int(b.n_random)

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        K = max(args.steps, 1)
        ms_per_step = elapsed / K * 1e3
        total_pairs = n_pairs * world
        value = total_pairs * K / elapsed / 1e6
        sim_ms = stats["sim_kernel_ms"] / K
        achieved = ALGO_BYTES_PER_PAIR * n_pairs / (sim_ms * 1e-3) / 1e9 if sim_ms > 0 else 0.0
        traffic = None          # HBM bytes per launch from the PMC passes kept under profiles/ (same workload only)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj["pairs_per_launch"] == n_pairs and flags == FLAGS:
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "M read-pairs/sec (2x150 bp PE)", "value": round(value, 3), "unit": "M read-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{'S3' if args.workload == 'chr20' else 'S2'} {name}: one {tot_len} bp uniform-random contig (BASELINE configs[{2 if args.workload == 'chr20' else 1}] stand-in), dwgsim {flags}, "
                                   f"{n_pairs} pairs per GPU per step; step = mutation walk + all pairs, FASTQ text left in HBM",
                       "pairs_per_gpu": n_pairs, "fastq_bytes_per_step_per_gpu": stats["bytes"], "fastq_gb_per_s": round(stats["bytes"] * world * K / elapsed / 1e9, 2), "random_pairs": stats["n_random"],
                       "parallelism": f"read-index shards x{world}"},
            "breakdown_ms": {"walk": round(stats["walk_ms"] / K, 4), "rand_count_exchange": round(stats["count_ms"] / K, 4),
                             "batch_kernels": round(stats["kernel_ms"] / K, 4), "simulate_kernel": round(sim_ms, 4)},
            "roofline": {"bound": "hbm", "kernel": f"k_simulate<{2 if params.length[1] > 0 else 1},{[3, 1, 2][params.reads_output_type]},{params.data_type}>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(ALGO_BYTES_PER_PAIR * n_pairs),
                         "note": "863 algorithmic B/pair x pairs per launch / HIP-event time of the launch; the kernel is Philox+fp64 ALU bound, not HBM bound (DESIGN.md)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(contigs)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
