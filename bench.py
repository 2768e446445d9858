#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X dwgsim hot path on BASELINE.json's metric
("M read-pairs/sec, 2x150 bp PE").

A step = one pass of the hot path over one job held resident in HBM: the mutation walk of the
contig (replaces mut_diref) followed by the per-pair loop over the job's whole read-index range
(replaces dwgsim.c:636-1099), FASTQ text left packed in HBM.

Workload at N=1 (default): BASELINE configs[2], the largest single-GPU configuration -- S3, a chr20-sized
synthetic contig (64 444 167 bp with telomere / centromere N blocks), `-z 13 -1 150 -2 150 -C 30 -o 1`
(-r 0.001 -R 0.1 are dwgsim's defaults) => 6 783 597 pairs and 4.9 GB of FASTQ text per step.
`--workload ecoli` is configs[1] (S2, 488 595 pairs), `--workload grch38` the whole-genome S4 job.

N>1 (one process per GPU, launched by torch.distributed.run): no data-path collective and no RCCL -- the ranks
own disjoint read-index ranges and exchange ONE integer each per step (the random-read count that offsets
rand_ii, dwgsim.c:1042,1096) through a host-side (gloo) all-gather.
  --mode weak   (default) every rank simulates a full job's worth of pairs: rank r owns [r n, (r+1) n) of an N-times deeper job
  --mode strong one fixed job (e.g. the S4 genome at 30x) split over the ranks per contig

`value` = pairs of all ranks / max-over-ranks wall time of the timed steps, text left in HBM (the contract).  Two more
legs are measured at N=1 and reported beside it (never as `value`): `host_landed` (text copied into page-locked host
memory through the asynchronous two-slot pipeline, copies overlapped with kernels) and `end_to_end` (the dwgsim-hip
executable: FASTA in, five output files out, gzip included).

Prints ONE JSON line on rank 0.
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLAGS = "-z 13 -1 150 -2 150 -C 30 -o 1"
ION_FLAGS = "-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.01 -o 1"
ALGO_BYTES_PER_PAIR_2x150 = 863.0  # 713 B FASTQ written + 150 B haplotype bases read at 4 bit/base
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
WORKLOADS = {"ecoli": ("S2", 1), "chr20": ("S3", 2), "grch38": ("S4", 3), "grch38_mini": ("S4/64", 3)}


def _tmpdir():
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None


def effective_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(contigs, flags, sample_pairs=250000):
    """The unmodified reference (oracle/_ref/dwgsim, kind 'reference') -- or the oracle port in drand48 mode if the prebuilt
    binary is absent -- timed on this box's host cores on a bounded sample of the same workload (same contig, same flags,
    -N sample instead of -C): (1) one process as the reference runs (gzip included), (2) the same with the gzip calls made
    no-ops (LD_PRELOAD shim oracle/build/libnullgz.so: what the simulation itself costs), (3) one process per host core,
    each with its own seed, all at once (the reference is single-threaded; this is how a user would fill the box)."""
    from dwgsim_amd import synth
    ncores = effective_cores()
    with tempfile.TemporaryDirectory(dir=_tmpdir()) as t:
        fa = os.path.join(t, "ref.fa")
        synth.write_fasta(fa, contigs)
        ref = os.path.join(ROOT, "oracle", "_ref", "dwgsim")
        base = [f for f in flags.split()]
        i = base.index("-C"); del base[i:i + 2]
        base += ["-N", str(sample_pairs)]
        if os.path.exists(ref):
            kind, exe = "reference", [ref]
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
            kind, exe = "port", [os.path.join(ROOT, "oracle", "build", "dwgsim_oracle"), "--rng", "drand48"]

        def run(seed, tag, env=None):
            b = list(base); b[b.index("-z") + 1] = str(seed)
            return subprocess.Popen(exe + b + [fa, os.path.join(t, tag)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)

        t0 = time.time(); assert run(13, "one").wait() == 0; dt1 = time.time() - t0
        nullgz = os.path.join(ROOT, "oracle", "build", "libnullgz.so")
        dt_null = None
        if os.path.exists(nullgz):
            t0 = time.time(); assert run(13, "null", dict(os.environ, LD_PRELOAD=nullgz)).wait() == 0; dt_null = time.time() - t0
        # all cores at once: one process per core, each with its own seed, on the E. coli-sized contig S2 (the reference walks its whole
        # input before the first read: with the chr20-sized contig that fixed cost, times 256 processes, would dominate a bounded sample)
        per = 20000
        fa_small = os.path.join(t, "s2.fa")
        synth.write_fasta(fa_small, synth.workload_contigs("ecoli"))
        b_all = list(base); b_all[b_all.index("-N") + 1] = str(per)
        t0 = time.time()
        procs = []
        for k in range(ncores):
            bb = list(b_all); bb[bb.index("-z") + 1] = str(100 + k)
            procs.append(subprocess.Popen(exe + bb + [fa_small, os.path.join(t, f"p{k}")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        assert all(p.wait() == 0 for p in procs)
        dtn = time.time() - t0
    out = {"value": round(sample_pairs / dt1 / 1e6, 6), "unit": "M read-pairs/s", "cores": 1, "kind": kind,
           "sample": f"{sample_pairs} pairs of the same workload (same contig and flags, -N {sample_pairs}; gzip FASTQ as the reference writes it), {dt1:.1f} s wall, one thread (the reference is single-threaded)",
           "all_cores": {"value": round(per * ncores / dtn / 1e6, 6), "cores": ncores, "hardware_threads": os.cpu_count(), "sample": f"{ncores} processes x {per} pairs (same flags, S2 contig), seeds 100.., {dtn:.1f} s wall"}}
    if dt_null is not None:
        out["null_sink"] = {"value": round(sample_pairs / dt_null / 1e6, 6), "cores": 1, "sample": f"the same run with gzopen/gzwrite/gzputc/gzclose made no-ops by LD_PRELOAD, {dt_null:.1f} s wall"}
    return out


def host_landed_leg(api, ctx, cid, n_pairs, params, steps=2, batch=1 << 20):
    """Text of every batch copied into page-locked host memory while the next batch is being computed (two slots, copy stream)."""
    import ctypes as C
    lib = ctx.lib
    nstreams = [s for s in range(3)]
    bufs = {}

    def ensure(slot, s, n):
        cap, p = bufs.get((slot, s), (0, None))
        if n > cap:
            if p:
                lib.dwgsim_hip_host_free(p)
            cap = int(n * 1.1) + 4096
            p = lib.dwgsim_hip_host_alloc(cap)
            assert p, "page-locked allocation failed"
            bufs[(slot, s)] = (cap, p)
        return bufs[(slot, s)]

    def finish(slot):
        b = ctx.wait(slot)
        for s in nstreams:
            if b.bytes[s]:
                cap, p = ensure(slot, s, b.bytes[s])
                ctx.fetch_async(slot, s, p, cap)
        return b

    def one_pass():
        tot = 0
        k = 0
        for off in range(0, n_pairs, batch):
            n = min(batch, n_pairs - off)
            slot = k & 1
            ctx.simulate_async(cid, off, n, 0 if off == 0 else api.RAND_CHAIN, slot)
            if k > 0:
                b = finish((k - 1) & 1); tot += sum(b.bytes)
            k += 1
        b = finish((k - 1) & 1); tot += sum(b.bytes)
        ctx.fetch_wait(0); ctx.fetch_wait(1)
        return tot
    one_pass()                                                # warm-up: allocations, page-locking
    t0 = time.perf_counter()
    tot = 0
    for _ in range(steps):
        tot += one_pass()
    dt = time.perf_counter() - t0
    for cap, p in bufs.values():
        lib.dwgsim_hip_host_free(p)
    return {"value": round(n_pairs * steps / dt / 1e6, 3), "unit": "M read-pairs/s", "gb_per_s": round(tot / dt / 1e9, 2), "steps": steps, "batch_pairs": batch,
            "note": "FASTQ text landed in page-locked host memory: simulate_async / wait / fetch_async on two slots, copies on a second stream overlapped with the kernels of the next batch"}


def end_to_end_leg(contigs, flags, n_pairs, gzip_mode="gpu"):
    """The dwgsim-hip executable on the same job: FASTA parse, upload, walk, mutation files, reads, gzip (members made on the GPU, or
    zlib on the host cores with DWGSIM_HIP_GZIP=cpu), page-locked copies, the five output files written (to tmpfs when there is one)."""
    from dwgsim_amd import synth
    exe = os.path.join(ROOT, "dwgsim_amd", "dwgsim-hip")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory(dir=_tmpdir()) as t:
        fa = os.path.join(t, "ref.fa")
        synth.write_fasta(fa, contigs)
        t0 = time.time()
        r = subprocess.run([exe] + flags.split() + [fa, os.path.join(t, "out")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                           env=dict(os.environ, DWGSIM_HIP_GZIP=gzip_mode, DWGSIM_HIP_TIMING="1"))
        dt = time.time() - t0
        if r.returncode != 0:
            return {"error": r.stderr.decode(errors="replace")[-300:]}
        gz = sum(os.path.getsize(os.path.join(t, f)) for f in os.listdir(t) if f.endswith(".gz"))
        stages = [ln for ln in r.stderr.decode(errors="replace").splitlines() if ln.startswith("[dwgsim-hip]")]
    gzinfo = ({"where": "gpu", "members": "32 KiB of text each, dynamic Huffman codes (k_gzip)"} if gzip_mode == "gpu" else
              {"where": "cpu", "threads": effective_cores(), "zlib_level": int(os.environ.get("DWGSIM_HIP_GZIP_LEVEL", "1")), "members": "independent 1 MiB gzip members"})
    return {"seconds": round(dt, 2), "value": round(n_pairs / dt / 1e6, 3), "unit": "M read-pairs/s", "gz_bytes": gz, "gzip": gzinfo,
            "stages": stages[-1][13:] if stages else None,
            "note": "wall time of `dwgsim-hip <flags> ref.fa out` (process start to exit), outputs on " + ("tmpfs" if _tmpdir() else "the temp dir")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="chr20", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--ion", action="store_true", help="BASELINE configs[4] flags (Ion Torrent flow model, 400 bp SE, 50x) instead of 2x150 Illumina")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the host_landed / end_to_end legs")
    ap.add_argument("--flags", default=None, help="analysis only: override the dwgsim flags of the workload (the default is the BASELINE configuration)")
    ap.add_argument("--phases", action="store_true", help="analysis only: print the phase split of the -DDW_PHASE_TIMING build (DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_phases.so)")
    ap.add_argument("--share-gpu", action="store_true", help="analysis only: several ranks on one GPU (1-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) ourselves
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)

    import torch
    from dwgsim_amd import api, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback of the hot path)")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if world > torch.cuda.device_count() and not args.share_gpu:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible (--share-gpu is the analysis-only way to put several ranks on one)")
    dev = local_rank % torch.cuda.device_count() if args.share_gpu else local_rank
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")          # host-side exchange of one integer per rank and step; no RCCL on this path

    lib = api.load()
    flags = args.flags or (ION_FLAGS if args.ion else FLAGS)
    params = api.parse_flags(flags, lib)
    contigs = synth.workload_contigs(args.workload)
    tot_len = sum(len(a) for _, a in contigs)
    paired = params.length[1] > 0

    ctx = api.Context(params, dev, lib)
    if args.phases:
        ctx.debug_option("phases", 1)
    # the job: pairs per contig exactly as dwgsim_core schedules them (dwgsim.c:582-590); every contig stays resident
    job = []
    n_sim = 0
    for ci, (name, arr) in enumerate(contigs):
        n = api.pairs_for_contig(params, len(arr), tot_len, ci == len(contigs) - 1, n_sim, lib)
        if n < 0:
            continue
        cid = ctx.add_contig(name, arr, ci)
        job.append((cid, n))
        n_sim += n
    job_pairs = sum(n for _, n in job)
    if args.mode == "weak":
        my_pairs = job_pairs
        ranges = [(cid, rank * n, n) for cid, n in job]                   # rank r: [r n, (r+1) n) of the N-times deeper job
    else:
        ranges = []
        for cid, n in job:
            first, cnt = api.shard_range(n, rank, world, lib)
            ranges.append((cid, first, cnt))
        my_pairs = sum(c for _, _, c in ranges)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stats = {"walk_ms": 0.0, "exch_ms": 0.0, "sim_kernel_ms": 0.0, "bytes": 0, "n_random": 0, "launches": 0}

    def step(record):
        rand_before = 0
        nbytes = 0; nrand = 0
        for cid, first, cnt in ranges:
            t0 = time.perf_counter()
            ctx.mutate(cid)                                        # mutation walk on the GPU (every rank re-walks: deterministic, cheap)
            t1 = time.perf_counter()
            rand_base = rand_before
            if world > 1:
                # one integer per rank (no data-path collective): random reads in the ranges of lower ranks, host-side all-gather
                mine = torch.tensor([ctx.count_random(cid, first, cnt) if cnt else 0], dtype=torch.int64)
                allc = torch.empty(world, dtype=torch.int64)
                dist.all_gather_into_tensor(allc, mine)
                rand_base += int(allc[:rank].sum())
                rand_before += int(allc.sum())
            t2 = time.perf_counter()
            if cnt:
                b = ctx.simulate(cid, first, cnt, rand_base, 0)
                nbytes += int(b.bytes[0] + b.bytes[1] + b.bytes[2]); nrand += int(b.n_random)
                if world == 1:
                    rand_before += int(b.n_random)
                if record:
                    stats["sim_kernel_ms"] += b.sim_kernel_ms; stats["launches"] += 1
            if record:
                stats["walk_ms"] += (t1 - t0) * 1e3; stats["exch_ms"] += (t2 - t1) * 1e3
        if record:
            stats["bytes"] = nbytes; stats["n_random"] = nrand

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    total_pairs = my_pairs
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tp = torch.tensor([my_pairs], dtype=torch.int64)
        dist.all_reduce(tp)
        total_pairs = int(tp.item())

    if rank == 0:
        K = max(args.steps, 1)
        ms_per_step = elapsed / K * 1e3
        value = total_pairs * K / elapsed / 1e6
        sim_ms_launch = stats["sim_kernel_ms"] / max(stats["launches"], 1)          # average duration of one k_simulate launch
        pairs_per_launch = my_pairs / max(len([1 for _, _, c in ranges if c]), 1)
        text_per_pair = stats["bytes"] / max(my_pairs, 1)
        if not args.ion and args.flags is None:
            algo_per_pair = ALGO_BYTES_PER_PAIR_2x150
            algo_note = "863 algorithmic B/pair (SURVEY 8d: 713 B of FASTQ written + 150 B of haplotype bases read at 4 bit/base)"
        else:
            algo_per_pair = text_per_pair + (params.length[0] + params.length[1]) / 2.0
            algo_note = f"{algo_per_pair:.0f} algorithmic B per pair/read = {text_per_pair:.1f} B of FASTQ written (measured) + {(params.length[0] + params.length[1]) / 2:.0f} B of haplotype bases read at 4 bit/base"
        achieved = algo_per_pair * pairs_per_launch / (sim_ms_launch * 1e-3) / 1e9 if sim_ms_launch > 0 else 0.0
        prof = {}
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r02_counters.json"))).get(f"{args.workload}{'_ion' if args.ion else ''}", {})
        except Exception:
            pass
        sname, cfg_i = WORKLOADS[args.workload]
        out = {
            "metric": "M read-pairs/sec (2x150 bp PE)" if not args.ion else "M reads/sec (Ion Torrent 400 bp SE)", "value": round(value, 3), "unit": "M read-pairs/s" if paired else "M reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.mode, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{sname} ({args.workload}): {len(job)} uniform-random contig(s), {tot_len} bp in all (BASELINE configs[{4 if args.ion else cfg_i}] stand-in), dwgsim {flags}, "
                                   f"{job_pairs} pairs per job; step = mutation walk of every contig + all pairs of this rank's read-index ranges, FASTQ text left in HBM",
                       "pairs_per_gpu_per_step": my_pairs, "fastq_bytes_per_step_per_gpu": stats["bytes"], "fastq_gb_per_s": round(stats["bytes"] * world * K / elapsed / 1e9, 2),
                       "random_pairs": stats["n_random"], "parallelism": f"read-index shards x{world} ({args.mode}), host-side exchange of one integer per rank"},
            "breakdown_ms": {"walk": round(stats["walk_ms"] / K, 4), "rand_count_exchange": round(stats["exch_ms"] / K, 4), "simulate_kernels": round(stats["sim_kernel_ms"] / K, 4)},
            "roofline": {"bound": "valu", "kernel": f"k_simulate<{2 if paired else 1},{[3, 1, 2][params.reads_output_type]},{params.data_type}>",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": prof.get("traffic_bytes_per_launch") if prof.get("pairs_per_launch") == int(pairs_per_launch) else None,
                         "algorithmic_bytes_per_launch": int(algo_per_pair * pairs_per_launch), "launch_ms": round(sim_ms_launch, 4),
                         "valu": {k: prof[k] for k in ("valu_instr_per_pair", "valu_instr_per_wave", "salu_instr_per_wave", "valu_issue_active_pct", "source") if k in prof},
                         "note": algo_note + " x pairs per launch / HIP-event time of the launch (events on the library's own stream); the kernel is bound by VALU issue "
                                 "(Philox rounds, fp32 / fp64 quality normals, text formatting), not by HBM: the fraction of the HBM roof says how far the ALU work has been squeezed"},
        }
        if world == 1 and not args.no_legs:
            cid0, n0 = max(job, key=lambda x: x[1])
            ctx.mutate(cid0)
            out["host_landed"] = host_landed_leg(api, ctx, cid0, n0, params)
        ctx.close(); ctx = None
        if world == 1 and not args.no_legs and args.workload in ("ecoli", "chr20"):
            out["end_to_end"] = end_to_end_leg(contigs, flags, job_pairs)
            cpu_gz = end_to_end_leg(contigs, flags, job_pairs, "cpu")
            if cpu_gz and "seconds" in cpu_gz:
                out["end_to_end"]["with_zlib_on_host"] = {k: cpu_gz[k] for k in ("seconds", "value", "gz_bytes", "gzip")}
        if world == 1 and not args.no_cpu_baseline:
            base_contigs = contigs[:1] if len(contigs) == 1 else [c for c in contigs if c[0] == "chr20"] or contigs[:1]      # (whole-genome jobs: the chr20-sized contig)
            out["cpu_baseline"] = cpu_baseline(base_contigs, flags)
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
