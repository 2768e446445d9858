#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X dwgsim hot path on BASELINE.json's metric
("M read-pairs/sec, 2x150 bp PE").

A step = one pass of the hot path over one job held resident in HBM: the mutation walk of every contig
(replaces mut_diref) followed by the per-pair loop over this rank's read-index ranges (replaces
dwgsim.c:636-1099), FASTQ text left packed in HBM.  Contigs are resident in groups of up to --group-bp
bases (dwgsim_hip_add_contigs): one chain of walk kernels and a few launches per group, however many contigs.

Workload at N=1 (default): BASELINE configs[2], the largest single-GPU configuration -- S3, a chr20-sized
synthetic contig (64 444 167 bp with telomere / centromere N blocks), `-z 13 -1 150 -2 150 -C 30 -o 1`
(-r 0.001 -R 0.1 are dwgsim's defaults) => 6 783 597 pairs and 4.9 GB of FASTQ text per step.
`--workload ecoli` is configs[1] (S2, 488 595 pairs), `--workload chr20_like` / `ecoli_like` the same sizes with a genome's composition instead
of uniform bases (dwgsim_amd/synth.py genome_like_contig), `--workload grch38` the whole-genome S4 job,
`--workload assembly5k` a scaffold-level assembly (5000 contigs, N50 ~ 50 kb).

N>1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run) when no launcher did.
Sharding is the product's (dw_job.cpp): the pairs of every group of contigs, in file order, are cut into batches of read-index ranges and
batch b belongs to rank b mod N; every rank walks every contig itself.  No data-path collective and no RCCL: what crosses ranks is ONE
host-side (gloo) all-gather of integers per step -- the random reads of every batch, counted (k_place) before anything is simulated, whose
running sum offsets rand_ii (dwgsim.c:1042,1096).
Steps are pipelined as the groups of a long job are (dw_job.cpp): every contig is resident twice, and the preparation of step k+1 -- walk,
random-read count, exchange: walk stream and host -- runs beside the kernels of step k.  `--no-pipeline`: every step prepares itself first.
  --mode weak   (default) per-GPU work is fixed: the job's coverage is N times the workload's (-C 30 N), i.e. N times the pairs
  --mode strong one fixed job (e.g. --workload grch38: the 325 M-pair S4 genome) split over the ranks

`value` = pairs of all ranks / max-over-ranks wall time of the timed steps, text left in HBM (the contract).  More legs are measured at
N=1 and reported beside it (never as `value`): `host_landed` (text copied into page-locked host memory through the asynchronous
two-slot pipeline), `host_landed_gz` (the same with the gzip members made on the GPU -- what the product moves), `end_to_end` (the
dwgsim-hip executable: FASTA in, five output files out) and `end_to_end_genome` (dwgsim-hip on the whole S4 genome, counting sink).

Prints ONE JSON line on rank 0.
"""
import argparse, hashlib, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLAGS = "-z 13 -1 150 -2 150 -C 30 -o 1"
ION_FLAGS = "-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.01 -o 1"
ALGO_BYTES_PER_PAIR_2x150 = 863.0  # 713 B FASTQ written + 150 B haplotype bases read at 4 bit/base
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
WORKLOADS = {"ecoli": ("S2", 1), "chr20": ("S3", 2), "ecoli_like": ("S2 with a bacterial genome's composition", 1), "chr20_like": ("S3 with a human chromosome's composition", 2), "grch38": ("S4", 3), "grch38_mini": ("S4/64", 3), "assembly5k": ("5000 scaffolds, N50 ~ 50 kb", 3)}
COUNTERS_JSON = os.path.join(ROOT, "profiles", "r06_counters.json")
STRONG_GROUP_BP = (1 << 31) - (1 << 24)      # whole-genome groups for the strong-scaling job: a group's coordinate space holds < 2^31 cells (dw_host.cpp dwgsim_hip_add_contigs): GRCh38 = 2 groups, 2 walk chains
MAX_LAUNCH_PAIRS = int(os.environ.get("DWGSIM_BENCH_MAX_LAUNCH_PAIRS", 1 << 23))         # pairs per launch at most (a launch's text buffers are sized for it: 6 GB at 2 x 150 bp)


def file_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


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


def host_landed_leg(api, ctx, cid, n_pairs, gz=False, steps=2, batch=1 << 20):
    """Output of every batch copied into page-locked host memory while the next batch is being computed (two slots, copy stream):
    the text, or (gz) the gzip members k_gzip made of it -- what dwgsim-hip moves and writes."""
    lib = ctx.lib
    bufs = {}
    ctx.set_gzip(gz)

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
        sizes = b.gz_bytes if gz else b.bytes
        for s in range(3):
            if sizes[s]:
                cap, p = ensure(slot, s, sizes[s])
                if gz:
                    ctx._chk(lib.dwgsim_hip_fetch_gz_async(ctx.h, slot, s, p, cap))
                else:
                    ctx.fetch_async(slot, s, p, cap)
        return sum(sizes), sum(b.bytes)

    def one_pass():
        moved = text = 0
        k = 0
        for off in range(0, n_pairs, batch):
            n = min(batch, n_pairs - off)
            slot = k & 1
            ctx.fetch_wait(slot)
            ctx.simulate_async(cid, off, n, 0 if off == 0 else api.RAND_CHAIN, slot)
            if k > 0:
                a, b = finish((k - 1) & 1); moved += a; text += b
            k += 1
        a, b = finish((k - 1) & 1); moved += a; text += b
        ctx.fetch_wait(0); ctx.fetch_wait(1)
        return moved, text
    one_pass()                                                # warm-up: allocations, page-locking
    t0 = time.perf_counter()
    moved = text = 0
    for _ in range(steps):
        a, b = one_pass(); moved += a; text += b
    dt = time.perf_counter() - t0
    for cap, p in bufs.values():
        lib.dwgsim_hip_host_free(p)
    ctx.set_gzip(False)
    out = {"value": round(n_pairs * steps / dt / 1e6, 3), "unit": "M read-pairs/s", "gb_per_s": round(moved / dt / 1e9, 2), "steps": steps, "batch_pairs": batch}
    if gz:
        out["text_gb_per_s"] = round(text / dt / 1e9, 2); out["gz_ratio"] = round(moved / max(text, 1), 4)
        out["note"] = "gzip members made on the GPU (k_gzip behind k_simulate) landed in page-locked host memory: what dwgsim-hip writes to its .gz files"
    else:
        out["note"] = "FASTQ text landed in page-locked host memory: simulate_async / wait / fetch_async on two slots, copies on a second stream overlapped with the kernels of the next batch"
    return out


def end_to_end_leg(contigs, flags, n_pairs, gzip_mode="gpu", fai=False, null_sink=False):
    """The dwgsim-hip executable on the same job: FASTA parse, upload, walk, mutation files, reads, gzip (members made on the GPU, or
    zlib on the host cores with DWGSIM_HIP_GZIP=cpu), page-locked copies, the five output files written (to tmpfs when there is one)."""
    from dwgsim_amd import synth
    exe = os.path.join(ROOT, "dwgsim_amd", "dwgsim-hip")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory(dir=_tmpdir()) as t:
        fa = os.path.join(t, "ref.fa")
        synth.write_fasta(fa, contigs)
        if fai:      # the index the reference reads its contig table from (dwgsim.c:465-478): with it the FASTA is streamed, contig by contig
            with open(fa + ".fai", "w") as f:
                off = 0
                for name, arr in contigs:
                    off += len(name) + 2
                    f.write(f"{name}\t{len(arr)}\t{off}\t60\t61\n")
                    off += len(arr) + (len(arr) + 59) // 60
        env = dict(os.environ, DWGSIM_HIP_GZIP=gzip_mode, DWGSIM_HIP_TIMING="1", DWGSIM_HIP_DEVICES="1")
        if null_sink:
            env["DWGSIM_HIP_SINK"] = "null"
        t0 = time.time()
        r = subprocess.run([exe] + flags.split() + [fa, os.path.join(t, "out")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        if r.returncode != 0:
            return {"error": r.stderr.decode(errors="replace")[-300:]}
        gz = sum(os.path.getsize(os.path.join(t, f)) for f in os.listdir(t) if f.endswith(".gz"))
        stages = [ln for ln in r.stderr.decode(errors="replace").splitlines() if ln.startswith("[dwgsim-hip]")]
    gzinfo = ({"where": "gpu", "members": "32 KiB of text each, dynamic Huffman codes + LZ77 matches on the name lines (k_gzip)"} if gzip_mode == "gpu" else
              {"where": "cpu", "threads": effective_cores(), "zlib_level": int(os.environ.get("DWGSIM_HIP_GZIP_LEVEL", "1")), "members": "independent 1 MiB gzip members"})
    return {"seconds": round(dt, 2), "value": round(n_pairs / dt / 1e6, 3), "unit": "M read-pairs/s", "gz_bytes": gz, "gzip": gzinfo,
            "stages": stages[-1][13:] if stages else None,
            "note": "wall time of `dwgsim-hip <flags> ref.fa out` (process start to exit, one GPU), " +
                    ("FASTQ deliveries counted, not written (DWGSIM_HIP_SINK=null)" if null_sink else "outputs on " + ("tmpfs" if _tmpdir() else "the temp dir"))}


def bind_to_device_node(lib, device):
    """This process on the cores of the NUMA node its GPU hangs off (dwgsim_hip_device_numa_node: /sys/bus/pci/devices/<bus id>/numa_node), so that what it
    page-locks and the copies it waits for do not cross the socket link -- as dw_job.cpp does for the job level's workers.  -> the node, or None where it is
    unknown (-1: single-socket boxes, containers without sysfs), where none of its cores may be used by this process, or with DWGSIM_HIP_NO_PIN set."""
    if os.environ.get("DWGSIM_HIP_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        node = int(lib.dwgsim_hip_device_numa_node(device))
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def solo_sweep(args, flags, setup, measure, worlds, lib, dev, numa):
    """Every rank of every world size in `worlds`, ONE AT A TIME on this one GPU (--solo-sweep): what N GPUs would each do, measured -- not a hardware curve.
    weak: the default line's job at coverage 30 W (per-GPU work fixed); strong: BASELINE configs[3], the whole-genome job, split over W ranks.
    T(1) is measured in the same process on the same resident job.  The one thing a real run adds and this cannot show is the host-side all-gather of one
    integer per launch (0.1-0.2 ms per step, beside the kernels of the step before: DESIGN.md 6)."""
    out = {"what": "solo-rank, one GPU at a time: rank r of W runs alone on this GPU -- its walks of every group, its random-read counts (k_place), its launches -- "
                   "with the other ranks' counts taken from a reference pass; efficiency weak = T(1) / max_r T_r(W), strong = T(1) / (W x max_r T_r(W))",
           "numa_node": numa, "weak": {}, "strong": {}}
    K, Wm = args.steps, args.warmup
    for mode in ("weak", "strong"):
        if mode == "strong" and args.no_strong_leg:
            continue
        wl, fl, gbp, k, wm = (args.workload, flags, args.group_bp, K, Wm) if mode == "weak" else ("grch38", FLAGS, STRONG_GROUP_BP, 3, 1)
        S = None
        t1 = None
        for W in [1] + [w for w in worlds if w > 1]:
            if S is None or mode == "weak":      # (a weak job's coverage depends on W: a context of its own; the strong job is made resident once)
                if S is not None:
                    S["ctx"].close()
                S = setup(wl, mode, fl, gbp, W)
            per = []
            for r in range(W):
                m = measure(S, k, wm, W, r, True)
                per.append({"rank": r, "ms_per_step": round(m["elapsed"] / k * 1e3, 4), "pairs": m["my_pairs"], "launches": m["n_my_launches"],
                            "simulate_kernels_ms": round(m["stats"]["sim_kernel_ms"] / k, 4), "walk_gpu_ms": round(m["stats"]["walk_gpu_ms"] / k, 4),
                            "count_random_gpu_ms": round(m["stats"]["count_gpu_ms"] / k, 4), "host_count_random_ms": round(m["stats"]["count_ms"] / k, 4)})
            tmax = max(p["ms_per_step"] for p in per)
            if W == 1:
                t1 = tmax
                out[mode]["job"] = f"{wl}: {len(S['job'])} contig(s), {S['tot_len']} bp, dwgsim {S['job_flags'] if mode == 'strong' else fl + ' (coverage x W)'}, {k} timed steps per rank"
            pairs_all = sum(p["pairs"] for p in per)
            out[mode][str(W)] = {"ranks": per, "max_ms_per_step": tmax, "min_ms_per_step": min(p["ms_per_step"] for p in per),
                                 "value_if_ranks_ran_side_by_side": round(pairs_all / tmax / 1e3, 3), "unit": "M read-pairs/s",
                                 "efficiency": round(t1 / tmax if mode == "weak" else t1 / (W * tmax), 4)}
        S["ctx"].close()
    return out


def make_groups(job, group_bp):
    """consecutive contigs, up to group_bp bases together (a contig that is longer stands alone)"""
    groups, cur, cur_bp = [], [], 0
    for ent in job:
        l = (len(ent[1]) + 4095) // 4096 * 4096
        if cur and cur_bp + l > group_bp:
            groups.append(cur); cur, cur_bp = [], 0
        cur.append(ent); cur_bp += l
    if cur:
        groups.append(cur)
    return groups


def balanced_batches(api, ranges, world, max_pairs):
    """the pairs of a group, in file order, in a multiple of `world` near-equal batches of at most max_pairs (batch b: rank b mod world)"""
    pairs = sum(n for _, _, n in ranges)
    if pairs == 0:
        return []
    nb = world * -(-pairs // (world * max_pairs))
    return list(api.split_ranges(ranges, -(-pairs // nb)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="chr20", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--group-bp", type=int, default=32 << 20, help="contigs are resident together (one walk chain, launches across contigs) up to this many bases")
    ap.add_argument("--ion", action="store_true", help="BASELINE configs[4] flags (Ion Torrent flow model, 400 bp SE, 50x) instead of 2x150 Illumina")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the host_landed / end_to_end legs")
    ap.add_argument("--no-genome-leg", action="store_true", help="skip the end_to_end_genome leg (dwgsim-hip on the whole S4 genome: about a minute, most of it making the synthetic FASTA)")
    ap.add_argument("--flags", default=None, help="analysis only: override the dwgsim flags of the workload (the default is the BASELINE configuration)")
    ap.add_argument("--phases", action="store_true", help="analysis only: print the phase split of the -DDW_PHASE_TIMING build (DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_phases.so)")
    ap.add_argument("--depth", type=int, default=3, help="how many steps ahead the walks run (contigs resident depth + 2 times)")
    ap.add_argument("--no-carry", action="store_true", help="every step waits for its own launches before the next step's first launch is enqueued (the form of rounds 3-4)")
    ap.add_argument("--no-pipeline", action="store_true", help="every step prepares itself (walk, random-read count, exchange) before its first launch, on one resident copy of the contigs")
    ap.add_argument("--share-gpu", action="store_true", help="analysis only: several ranks on one GPU (1-GPU box)")
    ap.add_argument("--strong-leg", action="store_true", help="also measure the fixed whole-genome job (BASELINE configs[3]) split over the ranks: the `strong` object of the line (default with --gpus > 1, and at N = 1 unless --no-legs)")
    ap.add_argument("--no-strong-leg", action="store_true")
    ap.add_argument("--world", type=int, default=0, help="with --solo-rank: the world size W of the run this process plays one rank of")
    ap.add_argument("--solo-rank", type=int, default=None, help="run ALONE, on this one GPU, exactly what rank r of a --world W run does on its GPU: its walks, its random-read counts, its batches; "
                    "the all-gather is replaced by the counts one reference pass recorded (the path has no data-path collective, so a rank's GPU work does not depend on the others running)")
    ap.add_argument("--solo-sweep", default=None, help="e.g. 2,4,8: every rank of every world size, one at a time on this GPU, weak line and strong (whole-genome) job; prints one JSON object with "
                    "T_r(W), T(1) and efficiency = T(1) / max_r T_r(W) (weak), T(1) / (W x max_r T_r(W)) (strong) -- 'solo-rank, one GPU at a time', not a hardware curve")
    args = ap.parse_args()
    solo = args.solo_rank is not None or args.solo_sweep is not None
    if args.solo_rank is not None and not (0 <= args.solo_rank < max(args.world, 1)):
        raise SystemExit("bench.py: --solo-rank r needs --world W with 0 <= r < W")
    args.strong_leg = (args.strong_leg or args.gpus > 1 or not args.no_legs) and not args.no_strong_leg      # (N = 1 carries it too: the curve's efficiency is taken against that line)

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
    if solo and world != 1:
        raise SystemExit("bench.py: --solo-rank / --solo-sweep play one rank at a time in ONE process (no launcher, --gpus 1)")
    if world > torch.cuda.device_count() and not args.share_gpu:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible (--share-gpu is the analysis-only way to put several ranks on one)")
    dev = local_rank % torch.cuda.device_count() if args.share_gpu else local_rank
    torch.cuda.set_device(dev)
    lib = api.load()
    numa = bind_to_device_node(lib, dev)      # this rank's host thread (and what it page-locks) on the cores of its GPU's NUMA node, as the job level's workers are (dw_job.cpp)
    dist = None
    json_out = sys.stdout
    if world > 1:
        # gloo reports its connections on the C library's stdout ("[Gloo] Rank 0 is connected to 1 peer ranks"): file descriptor 1 goes to stderr for the
        # run, and the ONE JSON line is written to what was stdout
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        import torch.distributed as dist
        dist.init_process_group("gloo")          # host-side exchange of integers; no RCCL on this path

    flags = args.flags or (ION_FLAGS if args.ion else FLAGS)
    if args.solo_rank is not None:
        world, rank = args.world, args.solo_rank

    def setup(workload, mode, flags, group_bp, W):
        """The job made resident: the context (the job's flags: a weak job's coverage is W times the workload's) and depth + 2 copies of every group of
        contigs.  Nothing here depends on which rank this process is; measure() assigns the launches."""
        params = api.parse_flags(flags, lib)                  # the workload's own configuration (the legs and baselines run it)
        job_flags = flags
        if mode == "weak" and W > 1:                     # per-GPU work fixed: W times the coverage
            toks = flags.split(); i = toks.index("-C"); toks[i + 1] = repr(float(toks[i + 1]) * W); job_flags = " ".join(toks)
        job_params = api.parse_flags(job_flags, lib)
        contigs = synth.workload_contigs(workload)
        tot_len = sum(len(a) for _, a in contigs)
        ctx = api.Context(job_params, dev, lib)
        if args.phases:
            ctx.debug_option("phases", 1)
        for kv in filter(None, os.environ.get("DWGSIM_BENCH_DEBUG_OPTIONS", "").split(",")):      # analysis only: e.g. split=1,writer=0 (dwgsim_hip_debug_option)
            k, v = kv.split("="); ctx.debug_option(k, int(v))
        # the job: pairs per contig exactly as dwgsim_core schedules them (dwgsim.c:582-590); every contig stays resident, in groups
        job = []
        n_sim = 0
        for ci, (name, arr) in enumerate(contigs):
            n = api.pairs_for_contig(job_params, len(arr), tot_len, ci == len(contigs) - 1, n_sim, lib)
            if n < 0:
                continue
            job.append((name, arr, ci, n))
            n_sim += n
        # Every group is resident depth + 2 TIMES (copies used by consecutive steps in turn): the walk rewrites the haplotypes in place, so the walk + random-read
        # count + exchange of step k+1 can only run beside the kernels of step k on a copy of its own -- the pipeline a job of many groups has anyway
        # (dw_job.cpp: group g+1 is uploaded, walked and counted while the batches of group g run)
        # (ranks that SHARE one GPU -- the readiness runs of the N-rank paths on one-GPU boxes -- hold the whole-genome job once and prepare every step in
        # front of its launches: eight ranks x two or three copies of 17 GB + their output slots are more than the one device has)
        no_pipeline = args.no_pipeline or (args.share_gpu and W > 2 and workload == "grch38")
        raw = []
        for _copy in range(1 if no_pipeline else args.depth + 2):
            gl = []
            for grp in make_groups(job, group_bp):
                h0 = ctx.add_contigs([(name, arr) for name, arr, _, _ in grp], indices=[ci for _, _, ci, _ in grp])
                gl.append({"h0": h0, "members": [(h0 + k, n) for k, (_, _, _, n) in enumerate(grp)]})
            raw.append(gl)
        return dict(ctx=ctx, params=params, job_flags=job_flags, contigs=contigs, tot_len=tot_len, paired=params.length[1] > 0, job=job, job_pairs=sum(e[3] for e in job),
                    raw=raw, no_pipeline=no_pipeline, ref_counts={})

    def measure(S, n_steps, warmup, W, R, alone):
        """One measurement on the resident job S as rank R of W: `warmup` + `n_steps` steps, the barrier-bracketed time of the timed ones (maximum over the
        ranks).  alone: this process plays rank R of W by itself (--solo-rank): no process group; what the all-gather would bring -- the random-read
        counts of the other ranks' launches -- comes from one reference pass over ALL launches, made here before anything is timed (the walk is
        deterministic: every step of every rank sees the same counts); the rank's OWN counts are still taken live, every step, as in the real run, and
        must equal the recorded ones.  -> everything the JSON line is made of."""
        ctx, no_pipeline, depth = S["ctx"], S["no_pipeline"], args.depth
        group_dist = None if alone else dist
        copies = []          # per copy, per group: handle of its first contig, its launches (each a list of ranges), which of them are this rank's
        for gl0 in S["raw"]:
            gl = []
            for g0 in gl0:
                launches = balanced_batches(api, [(h, 0, n) for h, n in g0["members"] if n > 0], W, MAX_LAUNCH_PAIRS)
                gl.append({"h0": g0["h0"], "members": g0["members"], "launches": launches, "mine": [b for b in range(len(launches)) if b % W == R], "pairs": sum(n for l in launches for _, _, n in l)})
            copies.append(gl)
        groups = copies[0]
        my_pairs = sum(n for g in groups for b in g["mine"] for _, _, n in g["launches"][b])
        n_my_launches = sum(len(g["mine"]) for g in groups)
        ref = None
        if alone and W > 1:
            ref = S["ref_counts"].get(W)
            if ref is None:      # the reference pass: every launch of every rank, counted once on the walked copy 0
                ref = []
                for g in groups:
                    ctx.mutate_async(g["h0"]); ctx.mutate_wait(g["h0"])
                    flat = [r for l in g["launches"] for r in l]
                    per = iter(ctx.count_random_ranges(flat, per_range=True) if flat else [])
                    ref.append([sum(next(per) for _ in l) for l in g["launches"]])
                S["ref_counts"][W] = ref

        def barrier():
            torch.cuda.synchronize()
            if group_dist is not None:
                group_dist.barrier()
            torch.cuda.synchronize()

        stats = {"prep_ms": 0.0, "count_ms": 0.0, "exch_ms": 0.0, "sim_kernel_ms": 0.0, "bytes": 0, "n_random": 0, "launches": 0}

        def issue(gl):
            """the walk of every group, enqueued on the walk stream (every rank walks every group itself: deterministic, no broadcast)"""
            for g in gl:
                ctx.mutate_async(g["h0"])

        def prepare(gl, record, issued=False):
            """What a step needs before its first launch: the walks (issue), this rank's random-read counts (k_place, one launch per group, one count
            per batch) and ONE all-gather of them; -> the rand_ii base of every launch of this rank.  All of it on the walk stream / the host: it runs
            beside whatever the compute stream is doing."""
            t0 = time.perf_counter()
            if not issued:
                issue(gl)
            counts = []
            tc = 0.0
            for g in gl:
                ctx.mutate_wait(g["h0"])
                if W > 1:
                    t1 = time.perf_counter()
                    flat = [r for b in g["mine"] for r in g["launches"][b]]
                    per = iter(ctx.count_random_ranges(flat, per_range=True) if flat else [])
                    counts.append([sum(next(per) for _ in g["launches"][b]) for b in g["mine"]])
                    tc += time.perf_counter() - t1
            t2 = time.perf_counter()
            bases = {}
            if W > 1 and alone:      # the exchange, replaced: the others' counts as the reference pass recorded them, mine as just counted
                run = 0
                for q, g in enumerate(gl):
                    for b in range(len(g["launches"])):
                        if b % W == R:
                            bases[(q, b)] = run
                            if counts[q][b // W] != ref[q][b]:
                                raise SystemExit(f"bench.py: rank {R} of {W} counted {counts[q][b // W]} random reads in launch {b} of group {q}, the reference pass {ref[q][b]}")
                        run += ref[q][b]
            elif W > 1:
                width = max(1, max(-(-len(g["launches"]) // W) for g in gl))
                mine_vec = torch.zeros(len(gl) * width, dtype=torch.int64)
                for q in range(len(gl)):
                    for k, cval in enumerate(counts[q]):
                        mine_vec[q * width + k] = cval
                allv = torch.empty(W * mine_vec.numel(), dtype=torch.int64)
                group_dist.all_gather_into_tensor(allv, mine_vec)          # one integer per launch: the only thing that crosses ranks
                allv = allv.view(W, len(gl), width)
                run = 0
                for q, g in enumerate(gl):      # launch b belongs to rank b mod W; its count sits at that rank's position b // W
                    for b in range(len(g["launches"])):
                        if b % W == R:
                            bases[(q, b)] = run
                        run += int(allv[b % W, q, b // W])
            t3 = time.perf_counter()
            if record:
                stats["prep_ms"] += (t2 - t0 - tc) * 1e3; stats["count_ms"] += tc * 1e3; stats["exch_ms"] += (t3 - t2) * 1e3
            return bases

        # launches in flight (slot, accumulator of their step): kept ACROSS steps -- the first launch of step k+1 is enqueued while the last launch of
        # step k still runs, as the batches of consecutive groups are in dw_job.cpp; every launch is waited for inside the timed region
        flight = {"pending": [], "slot": 0}

        def drain(keep, record):
            pending = flight["pending"]
            while len(pending) > keep:
                sl, acc = pending.pop(0)
                b = ctx.wait(sl)
                acc["bytes"] += int(b.bytes[0] + b.bytes[1] + b.bytes[2]); acc["rand"] += int(b.n_random)
                if record:
                    stats["sim_kernel_ms"] += b.sim_kernel_ms; stats["launches"] += 1

        def run(gl, bases, record, then=None, carry=False):
            """all launches of this rank for one step, two in flight; `then` (what is prepared for later steps) runs once the last one is enqueued -- the
            launches of the step before may still run: the walk issued there rewrites the copy of the step before THAT, whose launches were waited
            for before this step's first was enqueued; carry: leave this step's launches in flight for the next step to wait for"""
            acc = {"bytes": 0, "rand": 0}
            first_launch = True
            for q, g in enumerate(gl):
                for b in g["mine"]:
                    drain(1, record)
                    base = bases[(q, b)] if W > 1 else (0 if first_launch else api.RAND_CHAIN)
                    first_launch = False
                    ctx.simulate_ranges_async(g["launches"][b], base, flight["slot"])
                    flight["pending"].append((flight["slot"], acc)); flight["slot"] ^= 1
            nxt = then() if then else None
            if not carry:
                drain(0, record)
            if record:
                stats["acc"] = acc
            return nxt

        def steps(n, record):
            if no_pipeline:
                for _ in range(n):
                    run(copies[0], prepare(copies[0], record), record)
                return
            # depth D: once the launches of step k are enqueued (those of step k - 1 still run), the walks of step k + D go onto the walk stream and the
            # counts / exchange of step k + 1 are finished on the host.  Beside a k_simulate that fills the device nothing else runs
            # (profiles/r05_step_timeline.txt): what is on the walk stream gets the gaps between two launches, a walk needs two of them, the
            # random-read count of the ranks (N > 1) one more.  D = 1 with a wait for the walk before the next launch was the form of rounds 3-4:
            # 0.31 ms between two launches of 5.65 ms; with D = 3 the walk of step k + 1 is finished when step k is enqueued, its count runs in the
            # gap in front of step k, and the launch of step k + 1 is enqueued while step k runs
            D, C = depth, len(copies)
            bases = prepare(copies[0], False)                       # (the first step's preparation; every timed step prepares one successor)
            for d in range(1, D):
                issue(copies[d % C])
            barrier()
            t0 = time.perf_counter()
            for k in range(n):
                far, nx = copies[(k + D) % C], copies[(k + 1) % C]
                def then():
                    if D > 1:
                        issue(far)                                   # the copy step k - 2 read (C = D + 2): its launches were waited for before step k's first was enqueued
                    # (the count of step k + 1 below does not queue behind these walks: it runs on the context's count stream, behind the walk of ITS copy only)
                    return prepare(nx, record, issued=D > 1)
                bases = run(copies[k % C], bases, record, then=then, carry=not args.no_carry)
            drain(0, record)
            for d in range(1, D):                                    # (walks issued for steps beyond the last: waited for, inside the timed region)
                for g in copies[(n + d) % C]:
                    ctx.mutate_wait(g["h0"])
            return t0

        steps(warmup, False)
        barrier()
        gpu_us0 = (ctx.debug_get("walk_us"), ctx.debug_get("count_us"))
        t0 = time.perf_counter()
        t0 = steps(n_steps, True) or t0
        barrier()
        stats["bytes"] = stats["acc"]["bytes"]; stats["n_random"] = stats["acc"]["rand"]      # (of the last step: every step produces the same)
        elapsed = time.perf_counter() - t0
        stats["walk_gpu_ms"] = (ctx.debug_get("walk_us") - gpu_us0[0]) / 1e3
        stats["count_gpu_ms"] = (ctx.debug_get("count_us") - gpu_us0[1]) / 1e3
        total_pairs = my_pairs
        if group_dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64)
            group_dist.all_reduce(tt, op=group_dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            tp = torch.tensor([my_pairs], dtype=torch.int64)
            group_dist.all_reduce(tp)
            total_pairs = int(tp.item())
        out = dict(S)
        out.update(groups=groups, my_pairs=my_pairs, n_my_launches=n_my_launches, stats=stats, elapsed=elapsed, total_pairs=total_pairs)
        return out

    if args.solo_sweep:
        print(json.dumps(solo_sweep(args, flags, setup, measure, [int(x) for x in args.solo_sweep.split(",")], lib, dev, numa)), flush=True)
        return

    m = measure(setup(args.workload, args.mode, flags, args.group_bp, world), args.steps, args.warmup, world, rank, args.solo_rank is not None)
    ctx, params, job_flags, contigs, tot_len, paired, job, job_pairs, groups, my_pairs, n_my_launches, stats, elapsed, total_pairs = (m[k] for k in (
        "ctx", "params", "job_flags", "contigs", "tot_len", "paired", "job", "job_pairs", "groups", "my_pairs", "n_my_launches", "stats", "elapsed", "total_pairs"))
    # the fixed whole-genome job of BASELINE configs[3] split over the ranks, beside the (weak) line: the `strong` object of the line.  Whole-genome
    # groups (two of them: a group holds < 2^31 cells), three timed steps
    strong = None
    landed = {}
    printer = rank == 0 or args.solo_rank is not None
    n_gpus = 1 if args.solo_rank is not None else world
    if world == 1 and not args.no_legs:      # (the legs that need the resident contig: before the context makes room for the whole-genome job)
        g0 = max(groups, key=lambda g: g["pairs"])
        cid0, n0 = max(g0["members"], key=lambda mm: mm[1])          # the contig with the most pairs
        ctx.mutate(cid0)
        landed["host_landed"] = host_landed_leg(api, ctx, cid0, n0)
        landed["host_landed_gz"] = host_landed_leg(api, ctx, cid0, n0, gz=True)
    if args.strong_leg and not args.ion and args.flags is None and not (args.workload == "grch38" and args.mode == "strong"):
        ctx.close(); ctx = None; m["ctx"] = None
        ms = measure(setup("grch38", "strong", FLAGS, STRONG_GROUP_BP, world), 3, 1, world, rank, args.solo_rank is not None)
        ms["ctx"].close()
        if printer:
            Ks = 3
            strong = {"workload": f"S4 (grch38): {len(ms['job'])} contigs, {ms['tot_len']} bp, {ms['job_pairs']} pairs, dwgsim {ms['job_flags']}: ONE job split over {world} rank(s) (batch b of a group: rank b mod {world}); "
                                  f"every rank walks every group ({len(ms['groups'])} groups)",
                      "value": round(ms["total_pairs"] * Ks / ms["elapsed"] / 1e6, 3), "unit": "M read-pairs/s", "n_gpus": n_gpus, "steps": Ks, "ms_per_step": round(ms["elapsed"] / Ks * 1e3, 3),
                      "walk_gpu_ms": round(ms["stats"]["walk_gpu_ms"] / Ks, 3), "count_random_gpu_ms": round(ms["stats"]["count_gpu_ms"] / Ks, 3), "simulate_kernels_ms": round(ms["stats"]["sim_kernel_ms"] / Ks, 3),
                      "kernel_share": round(ms["stats"]["sim_kernel_ms"] / max(ms["elapsed"] * 1e3, 1e-9), 4),
                      "note": "efficiency at N GPUs = this value / (N x the value of the N = 1 line); kernel_share = k_simulate time of rank 0 / step time: what the walks, counts and the exchange leave"}
        ms = None

    if printer:
        K = max(args.steps, 1)
        ms_per_step = elapsed / K * 1e3
        value = total_pairs * K / elapsed / 1e6
        sim_ms_launch = stats["sim_kernel_ms"] / max(stats["launches"], 1)          # average duration of one k_simulate launch
        pairs_per_launch = my_pairs / max(n_my_launches, 1)
        text_per_pair = stats["bytes"] / max(my_pairs, 1)
        if not args.ion and args.flags is None:
            algo_per_pair = ALGO_BYTES_PER_PAIR_2x150
            algo_note = "863 algorithmic B/pair (SURVEY 8d: 713 B of FASTQ written + 150 B of haplotype bases read at 4 bit/base)"
        else:
            algo_per_pair = text_per_pair + (params.length[0] + params.length[1]) / 2.0
            algo_note = f"{algo_per_pair:.0f} algorithmic B per pair/read = {text_per_pair:.1f} B of FASTQ written (measured) + {(params.length[0] + params.length[1]) / 2:.0f} B of haplotype bases read at 4 bit/base"
        achieved = algo_per_pair * pairs_per_launch / (sim_ms_launch * 1e-3) / 1e9 if sim_ms_launch > 0 else 0.0
        # counter evidence (profiles/r06_counters.json, made by tools/make_counters_json.py from rocprofv3 PMC passes) is quoted only when it was
        # taken on exactly the library that is being timed now
        lib_sha = file_sha256(api.LIB_PATH)
        prof, prof_note = {}, None
        try:
            ent = json.load(open(COUNTERS_JSON)).get(f"{args.workload}{'_ion' if args.ion else ''}", {})
            if ent and ent.get("lib_sha256") == lib_sha and ent.get("pairs_per_launch") == int(pairs_per_launch):
                prof = ent
            elif ent:
                prof_note = (f"{os.path.relpath(COUNTERS_JSON, ROOT)} holds counters of another build (library sha256 {str(ent.get('lib_sha256'))[:12]}..., "
                             f"this one {lib_sha[:12]}...) or launch size: not quoted")
        except Exception:
            prof_note = "no counter file for this round"
        sname, cfg_i = WORKLOADS[args.workload]
        seq_kind = "genome-like (GC content, repeat families, microsatellites, homopolymers, soft-masked)" if args.workload.endswith("_like") else "uniform-random"
        out = {
            "metric": "M read-pairs/sec (2x150 bp PE)" if not args.ion else "M reads/sec (Ion Torrent 400 bp SE)", "value": round(value, 3), "unit": "M read-pairs/s" if paired else "M reads/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.mode, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{sname} ({args.workload}): {len(job)} {seq_kind} contig(s) in {len(groups)} resident group(s), {tot_len} bp in all (BASELINE configs[{4 if args.ion else cfg_i}] stand-in), dwgsim {job_flags}, "
                                   f"{job_pairs} pairs per job; step = mutation walk of every contig + all pairs of this rank's read-index ranges, FASTQ text left in HBM; " +
                                   ("every step prepares itself before its first launch" if args.no_pipeline else f"the walk of step k+{args.depth} and the random-read count and exchange of step k+1 run on the walk stream / the host beside the kernels of step k (contigs resident {args.depth + 2} times)"),
                       "pairs_per_gpu_per_step": my_pairs, "launches_per_gpu_per_step": n_my_launches, "fastq_bytes_per_step_per_gpu": stats["bytes"], "fastq_gb_per_s": round(stats["bytes"] * n_gpus * K / elapsed / 1e9, 2),
                       "random_pairs": stats["n_random"],
                       "parallelism": (f"read-index shards x{world} ({args.mode}; batch b of every group's pairs belongs to rank b mod {world}), one host-side all-gather of integers per step" if world > 1 else "one GPU")},
            "breakdown_ms": {"walk_gpu": round(stats["walk_gpu_ms"] / K, 4), "count_random_gpu": round(stats["count_gpu_ms"] / K, 4), "simulate_kernels": round(stats["sim_kernel_ms"] / K, 4),
                             "host_wait_for_walks": round(stats["prep_ms"] / K, 4), "host_count_random": round(stats["count_ms"] / K, 4), "host_exchange": round(stats["exch_ms"] / K, 4),
                             "note": "per step, rank 0.  walk_gpu / count_random_gpu: HIP-event time, first kernel to last, of the walk chains of all groups and of the random-read counts "
                                     "(k_place .. k_range_counts; only N > 1 counts) on the low-priority walk stream" + ("" if args.no_pipeline else " -- they run beside the previous step's k_simulate, so the time includes what that kernel makes them wait") +
                                     "; simulate_kernels: HIP-event time of the k_simulate launches; host_*: wall time the host spent waiting for / calling them"},
            "roofline": {"bound": "valu", "kernel": f"k_simulate<{2 if paired else 1},{[3, 1, 2][params.reads_output_type]},{params.data_type}>",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": prof.get("traffic_bytes_per_launch"),
                         "algorithmic_bytes_per_launch": int(algo_per_pair * pairs_per_launch), "launch_ms": round(sim_ms_launch, 4), "library_sha256": lib_sha,
                         "valu": {k: prof[k] for k in ("valu_instr_per_pair", "valu_instr_per_wave", "salu_instr_per_wave", "valu_issue_active_pct", "source", "git_head") if k in prof},
                         "note": algo_note + " x pairs per launch / HIP-event time of the launch (events on the library's own stream); the kernel is bound by instruction issue "
                                 "(12 k VALU per wave: quality normals, text formatting, Philox) and by what five waves per SIMD leave uncovered of the ordered output's one look-back front, not by HBM: "
                                 "the fraction of the HBM roof says how far that has been squeezed (valu_issue_active_pct = SQ_ACTIVE_INST_VALU, which ticks once per instruction in quad-cycles: DESIGN.md 7b)"},
        }
        if prof_note:
            out["roofline"]["counters_note"] = prof_note
        if args.solo_rank is not None:
            out["solo_rank"] = {"rank": rank, "world": world, "note": "this process ran ALONE on one GPU what rank r of a W-rank run does on its GPU (walks of every group, its random-read "
                                "counts, its launches; the other ranks' counts from a reference pass): `value` is THIS RANK's rate, W ranks side by side would give W times the slowest rank's"}
        if numa is not None:
            out["config"]["numa_node"] = numa
        if strong:
            out["strong"] = strong
        out.update(landed)
        if ctx is not None:
            ctx.close(); ctx = None
        small = [(name, arr) for name, arr, _, _ in job]
        if world == 1 and not args.no_legs and args.workload in ("ecoli", "chr20", "ecoli_like", "chr20_like"):
            out["end_to_end"] = end_to_end_leg(small, flags, job_pairs)
            cpu_gz = end_to_end_leg(small, flags, job_pairs, "cpu")
            if cpu_gz and "seconds" in cpu_gz:
                out["end_to_end"]["with_zlib_on_host"] = {k: cpu_gz[k] for k in ("seconds", "value", "gz_bytes", "gzip")}
        if world == 1 and not args.no_cpu_baseline:
            base_contigs = small[:1] if len(small) == 1 else [c for c in small if c[0] == "chr20"] or small[:1]      # (whole-genome jobs: the chr20-sized contig)
            out["cpu_baseline"] = cpu_baseline(base_contigs, flags)
        if world == 1 and not args.no_legs and not args.no_genome_leg and not args.ion and args.flags is None:
            # steady state of the product: dwgsim-hip on the whole S4 genome (BASELINE configs[3] on one GPU), FASTA streamed through its .fai,
            # 325 M pairs, 232 GB of text -> 115 GB of gzip members landed in host memory and counted (they would not fit a tmpfs)
            g38 = small if args.workload == "grch38" else synth.workload_contigs("grch38")
            job = groups = small = contigs = None
            tl = sum(len(a) for _, a in g38)
            npairs = 0
            for ci, (name, arr) in enumerate(g38):
                npairs += max(api.pairs_for_contig(params, len(arr), tl, ci == len(g38) - 1, npairs, lib), 0)
            out["end_to_end_genome"] = end_to_end_leg(g38, flags, npairs, fai=True, null_sink=True)
        print(json.dumps(out), file=json_out, flush=True)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
