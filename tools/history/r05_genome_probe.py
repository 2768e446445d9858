"""tools/r05_genome_probe.py -- analysis only (gpurun): the product on the whole S4 genome (3.09 Gb, 30 x, counting sink: what bench.py's
end_to_end_genome leg runs) -- wall and stage times for a few batch sizes / group sizes, then ONE run under
`rocprofv3 --kernel-trace --memory-copy-trace` and the busy fraction of the copy engine and of the kernels over the run (tools/copy_busy.py)."""
import os, sys, subprocess, tempfile, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dwgsim_amd import synth
contigs = synth.workload_contigs("grch38")
exe = os.path.join(ROOT, "dwgsim_amd", "dwgsim-hip")
flags = "-z 13 -1 150 -2 150 -C 30 -o 1"
variants = [v for v in os.environ.get("PROBE_VARIANTS", "default").split(";") if v]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as t:
    fa = os.path.join(t, "ref.fa")
    synth.write_fasta(fa, contigs)
    with open(fa + ".fai", "w") as f:
        off = 0
        for name, arr in contigs:
            off += len(name) + 2
            f.write(f"{name}\t{len(arr)}\t{off}\t60\t61\n")
            off += len(arr) + (len(arr) + 59) // 60
    del contigs
    def run(extra_env, label, prefix=()):
        env = dict(os.environ, DWGSIM_HIP_TIMING="1", DWGSIM_HIP_DEVICES="1", DWGSIM_HIP_SINK="null")
        env.update(extra_env)
        t0 = time.time()
        r = subprocess.run(list(prefix) + [exe] + flags.split() + [fa, os.path.join(t, "out")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        st = [ln for ln in r.stderr.decode(errors="replace").splitlines() if ln.startswith("[dwgsim-hip]")]
        ck = [ln for ln in r.stderr.decode(errors="replace").splitlines() if ln.startswith("[dwgsim-hip-clock]")]
        if ck:
            import re; a, b = (float(v) for v in re.findall(r"at ([0-9.]+)", ck[-1])[:2])
            print(f"    exec -> main {a - t0:.3f} s | main {b - a:.3f} s | output complete -> process gone {t0 + dt - b:.3f} s")
        if env.get("DWGSIM_HIP_TRACE"):
            print("\n".join(ln for ln in r.stderr.decode(errors="replace").splitlines() if ln.startswith("[trace")))
        print(f"{label:40s} wall {dt:.2f} s  rc {r.returncode}  {st[-1][13:] if st else r.stderr.decode(errors='replace')[-300:]}", flush=True)
    for v in variants:
        env = {}
        if v != "default":
            for kv in v.split(","):
                k, val = kv.split("=", 1); env[k] = val
        for rep in range(2):
            run(env, v)
    if os.environ.get("PROBE_TRACE", "1") == "1":
        out = os.path.join(ROOT, "gpurun_out", "genome_trace"); os.makedirs(out, exist_ok=True)
        env = {"DWGSIM_HIP_TEARDOWN": "1"}      # (the product leaves through _exit: the profiler writes its database in an exit handler)
        tv = os.environ.get("PROBE_TRACE_VARIANT", "default")
        if tv != "default":
            for kv in tv.split(","):
                k, val = kv.split("=", 1); env[k] = val
        run(env, "under rocprofv3 (" + tv + ")", prefix=["rocprofv3", "--kernel-trace", "--memory-copy-trace", "-d", out, "--"])
        for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "copy_busy.py"), db])
            os.remove(db)
