"""tools/r04_ingest_probe.py -- analysis only: what the FASTA ingest of dwgsim-hip costs on its own.  The whole S4 genome (3.09 Gb FASTA on tmpfs, with a
.fai) through `dwgsim-hip -C 0.01` (32 k pairs: the GPU side is the walk of every contig and next to no reads), stage times for 1 / 4 / 16 / all reader
threads (DWGSIM_HIP_READ_THREADS); then the same job at 30 x with the counting sink (what bench.py's end_to_end_genome leg runs)."""
import os, sys, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dwgsim_amd import synth
contigs = synth.workload_contigs("grch38")
exe = os.path.join(ROOT, "dwgsim_amd", "dwgsim-hip")
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as t:
    fa = os.path.join(t, "ref.fa")
    synth.write_fasta(fa, contigs)
    with open(fa + ".fai", "w") as f:
        off = 0
        for name, arr in contigs:
            off += len(name) + 2
            f.write(f"{name}\t{len(arr)}\t{off}\t60\t61\n")
            off += len(arr) + (len(arr) + 59) // 60
    for flags, label in (("-z 13 -1 150 -2 150 -C 0.01 -o 1", "ingest + walks"), ("-z 13 -1 150 -2 150 -C 30 -o 1", "whole job, 30 x")):
        for thr in ("1", "4", "16", ""):
            if label.startswith("whole") and thr not in ("", "1"):
                continue
            env = dict(os.environ, DWGSIM_HIP_TIMING="1", DWGSIM_HIP_DEVICES="1", DWGSIM_HIP_SINK="null")
            if thr:
                env["DWGSIM_HIP_READ_THREADS"] = thr
            t0 = time.time()
            r = subprocess.run([exe] + flags.split() + [fa, os.path.join(t, "out")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.time() - t0
            st = [ln for ln in r.stderr.decode(errors="replace").splitlines() if ln.startswith("[dwgsim-hip]")]
            print(f"{label:16s} reader threads {thr or 'all':>3s}: wall {dt:.2f} s  rc {r.returncode}  {st[-1][13:] if st else r.stderr.decode(errors='replace')[-200:]}", flush=True)
