"""analysis / test helper: the E. coli-sized contig through several option mixes, whole outputs compared with the oracle (Philox mode) by hash."""
import hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dwgsim_amd import api, synth

lib = api.load()
contigs = synth.workload_contigs("ecoli")
oracle = os.path.join(ROOT, "oracle", "build", "dwgsim_oracle")
bad = 0
with tempfile.TemporaryDirectory() as t:
    fa = os.path.join(t, "e.fa"); synth.write_fasta(fa, contigs)
    for flags in sys.argv[1:]:
        t0 = time.time()
        subprocess.run([oracle, "--rng", "philox"] + flags.split() + [fa, os.path.join(t, "o")], check=True, stderr=subprocess.DEVNULL)
        t1 = time.time()
        res = api.run_job(api.parse_flags(flags, lib), contigs, lib=lib)
        ok = True
        for k, suf in ((0, "bwa.read1.fastq"), (1, "bwa.read2.fastq"), (2, "bfast.fastq")):
            p = os.path.join(t, "o." + suf)
            want = open(p, "rb").read() if os.path.exists(p) else b""
            ok &= hashlib.sha256(res.streams[k]).digest() == hashlib.sha256(want).digest()
            if os.path.exists(p): os.remove(p)
        ok &= res.mutations_vcf == open(os.path.join(t, "o.mutations.vcf"), "rb").read()
        bad += not ok
        print(("OK  " if ok else "BAD ") + flags, f"pairs {res.n_pairs} oracle {t1 - t0:.0f} s", flush=True)
print("bad:", bad)
