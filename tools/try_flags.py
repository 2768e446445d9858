"""analysis only: run one flag set through run_job on the ecoli contig and report the outcome (errors included)."""
import sys
sys.path.insert(0, ".")
from dwgsim_amd import api, synth
lib = api.load()
contigs = synth.workload_contigs("ecoli")
for flags in sys.argv[1:]:
    try:
        res = api.run_job(api.parse_flags(flags, lib), contigs, lib=lib)
        print("OK  ", flags, "pairs", res.n_pairs, "bytes", [len(res.streams[k]) for k in range(3)])
    except Exception as e:
        print("FAIL", flags, "->", repr(e)[:300])
