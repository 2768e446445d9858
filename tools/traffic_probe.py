"""tools/traffic_probe.py <flags> -- analysis only: a few whole-contig chr20 simulate calls with the given dwgsim flags (run under rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dwgsim_amd import api, synth
lib = api.load()
flags = sys.argv[1]
contigs = synth.workload_contigs("chr20")
params = api.parse_flags(flags, lib)
with api.Context(params, 0, lib) as ctx:
    name, arr = contigs[0]
    cid = ctx.add_contig(name, arr, 0)
    ctx.mutate(cid)
    n = api.pairs_for_contig(params, len(arr), len(arr), True, 0, lib)
    for i in range(3):
        b = ctx.simulate(cid, 0, n, 0, 0)
    print(flags, n, list(b.bytes))
