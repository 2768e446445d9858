"""tools/time_probe.py <flags> -- analysis only: kernel-resident time of whole-contig chr20 simulate calls with the given dwgsim flags (library: DWGSIM_HIP_LIB)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dwgsim_amd import api, synth
lib = api.load()
flags = sys.argv[1]
contigs = synth.workload_contigs(os.environ.get("WL", "chr20"))
params = api.parse_flags(flags, lib)
with api.Context(params, 0, lib) as ctx:
    if os.environ.get("SIMT"):
        ctx.debug_option("sim_threads", int(os.environ["SIMT"]))
    if os.environ.get("FLOW_SLOTS"):
        ctx.debug_option("flow_slots", int(os.environ["FLOW_SLOTS"]))
    if os.environ.get("SPLIT"):
        ctx.debug_option("split", int(os.environ["SPLIT"]))
    if os.environ.get("WRITER"):
        ctx.debug_option("writer", int(os.environ["WRITER"]))
    name, arr = contigs[0]
    cid = ctx.add_contig(name, arr, 0)
    ctx.mutate(cid)
    n = api.pairs_for_contig(params, len(arr), len(arr), True, 0, lib)
    ctx.simulate(cid, 0, n, 0, 0)
    t = time.time()
    for i in range(10):
        b = ctx.simulate(cid, 0, n, 0, 0)
    dt = (time.time() - t) / 10
    print(os.environ.get("WRITER", "-"), "%-50s %9d pairs  %8.3f ms  %7.1f M pairs/s  %6.1f GB/s" % (flags, n, dt * 1e3, n / dt / 1e6, sum(b.bytes) / dt / 1e9), flush=True)
    if os.environ.get("PLACE"):
        ctx.count_random(cid, 0, n)
        t = time.time()
        for i in range(10):
            r = ctx.count_random(cid, 0, n)
        dt = (time.time() - t) / 10
        print("count_random (k_summarize once + k_place + scan): %8.3f ms  random=%d" % (dt * 1e3, r), flush=True)
