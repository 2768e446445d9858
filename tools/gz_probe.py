"""tools/gz_probe.py -- analysis only: GPU gzip on one chr20 batch: correctness of the members (gunzip == text) and the time the gzip kernels add."""
import gzip, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dwgsim_amd import api, synth
lib = api.load()
contigs = synth.workload_contigs("chr20")
params = api.parse_flags("-z 13 -1 150 -2 150 -C 30 -o 1", lib)
with api.Context(params, 0, lib) as ctx:
    name, arr = contigs[0]
    cid = ctx.add_contig(name, arr, 0)
    ctx.mutate(cid)
    n = 1 << 20
    for on in (False, True, False, True):
        ctx.set_gzip(on)
        ctx.simulate(cid, 0, n, 0, 0)
        t = time.time()
        for i in range(5):
            b = ctx.simulate(cid, i * n, n, 0, 0)
        dt = (time.time() - t) / 5
        print("gzip", on, "ms per 1M-pair batch %.3f" % (dt * 1e3), "text", list(b.bytes), "gz", list(b.gz_bytes), flush=True)
    for s in range(2):
        txt = ctx.fetch(0, s, b.bytes[s]); gz = ctx.fetch_gz(0, s, b.gz_bytes[s])
        t = time.time(); back = gzip.decompress(gz); print("gunzip s", time.time() - t)
        print("stream", s, "equal", back == txt, "ratio %.3f" % (len(gz) / len(txt)), flush=True)
        assert back == txt
