"""analysis only: host-landed rate of dwgsim_hip_fetch (pageable vs pinned destination) for the default workload."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
from dwgsim_amd import api, synth
lib = api.load()
params = api.parse_flags("-z 13 -1 150 -2 150 -C 30 -o 1", lib)
name, arr = synth.workload_contigs("ecoli")[0]
n = api.pairs_for_contig(params, len(arr), len(arr), False, 0, lib)
ctx = api.Context(params, 0, lib)
cid = ctx.add_contig(name, arr, 0)
ctx.mutate(cid)
b = ctx.simulate(cid, 0, n, 0, 0)
nbytes = [int(b.bytes[0]), int(b.bytes[1])]
for kind in ("pageable", "pinned"):
    bufs = [torch.empty(x, dtype=torch.uint8) for x in nbytes]
    if kind == "pinned":
        bufs = [t.pin_memory() for t in bufs]
    for rep in range(3):
        t0 = time.perf_counter()
        for s in (0, 1):
            rc = lib.dwgsim_hip_fetch(ctx.h, 0, s, C.c_void_p(bufs[s].data_ptr()), C.c_size_t(nbytes[s]))
            assert rc == 0
        dt = time.perf_counter() - t0
    tot = sum(nbytes)
    print(f"{kind:9s} {tot / dt / 1e9:6.1f} GB/s  -> {n / dt / 1e6:6.1f} M pairs/s host-landed ceiling ({dt * 1e3:.1f} ms for {tot / 1e6:.0f} MB)")
    assert bytes(bufs[0][:64].numpy().tobytes()).startswith(b"@")
