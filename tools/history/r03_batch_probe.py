"""tools/r03_batch_probe.py -- analysis only (gpurun): dwgsim-hip wall time on the chr20-sized job and the whole S4 genome for several batch sizes (DWGSIM_HIP_BATCH)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dwgsim_amd import synth, api
lib = api.load()
for wl, null in (("chr20", False), ("grch38", True)):
    contigs = synth.workload_contigs(wl)
    flags = bench.ILLUMINA_FLAGS if hasattr(bench, "ILLUMINA_FLAGS") else "-z 13 -1 150 -2 150 -C 30 -o 1"
    params = api.parse_flags(flags, lib)
    tot = sum(len(a) for _, a in contigs)
    n = sum(api.pairs_for_contig(params, len(a), tot, True, 0, lib) for _, a in contigs)
    for bp in (0, 1 << 20):
        os.environ.pop("DWGSIM_HIP_BATCH", None)
        if bp: os.environ["DWGSIM_HIP_BATCH"] = str(bp)
        for rep in range(2 if wl == "chr20" else 1):
            r = bench.end_to_end_leg(contigs, flags, n, fai=(wl != "chr20"), null_sink=null)
            print(wl, "batch", bp, r["seconds"], "s", r["value"], "M pairs/s |", r["stages"], flush=True)
