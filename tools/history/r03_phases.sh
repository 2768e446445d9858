#!/bin/bash
# tools/r03_phases.sh -- analysis only (gpurun): phase split of k_simulate (the -DDW_PHASE_TIMING build) and the cost of count_random
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r03_phases; mkdir -p $o
DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_phases.so timeout 300 python bench.py --phases --steps 3 --warmup 1 --no-legs --no-cpu-baseline > $o/phases.json 2> $o/phases.err
grep phases $o/phases.err | tail -3
python - <<'PY'
import sys,time; sys.path.insert(0,'.')
from dwgsim_amd import api, synth
lib=api.load(); p=api.parse_flags("-z 13 -1 150 -2 150 -C 30 -o 1",lib)
ctx=api.Context(p,0,lib); name,arr=synth.workload_contigs("chr20")[0]
h=ctx.add_contig(name,arr,0); ctx.mutate(h)
n=6783597
for k in range(3):
    t=time.perf_counter(); c=ctx.count_random(h,0,n); dt=time.perf_counter()-t
    print(f"count_random {n} pairs: {dt*1e3:.3f} ms -> {c}")
PY
