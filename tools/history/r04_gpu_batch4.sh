#!/bin/bash
# tools/r04_gpu_batch4.sh -- analysis only (gpurun): what the walk of the whole S4 genome costs on its own (no k_simulate beside it), kernel by kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b4; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats -d $o/kt -- python bench.py --workload grch38 --mode strong --no-pipeline --steps 2 --warmup 1 --no-legs --no-cpu-baseline > $o/grch38_nopipe.json 2> $o/kt.log
for db in $(find $o/kt -name '*.db' | head -1); do python tools/rocprof_summary.py $db | head -40; done > $o/kt_summary.txt 2>&1
python - <<'PY'
import time, ctypes, sys
sys.path.insert(0, ".")
from dwgsim_amd import api
lib = api.load()
lib.dwgsim_hip_host_alloc.restype = ctypes.c_void_p
for mb in (64, 320, 320, 320):
    t0 = time.perf_counter(); p = lib.dwgsim_hip_host_alloc(mb << 20); t1 = time.perf_counter()
    print(f"hipHostMalloc {mb} MB: {1e3 * (t1 - t0):.1f} ms", flush=True)
PY
cat $o/kt_summary.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_b4/grch38_nopipe.json").read().strip().splitlines()[-1])
b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"})
PY
