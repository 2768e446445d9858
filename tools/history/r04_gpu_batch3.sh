#!/bin/bash
# tools/r04_gpu_batch3.sh -- analysis only (gpurun): summaries written with the views, stream priorities, GPU time of walks / counts in the line;
# the N = 8 line on one GPU (eight ranks sharing it: readiness of the 8-rank path, not a speed-up)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b3; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "count_random or job_level or several_contexts or two_ranks or whole_node or kernel_parity" > $o/pytest.log 2>&1; tail -5 $o/pytest.log
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run n1 --steps 50
run n2_weak --gpus 2 --share-gpu --steps 20
run n8_weak_share --gpus 8 --share-gpu --steps 5 --warmup 2
run n1_strong_grch38 --mode strong --workload grch38 --steps 2 --warmup 1
run n2_strong_grch38 --gpus 2 --share-gpu --mode strong --workload grch38 --steps 2 --warmup 1
run n8_strong_grch38_share --gpus 8 --share-gpu --mode strong --workload grch38 --steps 1 --warmup 1
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
except Exception as e: print("ERR",e)
PY
done
