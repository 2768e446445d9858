#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r03_b6; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run n1 --steps 50
run n1_nopipe --steps 50 --no-pipeline
run n2_weak --gpus 2 --share-gpu --steps 20
run n2_weak_nopipe --gpus 2 --share-gpu --steps 20 --no-pipeline
run n2_strong_mini --gpus 2 --share-gpu --mode strong --workload grch38_mini --steps 10
run assembly --workload assembly5k --steps 20
run ecoli --workload ecoli --steps 50
run grch38 --workload grch38 --steps 3 --warmup 1
run n2_strong_grch38 --gpus 2 --share-gpu --mode strong --workload grch38 --steps 2 --warmup 1
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"], d["config"]["launches_per_gpu_per_step"])
except Exception as e: print("ERR",e)
PY
done
