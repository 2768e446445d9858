#!/bin/bash
# tools/r03_gpu_batch3.sh -- analysis only (gpurun): new bench.py lines (groups, legs, N=2 sharing the GPU) + the new GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r03_b3; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
run() { name=$1; shift; timeout 900 python bench.py "$@" > $o/$name.json 2> $o/$name.err; tail -c 400 $o/$name.err | grep -v amdgpu.ids | tail -3; }
run bench_default --steps 50
run bench_assembly --workload assembly5k --steps 20 --no-legs --no-cpu-baseline
run bench_mini --workload grch38_mini --steps 20 --no-legs --no-cpu-baseline
run bench_grch38 --workload grch38 --steps 3 --warmup 1 --no-legs --no-cpu-baseline
run bench_n2_weak --gpus 2 --share-gpu --steps 20 --no-legs --no-cpu-baseline
run bench_n2_strong_mini --gpus 2 --share-gpu --mode strong --workload grch38_mini --steps 10 --no-legs --no-cpu-baseline
run bench_n2_strong_grch38 --gpus 2 --share-gpu --mode strong --workload grch38 --steps 2 --warmup 1 --no-legs --no-cpu-baseline
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, d["breakdown_ms"], d["roofline"]["frac"], d["config"]["launches_per_gpu_per_step"])
    for k in ("host_landed","host_landed_gz","end_to_end","end_to_end_genome"):
        if k in d: print("  ",k, {q:d[k].get(q) for q in ("value","gb_per_s","seconds","stages","error")})
except Exception as e: print("ERR",e)
PY
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "job_level or two_ranks or cli or two_hundred" > $o/pytest_new.log 2>&1; tail -5 $o/pytest_new.log
