#!/bin/bash
# tools/r03_gpu_batch2.sh -- analysis only (gpurun): the GPU suite + bench lines after the group refactor
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r03_b2; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 300 python bench.py --steps 50 --no-legs --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err
timeout 300 python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline > $o/bench_ecoli.json 2> $o/bench_ecoli.err
timeout 300 python bench.py --workload grch38_mini --steps 20 --no-legs --no-cpu-baseline > $o/bench_mini.json 2> $o/bench_mini.err
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","n_gpus","ms_per_step","breakdown_ms")}, d["roofline"]["frac"])
except Exception as e: print("ERR",e)
PY
done
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -5 $o/pytest_gpu.log
