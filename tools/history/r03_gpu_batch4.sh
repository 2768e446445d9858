#!/bin/bash
# tools/r03_gpu_batch4.sh -- analysis only (gpurun): timing of the new extraction + the parity suites that exercise it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r03_b4; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
for f in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 1 -Q 0" "-z 13 -1 50 -2 50 -C 10 -o 1"; do for w in 1 0; do WRITER=$w python tools/time_probe.py "$f" 2>/dev/null; done; done | tee $o/probe.txt
timeout 300 python bench.py --steps 50 --no-legs --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err
timeout 300 python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline > $o/bench_ecoli.json 2> $o/bench_ecoli.err
timeout 300 python bench.py --ion --workload ecoli --steps 20 --no-legs --no-cpu-baseline > $o/bench_ion.json 2> $o/bench_ion.err
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, d["breakdown_ms"]["simulate_kernels"], d["roofline"]["frac"])
except Exception as e: print("ERR",e)
PY
done
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -4 $o/pytest_gpu.log
