#!/bin/bash
# tools/r03_gpu_batch1.sh -- analysis only (gpurun): round-3 baseline of the round-2 kernels on this round's box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r03_b1; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 300 python bench.py --steps 50 > $o/bench_default.json 2> $o/bench_default.err
timeout 300 python bench.py --gpus 2 --share-gpu --steps 20 --no-legs --no-cpu-baseline > $o/bench_n2_share.json 2> $o/bench_n2_share.err
timeout 400 python bench.py --workload grch38 --steps 3 --warmup 1 --no-legs --no-cpu-baseline > $o/bench_grch38.json 2> $o/bench_grch38.err
timeout 300 python bench.py --workload grch38_mini --steps 20 --no-legs --no-cpu-baseline > $o/bench_mini.json 2> $o/bench_mini.err
tail -c 600 $o/*.err
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","n_gpus","ms_per_step","breakdown_ms")}, d["roofline"]["frac"], d.get("host_landed",{}).get("value"), (d.get("end_to_end") or {}).get("seconds"))
except Exception as e: print("ERR",e)
PY
done
