#!/bin/bash
# GPU gzip: tests, then end-to-end timing of both gzip modes on chr20
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cli or gzip" 2>&1 | tail -6 > gpurun_out/t_gz.txt; cat gpurun_out/t_gz.txt
timeout 600 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_gz.json 2> gpurun_out/bench_gz.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_gz.json").read().strip().splitlines()[-1])
print(d["value"], json.dumps(d.get("host_landed")), json.dumps(d.get("end_to_end"), indent=1))
PY
tail -3 gpurun_out/bench_gz.err
