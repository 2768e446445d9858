#!/bin/bash
# quick A/B: parity subset + bench chr20 / ecoli / ion, kernel-only
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for wl in chr20 ecoli; do timeout 300 python bench.py --workload $wl --steps 30 --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['breakdown_ms'])"; done
timeout 300 python bench.py --workload ecoli --ion --steps 20 --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ion', d['value'], d['ms_per_step'], d['breakdown_ms'])"
