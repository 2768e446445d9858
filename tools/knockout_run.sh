#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 0 2 64; do
  L=dwgsim_amd/libdwgsim_hip_knock$k.so; [ $k = 0 ] && L=dwgsim_amd/libdwgsim_hip.so
  DWGSIM_HIP_LIB=$L python bench.py --workload chr20 --no-legs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('knock %2d  kernel %8.3f ms' % ($k, d['breakdown_ms']['simulate_kernels']))"
done
