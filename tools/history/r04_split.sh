#!/bin/bash
# analysis only (gpurun): k_simulate as one kernel with waiting look-backs (SPLIT=0), as two kernels (1), as one kernel with deferred text (3, LAG tiles behind)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernel_parity or writers or resident" 2>&1 | tail -3
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 50 -2 50 -C 10 -o 1"; do
  for sp in 0 1 3; do echo -n "split $sp "; SPLIT=$sp python tools/time_probe.py "$fl" 2>/dev/null; done
  for lag in 0 256 1024 1536 4096 8192; do echo -n "split 3 lag $lag "; SPLIT=3 LAG=$lag python tools/time_probe.py "$fl" 2>/dev/null; done
done
for fl in "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 150 -2 0 -C 30 -o 1" "-z 13 -1 250 -2 250 -C 30 -o 1"; do
  for sp in 0 3; do echo -n "split $sp "; SPLIT=$sp python tools/time_probe.py "$fl" 2>/dev/null; done
done
for sp in 0 1 3; do echo -n "ecoli split $sp "; WL=ecoli SPLIT=$sp python tools/time_probe.py "-z 13 -1 150 -2 150 -C 30 -o 1" 2>/dev/null; done
