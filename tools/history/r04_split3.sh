#!/bin/bash
# analysis only (gpurun): where the two-kernel form of the Illumina read kernel wins: read lengths, single end, launch sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fl in "-z 13 -1 75 -2 75 -C 15 -o 1" "-z 13 -1 100 -2 100 -C 20 -o 1" "-z 13 -1 125 -2 125 -C 25 -o 1" "-z 13 -1 150 -2 0 -C 30 -o 1" "-z 13 -1 100 -2 0 -C 20 -o 1" "-z 13 -1 36 -2 36 -C 8 -o 1" "-z 13 -1 100 -2 100 -C 20 -o 0" "-z 13 -1 50 -2 50 -C 10 -o 2"; do
  for sp in 0 1; do echo -n "split $sp "; SPLIT=$sp python tools/time_probe.py "$fl" 2>/dev/null; done
done
for fl in "-z 13 -1 100 -2 100 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 50 -2 50 -C 30 -o 1"; do
for sp in 0 1; do echo -n "ecoli split $sp "; WL=ecoli SPLIT=$sp python tools/time_probe.py "$fl" 2>/dev/null; done
done
