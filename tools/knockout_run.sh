#!/bin/bash
# tools/knockout_run.sh [bits...] -- analysis only (GPU box): kernel time of the chr20-sized launch with each knock-out library, both record writers
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 0 ${@:-2 64}; do
  L=dwgsim_amd/libdwgsim_hip_knock$k.so; [ $k = 0 ] && L=dwgsim_amd/libdwgsim_hip.so
  for w in 1 0; do
    echo -n "knock $k "; DWGSIM_HIP_LIB=$L WRITER=$w python tools/time_probe.py "-z 13 -1 150 -2 150 -C 30 -o 1" 2>/dev/null
  done
done
