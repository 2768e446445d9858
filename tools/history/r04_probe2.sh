#!/bin/bash
# analysis only (gpurun): what retries near N runs cost (the same launch with every N accepted), E. coli-sized vs chr20-sized contigs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 1 -n 1000" "-z 13 -1 150 -2 150 -C 30 -o 1 -y 0" "-z 13 -1 150 -2 150 -C 30 -o 1 -n 1000 -y 0"; do
  for k in 0 2048; do
  L=dwgsim_amd/libdwgsim_hip_knock$k.so; [ $k = 0 ] && L=dwgsim_amd/libdwgsim_hip.so
  echo -n "knock $k "; DWGSIM_HIP_LIB=$L python tools/time_probe.py "$fl" 2>/dev/null
  done
done
