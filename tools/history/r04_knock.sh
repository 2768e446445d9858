#!/bin/bash
# tools/r04_knock.sh -- analysis only (gpurun): kernel time of the chr20-sized launch with parts of k_simulate<2,1,0> switched off (libraries of
# tools/knockout_build.sh), at 2 x 150 and 2 x 50 bp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 0 "$@"; do
  L=dwgsim_amd/libdwgsim_hip_knock$k.so; [ $k = 0 ] && L=dwgsim_amd/libdwgsim_hip.so
  echo -n "knock $k "; DWGSIM_HIP_LIB=$L python tools/time_probe.py "-z 13 -1 150 -2 150 -C 30 -o 1" 2>/dev/null
  echo -n "knock $k "; DWGSIM_HIP_LIB=$L python tools/time_probe.py "-z 13 -1 50 -2 50 -C 10 -o 1" 2>/dev/null
done
