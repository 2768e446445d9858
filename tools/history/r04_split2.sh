#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in b5 b6 ""; do
  L=dwgsim_amd/libdwgsim_hip_var_$v.so; [ -z "$v" ] && L=dwgsim_amd/libdwgsim_hip.so
  for wr in 1 0; do echo -n "lib ${v:-b8} split 1 writer $wr "; DWGSIM_HIP_LIB=$L SPLIT=1 WRITER=$wr python tools/time_probe.py "-z 13 -1 150 -2 150 -C 30 -o 1" 2>/dev/null; done
done
