#!/bin/bash
# tools/r03_ion_probe.sh -- analysis only (gpurun): the Ion Torrent launch (400-bp single-end reads) of the product library and of every
# dwgsim_amd/libdwgsim_hip_var_ion*.so (tools/variant_build.sh ion<name> "<flags>" 4).  Knock-out variants write garbage by construction.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.01 -o 1"
for wl in ecoli chr20; do
  echo "== $wl"
  WL=$wl python tools/time_probe.py "$F" 2>&1 | tail -1
  WL=$wl python tools/time_probe.py "$F -e 0.000001" 2>&1 | tail -1
  for v in dwgsim_amd/libdwgsim_hip_var_ion*.so; do [ -e $v ] || continue; echo -n "$(basename $v .so | sed s/libdwgsim_hip_var_//) "; WL=$wl DWGSIM_HIP_LIB=$v python tools/time_probe.py "$F" 2>&1 | tail -1; done
done
