#!/bin/bash
# tools/r04_prio.sh -- analysis only (gpurun): wave priority by phase in k_simulate<2,*,0> (s_setprio 3 before the record lengths are published, 0 after;
# and the other way round) against the product; variant objects are built on the box from sed-ed copies of dw_simulate.hip
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_prio; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
cd dwgsim_amd/csrc; mkdir -p build/knock
for v in hi_first lo_first all3 hi1 lo1; do
  case $v in hi_first) A=3; B=0;; lo_first) A=0; B=3;; all3) A=3; B=3;; hi1) A=1; B=0;; lo1) A=0; B=1;; esac
  sed "s|    DW_PROBE_MARK(a, 0);     // ticket, fixed strings|    __builtin_amdgcn_s_setprio($A); DW_PROBE_MARK(a, 0);|; s|    DW_PROBE_MARK(a, 3);     // name lengths, block scan, look-back|    __builtin_amdgcn_s_setprio($B); DW_PROBE_MARK(a, 3);|" dw_simulate.hip > build/knock/dw_simulate_$v.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -I. -O3 -std=c++17 -ffp-contract=off -fPIC -DDW_PART=1 -c build/knock/dw_simulate_$v.hip -o build/knock/s1_$v.o &
done; wait
for v in hi_first lo_first all3 hi1 lo1; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/gzip.o build/host.o build/mutin.o build/job.o build/s0.o build/knock/s1_$v.o build/s[2-9].o build/s10.o -lpthread -o ../libdwgsim_hip_knock_$v.so; done
cd ../..
for v in product hi_first lo_first all3 hi1 lo1 product; do
  lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_knock_$v.so
  for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 100 -2 100 -C 30 -o 1"; do DWGSIM_HIP_LIB=$lib SPLIT=0 timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$v /"; done
done | tee $o/prio.txt
