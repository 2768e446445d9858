#!/bin/bash
# tools/knockout_build.sh -- analysis only: libraries with parts of k_simulate<2,1,0> switched off (-DDW_KNOCK bits: 1 no text assembly at all,
# 2 text assembled but not stored, 4 no error tests, 8 no base extraction, 16 no header), to weigh the parts of the VALU-bound kernel.
# Output is garbage by construction; only the kernel time is meaningful.  Run the result with tools/knockout_run.sh on the GPU box.
set -e
cd "$(dirname "$0")/../dwgsim_amd/csrc"
mkdir -p build/knock
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $F -c dw_walk.hip -o build/knock/walk.o &
/opt/rocm/bin/hipcc $F -c dw_host.cpp -o build/knock/host.o &
/opt/rocm/bin/hipcc $F -c dw_mutin.cpp -o build/knock/mutin.o &
/opt/rocm/bin/hipcc $F -c dw_gzip.hip -o build/knock/gzip.o &
/opt/rocm/bin/hipcc $F -DDW_PART=0 -c dw_simulate.hip -o build/knock/s0.o &
for p in 2 3 4 5 6 7 8; do /opt/rocm/bin/hipcc $F -DDW_PART=$p -c dw_simulate.hip -o build/knock/s$p.o & done
wait
for k in 2 64; do
  /opt/rocm/bin/hipcc $F -DDW_PART=1 -DDW_KNOCK=$k -c dw_simulate.hip -o build/knock/s1_k$k.o &
done
wait
for k in 2 64; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/knock/walk.o build/knock/gzip.o build/knock/host.o build/knock/mutin.o build/knock/s0.o build/knock/s1_k$k.o build/knock/s[2-8].o -o ../libdwgsim_hip_knock$k.so
done
echo built knock libs
