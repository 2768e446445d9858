#!/bin/bash
# tools/knockout_build.sh [bits...] -- analysis only: libraries with parts of k_simulate<2,*,0> switched off (-DDW_KNOCK bits: 1 no text assembly at
# all, 2 text assembled but not stored, 4 no error tests, 8 no base extraction, 16 no header, 64 producers kept alive but no assembly), to weigh
# the parts of the VALU-bound kernel.  Output is garbage by construction; only the kernel time is meaningful.  The other objects are the
# product's own (csrc/build); run the result with tools/knockout_run.sh on the GPU box.
set -e
cd "$(dirname "$0")/../dwgsim_amd/csrc"
make -s -j12 all
mkdir -p build/knock
F="--offload-arch=gfx950 -I../../tools/probe -I. -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result"
bits="${@:-2 64}"
for k in $bits; do /opt/rocm/bin/hipcc $F -DDW_PART=1 -DDW_KNOCK=$k -c dw_simulate.hip -o build/knock/s1_k$k.o & done
wait
for k in $bits; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/gzip.o build/host.o build/mutin.o build/job.o build/s0.o build/knock/s1_k$k.o build/s[2-9].o build/s10.o -lpthread -o ../libdwgsim_hip_knock$k.so
done
echo built knock libs: $bits
