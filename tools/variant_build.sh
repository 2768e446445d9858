#!/bin/bash
# tools/variant_build.sh <name> "<extra hipcc flags>" [part] -- analysis only: dwgsim_amd/libdwgsim_hip_var_<name>.so = the product's objects with the
# k_simulate<2,*,0> part (DW_PART=1) recompiled with extra flags (e.g. -DDW_SIM_WAVES=6); time it with DWGSIM_HIP_LIB=... tools/time_probe.py
set -e
cd "$(dirname "$0")/../dwgsim_amd/csrc"
make -s -j12 all
mkdir -p build/var
F="--offload-arch=gfx950 -I../../tools/probe -I. -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result"
P=${3:-1}
/opt/rocm/bin/hipcc $F -DDW_PART=$P $2 -c dw_simulate.hip -o build/var/s${P}_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/gzip.o build/host.o build/mutin.o build/job.o build/s0.o build/var/s${P}_$1.o $(ls build/s[1-9].o build/s1[0-5].o | grep -v "build/s$P.o") -lpthread -o ../libdwgsim_hip_var_$1.so
echo built libdwgsim_hip_var_$1.so
