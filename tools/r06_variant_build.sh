#!/bin/bash
# tools/r06_variant_build.sh <name> <extra hipcc flags...> -- analysis only: a variant of the library (all of dw_simulate.hip's parts recompiled with the flags,
# the other objects are the product's own) as dwgsim_amd/libdwgsim_hip_var_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../dwgsim_amd/csrc"
mkdir -p build/var_$name
F="--offload-arch=gfx950 $PRE_FLAGS -I. -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result $@"
for k in $(seq 0 15); do /opt/rocm/bin/hipcc $F -DDW_PART=$k -c dw_simulate.hip -o build/var_$name/s$k.o & if (( (k + 1) % 8 == 0 )); then wait; fi; done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/gzip.o build/host.o build/mutin.o build/job.o build/var_$name/s*.o -lpthread -o ../libdwgsim_hip_var_$name.so
echo built ../libdwgsim_hip_var_$name.so
