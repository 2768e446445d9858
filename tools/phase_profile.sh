#!/bin/bash
# tools/phase_profile.sh -- analysis only: build a phase-timing variant of the library
# (dwgsim_amd/libdwgsim_hip_phases.so, -DDW_PHASE_TIMING) next to the product library.
# Use:  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_phases.so python bench.py --phases --steps 3 --warmup 1 --no-legs --no-cpu-baseline
set -e
cd "$(dirname "$0")/../dwgsim_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -I../../tools/probe -I. -O3 -std=c++17 -ffp-contract=off -fPIC -DDW_PHASE_TIMING $PHASE_FLAGS -Wno-unused-value -shared dw_walk.hip dw_gzip.hip dw_simulate.hip dw_host.cpp dw_mutin.cpp dw_job.cpp -lpthread -o ../libdwgsim_hip_phases.so
echo built dwgsim_amd/libdwgsim_hip_phases.so
