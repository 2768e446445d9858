#!/bin/bash
# tests/emu/build.sh -- compile the product's kernel + host sources against the CPU SIMT emulation shim
# (TEST INFRASTRUCTURE ONLY).  Output: tests/emu/libdwgsim_emu.so exporting the same C-ABI.
set -e
cd "$(dirname "$0")"
SRC=../../dwgsim_amd/csrc
g++ -O2 -g -std=c++17 -ffp-contract=off -fPIC -shared -pthread -I. -I$SRC -x c++ $SRC/dw_walk.hip $SRC/dw_gzip.hip $SRC/dw_simulate.hip $SRC/dw_host.cpp $SRC/dw_mutin.cpp $SRC/dw_job.cpp hip_emu.cpp -o libdwgsim_emu.so
# the dwgsim-hip command line over the emulated library (exercises the multi-context / pipelined host code without a GPU)
g++ -O2 -g -std=c++17 -pthread $SRC/dwgsim_cli.cpp -o dwgsim-emu -L. -ldwgsim_emu -lz -Wl,-rpath,'$ORIGIN'
echo built tests/emu/libdwgsim_emu.so tests/emu/dwgsim-emu
