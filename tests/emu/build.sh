#!/bin/bash
# tests/emu/build.sh -- compile the product's kernel + host sources against the CPU SIMT emulation shim
# (TEST INFRASTRUCTURE ONLY).  Output: tests/emu/libdwgsim_emu.so exporting the same C-ABI.
set -e
cd "$(dirname "$0")"
SRC=../../dwgsim_amd/csrc
# one build at a time (pytest-xdist workers, the test modules that each ask for a build), and none when nothing is newer than what is there
exec 9> .build.lock; flock 9
if [ -x dwgsim-emu ] && [ -f libdwgsim_emu.so ] && [ -z "$(find $SRC ../../include . -maxdepth 2 -type f \( -name '*.hip' -o -name '*.hpp' -o -name '*.cpp' -o -name '*.h' -o -name 'build.sh' \) -newer libdwgsim_emu.so 2>/dev/null | head -1)" ] && [ dwgsim-emu -nt libdwgsim_emu.so -o ! libdwgsim_emu.so -nt dwgsim-emu ]; then
  echo up to date: tests/emu/libdwgsim_emu.so tests/emu/dwgsim-emu; exit 0
fi
g++ -O2 -g -std=c++17 -ffp-contract=off -fPIC -shared -pthread -I. -I$SRC -x c++ $SRC/dw_walk.hip $SRC/dw_gzip.hip $SRC/dw_simulate.hip $SRC/dw_host.cpp $SRC/dw_mutin.cpp $SRC/dw_job.cpp hip_emu.cpp -o libdwgsim_emu.so
# the dwgsim-hip command line over the emulated library (exercises the multi-context / pipelined host code without a GPU)
g++ -O2 -g -std=c++17 -pthread $SRC/dwgsim_cli.cpp -o dwgsim-emu -L. -ldwgsim_emu -lz -Wl,-rpath,'$ORIGIN'
echo built tests/emu/libdwgsim_emu.so tests/emu/dwgsim-emu
