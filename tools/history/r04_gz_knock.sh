#!/bin/bash
# tools/r04_gz_knock.sh -- analysis only (gpurun): k_gzip cut off after each of its phases (-DDW_KNOCK bits 2^20 .. 2^24 through tools/probe/dw_probe.hpp;
# output is garbage by construction, only the time means something): where the kernel's time goes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_gzk; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
cd dwgsim_amd/csrc; mkdir -p build/knock
F="--offload-arch=gfx950 -I../../tools/probe -I. -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result"
for b in 22 23 27 24 25 26; do k=$((1 << b)); [ $b = 26 ] && k=$(( (1 << 25) | (1 << 26) )); /opt/rocm/bin/hipcc $F -DDW_KNOCK=$k -c dw_gzip.hip -o build/knock/gzip_k$b.o & done; wait
for b in 22 23 27 24 25 26; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/knock/gzip_k$b.o build/host.o build/mutin.o build/job.o build/s[0-9].o build/s10.o -lpthread -o ../libdwgsim_hip_knockgz$b.so; done
cd ../..
for b in 22 23 27 24 25 26 full; do
  lib=dwgsim_amd/libdwgsim_hip_knockgz$b.so; [ $b = full ] && lib=dwgsim_amd/libdwgsim_hip.so
  echo "== knock $b (22 / 23 / 27 / 24: cut off after histograms / codes / header tokens / look-back; 25: no parse; 26: no parse, text not staged in LDS): $(DWGSIM_HIP_LIB=$lib timeout 300 python tools/gz_probe.py 2>&1 | grep 'gzip True' | tail -1)"
done | tee $o/gz_knock.txt
