#!/bin/bash
# tools/r04_gz_variants.sh -- analysis only (gpurun): k_gzip as built (launch bound: 3 waves per SIMD) against the same source without the bound; phase cuts
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_gzv; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
cd dwgsim_amd/csrc; mkdir -p build/knock
sed 's/__launch_bounds__(GZ_THREADS, 3) k_gzip/__launch_bounds__(GZ_THREADS) k_gzip/' dw_gzip.hip > build/knock/dw_gzip_nb.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -I. -O3 -std=c++17 -ffp-contract=off -fPIC -c build/knock/dw_gzip_nb.hip -o build/knock/gzip_nb.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/knock/gzip_nb.o build/host.o build/mutin.o build/job.o build/s[0-9].o build/s10.o -lpthread -o ../libdwgsim_hip_knockgznb.so
cd ../..
for v in full nb; do
  lib=dwgsim_amd/libdwgsim_hip.so; [ $v = nb ] && lib=dwgsim_amd/libdwgsim_hip_knockgznb.so
  echo "== $v: $(DWGSIM_HIP_LIB=$lib timeout 300 python tools/gz_probe.py 2>&1 | grep 'gzip True\|equal' | tail -3 | tr '\n' ' ')"
done | tee $o/gz_variants.txt
bash tools/r04_gz_knock.sh 2>&1 | tail -7
