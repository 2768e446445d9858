#!/bin/bash
# tools/r04_gz_variants.sh -- analysis only (gpurun): k_gzip as built against source variants (sed on a copy): pass 3a reading the code table through the
# plain LDS pointer instead of the laundered one; without the scheduling fences
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_gzv; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
cd dwgsim_amd/csrc; mkdir -p build/knock
sed 's/const uint32_t \*codes = launder_lds(s_code);/const uint32_t *codes = s_code;/' dw_gzip.hip > build/knock/dw_gzip_v1.hip
sed 's/const uint32_t \*codes = launder_lds(s_code);/const uint32_t *codes = s_code;/; s/sched_fence();/;/' dw_gzip.hip > build/knock/dw_gzip_v2.hip
for v in v1 v2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -I. -O3 -std=c++17 -ffp-contract=off -fPIC -c build/knock/dw_gzip_$v.hip -o build/knock/gzip_$v.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '; echo " <- $v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/knock/gzip_$v.o build/host.o build/mutin.o build/job.o build/s[0-9].o build/s10.o -lpthread -o ../libdwgsim_hip_knockgz$v.so
done
cd ../..
for v in full v1 v2; do
  lib=dwgsim_amd/libdwgsim_hip.so; [ $v != full ] && lib=dwgsim_amd/libdwgsim_hip_knockgz$v.so
  echo "== $v: $(DWGSIM_HIP_LIB=$lib timeout 300 python tools/gz_probe.py 2>&1 | grep 'gzip True\|equal' | tail -3 | tr '\n' ' ')"
done | tee $o/gz_variants.txt
