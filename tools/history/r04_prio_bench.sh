#!/bin/bash
# tools/r04_prio_bench.sh -- analysis only (gpurun): bench.py lines with and without the phase priorities of k_simulate on ONE box (the variant is a sed-ed copy)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_priob; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
cd dwgsim_amd/csrc; mkdir -p build/knock
sed 's/if (SPLIT == 0) wave_priority(1);/;/; s/if (SPLIT == 0) wave_priority(0);/;/' dw_simulate.hip > build/knock/dw_simulate_noprio.hip
for P in 1 2 3 4 5 6 7 8; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -I. -O3 -std=c++17 -ffp-contract=off -fPIC -DDW_PART=$P -c build/knock/dw_simulate_noprio.hip -o build/knock/np$P.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/gzip.o build/host.o build/mutin.o build/job.o build/s0.o build/knock/np[1-8].o build/s9.o build/s10.o -lpthread -o ../libdwgsim_hip_knock_noprio.so
cd ../..
for rep in 1 2; do for v in product noprio; do
  lib=dwgsim_amd/libdwgsim_hip.so; [ $v = noprio ] && lib=dwgsim_amd/libdwgsim_hip_knock_noprio.so
  for args in "--steps 50" "--steps 50 --no-pipeline" "--workload ecoli --steps 100" "--ion --steps 10" "--workload assembly5k --steps 10"; do
    DWGSIM_HIP_LIB=$lib timeout 600 python bench.py $args --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$args', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'])"
  done
done; done | tee $o/prio_bench.txt
