#!/bin/bash
# tools/r05_gpu_batch23.sh -- analysis only (gpurun): the SOLiD kernels with an occupancy hint (-DDW_SOLID_WAVES=4: 128 registers + spills against 152 and three waves),
# the 8-rank readiness line (ranks sharing the GPU hold the genome twice), the whole-genome product run on a quiet box
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b23; mkdir -p $o
( bash tools/variant_build.sh solid4 "-DDW_SOLID_WAVES=4" 5   # (the product since); ) > $o/variant.log 2>&1; grep built $o/variant.log
for fl in "-z 13 -c 1 -1 50 -2 50 -C 30 -o 0" "-z 13 -c 1 -1 50 -2 50 -C 30 -o 1" "-z 13 -c 1 -1 75 -2 35 -C 30 -o 2"; do
  for v in product solid4 product solid4; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; DWGSIM_HIP_LIB=$lib timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$v /"; done
done | tee $o/solid_probe.txt
PROBE_TRACE=0 PROBE_VARIANTS="default;default" timeout 900 python tools/r05_genome_probe.py 2>&1 | tee $o/genome_probe.txt | cut -c1-220
{ echo "## python bench.py --gpus 8 --share-gpu --no-legs --no-cpu-baseline --steps 5 --warmup 1   (eight ranks on the ONE GPU: the 8-rank path runs, weak line + strong object)"; timeout 1500 python bench.py --gpus 8 --share-gpu --no-legs --no-cpu-baseline --steps 5 --warmup 1 2>$o/n8.err; } > $o/n8_line.txt; cut -c1-300 $o/n8_line.txt; tail -3 $o/n8.err
