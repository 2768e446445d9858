#!/bin/bash
# tools/r06_gpu_batch33.sh -- (gpurun) an episode's cells fetched sixteen at once + the insertion table searched by interpolation, against the final library of batch 28 (same box, alternating); the -m gpu suite
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b33; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2 3; do for lib in dwgsim_amd/libdwgsim_hip_var_old.so dwgsim_amd/libdwgsim_hip.so; do
  DWGSIM_HIP_LIB=$lib python bench.py --steps 60 --no-legs --no-cpu-baseline 2>/dev/null | line "[$(basename $lib)] chr20 2x150" >> $o/lines.txt
  DWGSIM_HIP_LIB=$lib python bench.py --steps 40 --no-legs --no-cpu-baseline "--flags=-z 13 -1 150 -2 150 -C 30 -o 1 -r 0.01 -R 0.3" 2>/dev/null | line "[$(basename $lib)] chr20 2x150 -r 0.01 -R 0.3" >> $o/lines.txt
  DWGSIM_HIP_LIB=$lib python bench.py --steps 30 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "[$(basename $lib)] chr20 ion400" >> $o/lines.txt
done; done
sort $o/lines.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $o/gputest.txt 2>&1; tail -3 $o/gputest.txt
