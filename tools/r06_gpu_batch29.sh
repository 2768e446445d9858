#!/bin/bash
# tools/r06_gpu_batch29.sh -- (gpurun) analysis: the ticket taken behind the staging of a block's tables (-DDW_TICKET_LATE) against in front of it, alternating
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b29; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2 3; do for lib in dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_tl.so; do
  DWGSIM_HIP_LIB=$lib python bench.py --steps 60 --no-legs --no-cpu-baseline 2>/dev/null | line "[$(basename $lib)] chr20 2x150" >> $o/lines.txt
  DWGSIM_HIP_LIB=$lib python bench.py --workload ecoli --steps 200 --no-legs --no-cpu-baseline 2>/dev/null | line "[$(basename $lib)] ecoli 2x150" >> $o/lines.txt
  DWGSIM_HIP_LIB=$lib python bench.py --steps 40 --no-legs --no-cpu-baseline "--flags=-z 13 -1 100 -2 100 -C 30 -o 1" 2>/dev/null | line "[$(basename $lib)] chr20 2x100" >> $o/lines.txt
done; done
sort $o/lines.txt
