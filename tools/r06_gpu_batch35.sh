#!/bin/bash
# tools/r06_gpu_batch35.sh -- (gpurun) analysis: where a wave drops its raised issue priority, re-measured with one look-back (DW_PRIO_DROP 1 = behind the name line (product), 0 = behind the look-back, 2 = never: raised throughout)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b35; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2 3; do for v in "" _var_pd0 _var_pd2; do
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip$v.so python bench.py --steps 60 --no-legs --no-cpu-baseline 2>/dev/null | line "[lib$v] chr20 2x150" >> $o/lines.txt
done; done
sort $o/lines.txt
