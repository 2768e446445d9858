#!/bin/bash
# tools/r06_gpu_batch24.sh -- (gpurun) analysis: the walk stream's priority against the batches' (DWGSIM_HIP_WALK_PRIO), re-measured on the final kernels
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b24; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['roofline']['frac'])"; }
for rep in 1 2; do for pr in low high above mid; do
  DWGSIM_HIP_WALK_PRIO=$pr python bench.py --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "[chr20 walk prio $pr]" >> $o/lines.txt
  DWGSIM_HIP_WALK_PRIO=$pr python bench.py --workload ecoli --steps 200 --no-legs --no-cpu-baseline 2>/dev/null | line "[ecoli walk prio $pr]" >> $o/lines.txt
done; done
DWGSIM_HIP_WALK_PRIO=high python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "[genome strong N=1 walk prio high]" >> $o/lines.txt
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "[genome strong N=1 walk prio low]" >> $o/lines.txt
cat $o/lines.txt
