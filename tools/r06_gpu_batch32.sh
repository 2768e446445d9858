#!/bin/bash
# tools/r06_gpu_batch32.sh -- (gpurun) analysis, flags only, the final kernels (one look-back): what mutations (episodes of the per-cell logic), random reads and rejected placements cost the launch
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b32; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
B="-z 13 -1 150 -2 150 -C 30 -o 1"
for rep in 1 2; do for x in "" "-r 0" "-y 0" "-n 1000" "-e 0 -E 0" "-r 0 -y 0 -n 1000" "-r 0 -y 0 -n 1000 -e 0 -E 0"; do
  python bench.py --steps 40 --no-legs --no-cpu-baseline "--flags=$B $x" 2>/dev/null | line "[$x]" >> $o/lines.txt
done; done
sort $o/lines.txt
