#!/bin/bash
# tools/r06_gpu_batch19.sh -- (gpurun) with ONE look-back in the single kernel: where does the two-kernel form still win?  read lengths and launch sizes; the phase split
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b19; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for fl in "-z 13 -1 36 -2 36 -C 30 -o 1" "-z 13 -1 50 -2 50 -C 30 -o 1" "-z 13 -1 75 -2 75 -C 30 -o 1" "-z 13 -1 100 -2 100 -C 30 -o 1" "-z 13 -1 100 -2 0 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 1"; do
  for opt in split=0 split=1; do
    DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --steps 30 --no-legs --no-cpu-baseline "--flags=$fl" 2>/dev/null | line "[chr20 $opt] $fl" >> $o/lines.txt
  done
done
for wl in ecoli; do for opt in split=0 split=1; do
  DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --workload $wl --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "[$wl $opt] 2x150" >> $o/lines.txt
done; done
for mp in 131072 262144 524288 1048576 2097152; do for opt in split=0 split=1; do
  DWGSIM_BENCH_MAX_LAUNCH_PAIRS=$mp DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --steps 20 --no-legs --no-cpu-baseline 2>/dev/null | line "[chr20 launches of $mp pairs $opt] 2x150" >> $o/lines.txt
done; done
cat $o/lines.txt
