#!/bin/bash
# tools/r06_gpu_batch23.sh -- (gpurun) one stream operation in front of a launch (k_launch_init) instead of three or four: the E. coli-sized line, small launches, the default line; the -m gpu suite
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b23; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2 3; do for opt in split=0 split=1; do for wl in ecoli ecoli_like; do
  DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --workload $wl --steps 200 --no-legs --no-cpu-baseline 2>/dev/null | line "[$wl $opt]" >> $o/lines.txt
done; done; done
for mp in 131072 262144 524288; do for opt in split=0 split=1; do
  DWGSIM_BENCH_MAX_LAUNCH_PAIRS=$mp DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --steps 20 --no-legs --no-cpu-baseline 2>/dev/null | line "[chr20 launches of $mp pairs $opt] 2x150" >> $o/lines.txt
done; done
for rep in 1 2; do python bench.py --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "[chr20] 2x150" >> $o/lines.txt; done
cat $o/lines.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $o/gputest.txt 2>&1; tail -3 $o/gputest.txt
