#!/bin/bash
# tools/r06_gpu_batch17.sh -- (gpurun) the quality lines drawn inside the look-back wait (a.qual_scratch): on against off, the phase split, parity
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b17; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for opt in "" "qual_early=1" "" "qual_early=1"; do
  for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 250 -2 250 -C 30 -o 1" "-z 13 -1 125 -2 125 -C 30 -o 1"; do
    DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --steps 40 --no-legs --no-cpu-baseline "--flags=$fl" 2>/dev/null | line "[$opt] $fl" >> $o/lines.txt
  done
done
cat $o/lines.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k early_quality 2>&1 | tail -5 | tee $o/gpu_parity.txt
