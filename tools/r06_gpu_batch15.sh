#!/bin/bash
# tools/r06_gpu_batch15.sh -- (gpurun) analysis: the two-kernel form (no look-backs) against the single kernel, re-measured on the round's final kernels; the phase split of both
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b15; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['kernel'], d['roofline']['frac'])"; }
for opt in "" "split=1" "split=1,writer=0" "writer=0"; do
  for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 250 -2 250 -C 30 -o 1" "-z 13 -1 100 -2 100 -C 30 -o 1"; do
    DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --steps 40 --no-legs --no-cpu-baseline "--flags=$fl" 2>/dev/null | line "[$opt] $fl" >> $o/lines.txt
  done
done
cat $o/lines.txt
DWGSIM_BENCH_DEBUG_OPTIONS=split=1 DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_phases.so python bench.py --phases --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>&1 | grep phases | tail -4 | tee $o/phases_split.txt
