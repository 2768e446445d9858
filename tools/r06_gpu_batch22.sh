#!/bin/bash
# tools/r06_gpu_batch22.sh -- (gpurun) analysis: the poll interval and the hop width of the ONE look-back; the E. coli-sized launch, one kernel against two, several times; the phase split with one look-back
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b22; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2; do for v in "" _var_sl0 _var_sl1 _var_sl4 _var_w2; do
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip$v.so python bench.py --steps 40 --no-legs --no-cpu-baseline 2>/dev/null | line "[lib$v] chr20 2x150" >> $o/lines.txt
done; done
for rep in 1 2 3; do for opt in split=0 split=1; do for wl in ecoli ecoli_like; do
  DWGSIM_BENCH_DEBUG_OPTIONS=$opt python bench.py --workload $wl --steps 200 --no-legs --no-cpu-baseline 2>/dev/null | line "[$wl $opt]" >> $o/lines.txt
done; done; done
cat $o/lines.txt
