#!/bin/bash
# tools/r06_gpu_batch36.sh -- (gpurun) analysis: lanes per k_simulate block with one look-back (-DDW_SIM_THREADS=128 / 512 builds of the whole library against the product's 256)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b36; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2; do for v in "" _var_t128 _var_t512; do
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip$v.so timeout 300 python bench.py --steps 60 --no-legs --no-cpu-baseline 2>/dev/null | line "[lib$v] chr20 2x150" >> $o/lines.txt
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip$v.so timeout 300 python bench.py --workload ecoli --steps 200 --no-legs --no-cpu-baseline 2>/dev/null | line "[lib$v] ecoli 2x150" >> $o/lines.txt
done; done
sort $o/lines.txt
