#!/bin/bash
# tools/r06_gpu_batch7.sh -- analysis only (gpurun): predecessors per hop of the decoupled look-back (DW_LB_W x 64: 64 / 128 / 256 (product) / 512), against a look-back that never waits
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b7; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2; do for v in product lb1 lb2 lb8 knock2048; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so
  DWGSIM_HIP_LIB=$lib python bench.py --steps 50 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150,$v"
  DWGSIM_HIP_LIB=$lib python bench.py --steps 30 --no-legs --no-cpu-baseline "--flags=-z 13 -1 150 -2 150 -C 30 -o 0" 2>/dev/null | line "chr20,2x150-o0,$v"
  DWGSIM_HIP_LIB=$lib python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,2x150,$v"
done; done | tee $o/lookback_width.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "option_surface or forms" > $o/pytest.log 2>&1; tail -2 $o/pytest.log
