#!/bin/bash
# tools/r06_gpu_batch3.sh -- analysis only (gpurun): how many parked lanes make an event round of the flow model worth running (DW_FLOW_EVENT_BATCH = 16 / 24 / 32 / 40 / 48)
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/r06b3; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for rep in 1 2; do for v in product eb1 eb4 eb8; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so
  DWGSIM_HIP_LIB=$lib python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "chr20,ion400,$v"
  DWGSIM_HIP_LIB=$lib python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "ecoli,ion400,$v"
done; done | tee $o/event_batch.txt
DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_eb4.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "ion or Ion or torrent or flow or genome_like" > $o/pytest_eb4.log 2>&1; tail -2 $o/pytest_eb4.log
