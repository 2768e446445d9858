#!/bin/bash
# tools/r06_gpu_batch18.sh -- (gpurun) ONE look-back (random reads and bytes in one word, the digits arithmetic) against the three of rounds 2-5: lines, then parity
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b18; mkdir -p $o; : > $o/lines.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for lib in dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_lb3.so dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_lb3.so; do
  for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 250 -2 250 -C 30 -o 1" "-z 13 -1 125 -2 125 -C 30 -o 1" "-z 13 -1 150 -2 0 -C 30 -o 1"; do
    DWGSIM_HIP_LIB=$lib python bench.py --steps 40 --no-legs --no-cpu-baseline "--flags=$fl" 2>/dev/null | line "[$(basename $lib)] $fl" >> $o/lines.txt
  done
done
cat $o/lines.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $o/gputest.txt
