#!/bin/bash
# tools/r06_gpu_batch20.sh -- (gpurun) ONE look-back also for SOLiD (two words, two chains walked side by side), the two-kernel cut at 50 bases: lines against the three-look-back build,
# the -m gpu suite, the final profiles and the bench line of this library
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b20; mkdir -p $o; : > $o/lines.txt
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
for lib in dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_lb3.so dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_lb3.so; do
  for fl in "-z 13 -c 1 -1 50 -2 50 -C 30 -o 1" "-z 13 -c 1 -1 50 -2 50 -C 30 -o 0" "-z 13 -c 1 -1 75 -2 0 -C 30 -o 0" "-z 13 -1 100 -2 100 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 1"; do
    DWGSIM_HIP_LIB=$lib python bench.py --steps 30 --no-legs --no-cpu-baseline "--flags=$fl" 2>/dev/null | line "[$(basename $lib)] $fl" >> $o/lines.txt
  done
done
cat $o/lines.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $o/gputest.txt 2>&1; tail -3 $o/gputest.txt
if grep -q " passed" $o/gputest.txt && ! grep -q "failed" $o/gputest.txt; then
  bash tools/r06_final_profiles.sh > $o/final.log 2>&1
  cp gpurun_out/final/r06_counters.json profiles/r06_counters.json
  python bench.py > gpurun_out/final/bench_line_n1.json 2> gpurun_out/final/bench_line_n1.err; python -c "import json; d=json.loads(open('gpurun_out/final/bench_line_n1.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['strong']['value'])"
fi
