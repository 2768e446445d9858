#!/bin/bash
# tools/r06_gpu_batch11.sh -- (gpurun) analysis: what the generator's rounds cost (a 7-round build, timing only -- its streams are not the product's), then the final profiles of the current library
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b11; mkdir -p $o; : > $o/lines.txt
for lib in dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_ph7.so; do
  for i in 1 2; do echo "$lib" >> $o/lines.txt; DWGSIM_HIP_LIB=$lib python bench.py --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])" >> $o/lines.txt; done
  echo "$lib ion" >> $o/lines.txt; DWGSIM_HIP_LIB=$lib python bench.py --steps 60 --no-legs --no-cpu-baseline --ion 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])" >> $o/lines.txt
done
cat $o/lines.txt
bash tools/r06_final_profiles.sh > $o/final.log 2>&1
cp gpurun_out/final/r06_counters.json profiles/r06_counters.json
python bench.py > gpurun_out/final/bench_line_n1.json 2> gpurun_out/final/bench_line_n1.err; tail -c 600 gpurun_out/final/bench_line_n1.json
