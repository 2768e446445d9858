#!/bin/bash
# tools/r04_phases.sh -- analysis only (gpurun): phase split of k_simulate (the -DDW_PHASE_TIMING build of tools/phase_profile.sh) at 2 x 150 and 2 x 50 bp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_phases; mkdir -p $o
for spec in "150:-z 13 -1 150 -2 150 -C 30 -o 1" "50:-z 13 -1 50 -2 50 -C 10 -o 1" "150q0:-z 13 -1 150 -2 150 -C 30 -o 1 -Q 0"; do
  tag=${spec%%:*}; fl=${spec#*:}
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_phases.so timeout 300 python bench.py --phases --steps 3 --warmup 1 --no-legs --no-cpu-baseline --flags "$fl" > $o/ph_$tag.json 2> $o/ph_$tag.err
  echo "== $tag"; grep phases $o/ph_$tag.err | tail -1
  python -c "
import json;d=json.loads(open('$o/ph_$tag.json').read().strip().splitlines()[-1]);print(d['value'], d['breakdown_ms']['simulate_kernels'])"
done
