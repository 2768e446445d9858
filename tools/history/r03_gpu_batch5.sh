#!/bin/bash
# tools/r03_gpu_batch5.sh -- analysis only (gpurun): full GPU suite, the default bench line with all legs, rocprof passes of the new kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r03_b5; mkdir -p $o
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $o/build_smoke.log 2>&1; tail -1 $o/build_smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -4 $o/pytest_gpu.log
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; tail -c 300 $o/bench_default.err | grep -v amdgpu
bash tools/r03_final_profiles.sh > $o/profiles.log 2>&1; tail -30 $o/profiles.log
