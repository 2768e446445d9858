#!/bin/bash
# tools/r05_gpu_batch13.sh -- analysis only (gpurun): state of HEAD after the container was re-created a second time: rocprofv3 kernel stats + PMC passes of
# every profiled workload (tools/r05_final_profiles.sh), then the whole GPU suite
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b13
bash tools/r05_final_profiles.sh > gpurun_out/b13/final.log 2>&1; tail -5 gpurun_out/b13/final.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/b13/pytest.log 2>&1; tail -3 gpurun_out/b13/pytest.log
