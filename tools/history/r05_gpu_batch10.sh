#!/bin/bash
# tools/r05_gpu_batch10.sh -- analysis only (gpurun): the job level with its batches in flight across group ends and the mutation text on a thread of its own:
# job-level / command-line / whole-genome parity tests, then the whole-genome product run (three times) and its copy / kernel trace
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b10
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grch38.py -x -q -m gpu -k "job or command or cli or grch38 or genome or two_hundred or mutation" > gpurun_out/b10/pytest.log 2>&1; tail -3 gpurun_out/b10/pytest.log
PROBE_VARIANTS="${PROBE_VARIANTS-default;default}" timeout 900 python tools/r05_genome_probe.py 2>&1 | tee gpurun_out/b10/probe.txt
