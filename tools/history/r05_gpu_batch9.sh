#!/bin/bash
# tools/r05_gpu_batch9.sh -- analysis only (gpurun): the whole-genome product run: batch sizes, and a copy / kernel trace of one run
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b9
PROBE_VARIANTS="${PROBE_VARIANTS-default}" timeout 1500 python tools/r05_genome_probe.py 2>&1 | tee gpurun_out/b9/probe.txt
