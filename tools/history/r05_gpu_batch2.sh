#!/bin/bash
# tools/r05_gpu_batch2.sh -- analysis only (gpurun): counters of the new Ion Torrent kernel, read buffers in scratch slots (0) / LDS (1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in 0 1; do
  DWGSIM_HIP_DEBUG="ion_lds=$m" timeout 900 bash tools/profile_round.sh ion_m$m chr20 "--ion --no-genome-leg" > gpurun_out/ion_m$m.log 2>&1
  echo "== ion_lds=$m"; head -3 gpurun_out/ion_m$m/kernel_stats.txt; cat gpurun_out/ion_m$m/pmc.txt
done
