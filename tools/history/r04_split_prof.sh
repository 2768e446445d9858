#!/bin/bash
# analysis only (gpurun): per-kernel times of the two-kernel form
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_split_prof; rm -rf $o; mkdir -p $o
for sp in 1 0; do
SPLIT=$sp rocprofv3 --kernel-trace --stats -d $o/kt$sp -- python tools/time_probe.py "-z 13 -1 150 -2 150 -C 30 -o 1" > $o/log$sp.txt 2>&1
python tools/rocprof_summary.py $(find $o/kt$sp -name '*.db' | head -1) | head -6
done
