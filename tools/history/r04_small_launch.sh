#!/bin/bash
# tools/r04_small_launch.sh -- analysis only (gpurun): small launches of the 2 x 150 bp kernel, one kernel with look-backs against the two-kernel form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_small; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
for N in 65536 131072 262144 524288 1048576 2097152; do for sp in 0 1; do
  SPLIT=$sp timeout 300 python tools/time_probe.py "-z 13 -1 150 -2 150 -N $N -o 1" 2>&1 | tail -1 | sed "s/^/split=$sp /"
done; done | tee $o/small.txt
