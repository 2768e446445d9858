#!/bin/bash
# tools/r04_slots_sweep.sh -- analysis only (gpurun): scratch slots per XCD for the one-wave long-read blocks (0 = the library's count: one per block an XCD can hold)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_slots; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
for L in 2000 10000; do for fs in 0 512 256 128 64; do
  FLOW_SLOTS=$fs timeout 300 python tools/time_probe.py "-z 13 -1 $L -2 0 -C 30 -o 1" 2>&1 | tail -1 | sed "s/^/slots=$fs /"
done; done | tee $o/slots.txt
