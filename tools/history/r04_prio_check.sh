#!/bin/bash
# tools/r04_prio_check.sh -- analysis only (gpurun): lone-launch times of every k_simulate family after the phase priorities (tools/time_probe.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 125 -2 125 -C 30 -o 1" "-z 13 -c 1 -1 50 -2 50 -C 30 -o 0" "-z 13 -1 2000 -2 0 -C 30 -o 1" "-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.01 -o 1"; do SPLIT=0 timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1; done
WL=ecoli timeout 300 python tools/time_probe.py "-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.01 -o 1" 2>&1 | tail -1
