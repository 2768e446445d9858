#!/bin/bash
# tools/r03_gpu_probe.sh -- analysis only (gpurun): kernel time of the chr20-sized launch, both writers, a few flag sets; a quick parity sample
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for f in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 1 -Q 0" "-z 13 -1 50 -2 50 -C 10 -o 1"; do for w in 1 0; do WRITER=$w python tools/time_probe.py "$f" 2>/dev/null; done; done
WL=ecoli python tools/time_probe.py "-z 13 -1 150 -2 150 -C 30 -o 1" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_exact_vs_oracle and not resident" 2>&1 | tail -3
