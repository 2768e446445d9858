#!/bin/bash
# tools/r04_gpu_batch13.sh -- analysis only (gpurun): k_gzip timing (gz_probe), then the WHOLE GPU suite on the build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b13; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
timeout 600 python tools/gz_probe.py 2>&1 | grep "gzip True\|equal" | tail -3
timeout 1800 python -m pytest tests -x -q -m gpu --durations=15 > $o/pytest.log 2>&1; tail -25 $o/pytest.log
