#!/bin/bash
# tools/r05_gpu_batch8.sh -- analysis only (gpurun): state of HEAD after the container was re-created: the whole GPU suite, the default bench
# line with every leg, Ion Torrent lines (both sizes), the whole-genome strong line (N = 1)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b8
sha256sum dwgsim_amd/libdwgsim_hip.so | tee gpurun_out/b8/lib.sha
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/b8/pytest.log 2>&1; tail -5 gpurun_out/b8/pytest.log
timeout 900 python bench.py 2> gpurun_out/b8/default.err | tee gpurun_out/b8/default.json | cut -c1-3000
for wl in chr20 ecoli; do
  timeout 600 python bench.py --workload $wl --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 2> gpurun_out/b8/ion_$wl.err | tee gpurun_out/b8/ion_$wl.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ion', d['value'], d['unit'], d['ms_per_step'], d['roofline'])"
done
timeout 600 python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/b8/strong.err | tee gpurun_out/b8/strong.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strong', d['value'], d['ms_per_step'], d['breakdown_ms'])"
