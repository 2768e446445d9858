#!/bin/bash
# tools/r04_gpu_batch10.sh -- analysis only (gpurun): k_gzip with the sort + two-queue code construction and ballot ranks: ratio, kernel time, gzip tests,
# the default line with its legs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b10; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
timeout 600 python tools/gz_probe.py > $o/gz_probe.txt 2>&1; cat $o/gz_probe.txt
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats -d $o/kt_gz -- python tools/gz_probe.py > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $o/kt_gz -name '*.db' | head -1)" > $o/gz_kernel_stats.txt 2>&1; rm -rf $o/kt_gz; head -5 $o/gz_kernel_stats.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gzip or command_line" > $o/pytest_gz.log 2>&1; tail -3 $o/pytest_gz.log
timeout 900 python bench.py --steps 20 --warmup 5 > $o/default.json 2> $o/default.err
python - "$o/default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
for k in ("host_landed","host_landed_gz","end_to_end","end_to_end_genome"):
    if k in d: print(k, {q:d[k].get(q) for q in ("value","seconds","gb_per_s","gz_ratio","gz_bytes","stages")})
PY
