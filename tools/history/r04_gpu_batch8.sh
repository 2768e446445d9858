#!/bin/bash
# tools/r04_gpu_batch8.sh -- analysis only (gpurun): the new tests with their durations; k_gzip with name-line matches (ratio, kernel time, landed rate);
# the default line with all legs; the read-length sweep around the new LDS / scratch-slot threshold
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b8; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=25 -k "scratch or staging_limit or outgrows or gzip" > $o/pytest_new.log 2>&1; tail -40 $o/pytest_new.log
timeout 600 python tools/gz_probe.py > $o/gz_probe.txt 2>&1; cat $o/gz_probe.txt
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats -d $o/kt_gz -- python tools/gz_probe.py > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $o/kt_gz -name '*.db' | head -1)" > $o/gz_kernel_stats.txt 2>&1; rm -rf $o/kt_gz; head -8 $o/gz_kernel_stats.txt
{
for L in 600 700 800; do
  for simt in 256 64 ""; do SIMT=$simt timeout 300 python tools/time_probe.py "-z 13 -1 $L -2 0 -C 30 -o 1" 2>&1 | tail -1 | sed "s/^/SIMT=${simt:-auto} /"; done
done
for simt in 256 64 ""; do SIMT=$simt timeout 300 python tools/time_probe.py "-z 13 -1 350 -2 350 -d 900 -C 30 -o 1" 2>&1 | tail -1 | sed "s/^/SIMT=${simt:-auto} /"; done
} > $o/length_sweep2.txt 2>&1
cat $o/length_sweep2.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $o/default.json 2> $o/default.err
python - "$o/default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
for k in ("host_landed","host_landed_gz","end_to_end","end_to_end_genome"):
    if k in d: print(k, {q:d[k].get(q) for q in ("value","seconds","gb_per_s","gz_ratio","gz_bytes","stages")})
PY
