#!/bin/bash
# tools/r04_gpu_batch9.sh -- analysis only (gpurun): k_gzip with the text staged in LDS for the parse, both Huffman codes in one merge loop and the
# mask-driven block header; the new tests with durations; default line with legs; FASTA ingest on its own; the genome's walk on its own
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b9; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
timeout 600 python tools/gz_probe.py > $o/gz_probe.txt 2>&1; cat $o/gz_probe.txt
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats -d $o/kt_gz -- python tools/gz_probe.py > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $o/kt_gz -name '*.db' | head -1)" > $o/gz_kernel_stats.txt 2>&1; rm -rf $o/kt_gz; head -6 $o/gz_kernel_stats.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=12 -k "scratch or staging_limit or outgrows or gzip" > $o/pytest_new.log 2>&1; tail -22 $o/pytest_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > $o/default.json 2> $o/default.err
python - "$o/default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
for k in ("host_landed","host_landed_gz","end_to_end","end_to_end_genome"):
    if k in d: print(k, {q:d[k].get(q) for q in ("value","seconds","gb_per_s","gz_ratio","gz_bytes","stages")})
PY
timeout 900 python tools/r04_ingest_probe.py > $o/ingest_probe.txt 2>&1; cat $o/ingest_probe.txt
rocprofv3 --kernel-trace --stats -d $o/kt_walk -- python bench.py --workload grch38 --mode strong --no-pipeline --steps 2 --warmup 1 --no-legs --no-cpu-baseline > $o/grch38_nopipe.json 2> $o/kt_walk.log
python tools/rocprof_summary.py "$(find $o/kt_walk -name '*.db' | head -1)" > $o/walk_kernel_stats.txt 2>&1; rm -rf $o/kt_walk; head -30 $o/walk_kernel_stats.txt
