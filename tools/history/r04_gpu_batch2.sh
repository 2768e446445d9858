#!/bin/bash
# tools/r04_gpu_batch2.sh -- analysis only (gpurun): the new count (k_place + k_place_rest) -- parity subset, then two ranks sharing the GPU under rocprofv3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b2; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "count_random or job_level or several_contexts or two_ranks" > $o/pytest.log 2>&1; tail -5 $o/pytest.log
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run n1 --steps 50
run n2_weak --gpus 2 --share-gpu --steps 20
run n2_strong_grch38 --gpus 2 --share-gpu --mode strong --workload grch38 --steps 2 --warmup 1
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats -d $o/kt_n2 -- python bench.py --gpus 2 --share-gpu --steps 6 --warmup 2 --no-legs --no-cpu-baseline > $o/kt_n2.log 2>&1
for db in $(find $o/kt_n2 -name '*.db' | head -1); do echo "== $db"; python tools/rocprof_summary.py $db | head -12; done > $o/kt_n2_summary.txt 2>&1
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
except Exception as e: print("ERR",e)
PY
done
cat $o/kt_n2_summary.txt
