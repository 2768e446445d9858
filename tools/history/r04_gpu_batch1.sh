#!/bin/bash
# tools/r04_gpu_batch1.sh -- analysis only (gpurun): round-4 baseline of the round-3 build on this round's box: the default line, two ranks
# sharing the GPU under rocprofv3 (k_place / k_scan_excl / walk kernels per launch), the Ion Torrent line, the stage times of dwgsim-hip on S4.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b1; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run n1 --steps 50
run n2_weak --gpus 2 --share-gpu --steps 20
run ion_chr20 --ion --steps 10
run ion_ecoli --ion --workload ecoli --steps 20
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats -d $o/kt_n2 -- python bench.py --gpus 2 --share-gpu --steps 6 --warmup 2 --no-legs --no-cpu-baseline > $o/kt_n2.log 2>&1
for db in $(find $o/kt_n2 -name '*.db'); do echo "== $db"; python tools/rocprof_summary.py $db | head -30; done > $o/kt_n2_summary.txt 2>&1
# whole-genome stage times of the executable (counting sink)
timeout 600 python bench.py --steps 3 --no-cpu-baseline > $o/default_with_legs.json 2> $o/default_with_legs.err
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
    for k in ("host_landed","host_landed_gz","end_to_end","end_to_end_genome"):
        if k in d: print(k, {q:d[k].get(q) for q in ("value","seconds","gz_ratio","stages")})
except Exception as e: print("ERR",e)
PY
done
cat $o/kt_n2_summary.txt | head -80
