#!/bin/bash
# tools/r04_gpu_batch5.sh -- analysis only (gpurun): state of the round-4 build after the re-entry: the whole GPU suite, the default line with all legs,
# two and eight ranks sharing the GPU (readiness of the N-rank path, not a speed-up), S4 strong, and the rocprofv3 passes of the default workload.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b5; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -5 $o/pytest.log
run() { name=$1; shift; timeout 900 python bench.py "$@" > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run default --steps 20 --warmup 5
run n1 --steps 50 --no-legs --no-cpu-baseline
run n1_ecoli --workload ecoli --steps 100 --no-legs --no-cpu-baseline
run n2_weak_share --gpus 2 --share-gpu --steps 20 --no-legs --no-cpu-baseline
run n8_weak_share --gpus 8 --share-gpu --steps 5 --warmup 2 --no-legs --no-cpu-baseline
run n1_strong_grch38 --mode strong --workload grch38 --steps 2 --warmup 1 --no-legs --no-cpu-baseline
run n8_strong_grch38_share --gpus 8 --share-gpu --mode strong --workload grch38 --steps 1 --warmup 1 --no-legs --no-cpu-baseline
run ion_chr20 --ion --steps 10 --no-legs --no-cpu-baseline
run ion_ecoli --ion --workload ecoli --steps 20 --no-legs --no-cpu-baseline
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
bash tools/profile_round.sh r04_b5/chr20 chr20 > $o/profile_chr20.log 2>&1
tail -40 $o/profile_chr20.log
