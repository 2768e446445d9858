#!/bin/bash
# tools/r04_gpu_batch6.sh -- analysis only (gpurun): the Ion Torrent scratch as slots handed from block to block inside an XCD (parity, lines, counters),
# and the rocprofv3 passes of the variants last profiled in round 1: -o 0 (both output families), SOLiD 2 x 50, a one-wave long-read variant.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b6; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
timeout 1500 python -m pytest tests -x -q -m gpu -k "ion or flow or scratch or kernel_parity or fuzz" > $o/pytest.log 2>&1; tail -5 $o/pytest.log
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run ion_chr20 --ion --steps 10
run ion_ecoli --ion --workload ecoli --steps 20
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
except Exception as e: print("ERR",e)
PY
done
prof() { tag=$1; wl=$2; extra=$3; timeout 900 bash tools/profile_round.sh r04_b6/$tag $wl "$extra" > $o/profile_$tag.log 2>&1; echo "=== $tag"; tail -45 $o/profile_$tag.log | cut -c1-200; }
prof ion_ecoli ecoli "--ion"
prof ion_chr20 chr20 "--ion"
prof o0 chr20 "--flags='-z 13 -1 150 -2 150 -C 30 -o 0'"
prof solid50 chr20 "--flags='-z 13 -c 1 -1 50 -2 50 -C 30 -o 0'"
prof long2000 chr20 "--flags='-z 13 -1 2000 -2 0 -C 30 -o 1'"
