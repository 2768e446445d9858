#!/bin/bash
# tools/r05_gpu_batch12.sh -- analysis only (gpurun): after the capacity limit of the flow model went from 16 x to 2^20 bases and the fuzzer stopped drawing option
# sets the reference does not finish: the Ion Torrent and fuzz tests, 300 random flow orders, the default bench line with every leg (strong object at N = 1),
# the walk: whole-genome strong line with 32 Mi groups and with whole-genome groups, kernel times of the walk chain
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b12
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "ion or outgrows or random or fresh or flow" > gpurun_out/b12/pytest.log 2>&1; tail -3 gpurun_out/b12/pytest.log
timeout 1200 python tests/fuzz_ion_flows.py 77 300 > gpurun_out/b12/ion_fuzz.txt 2>&1; tail -2 gpurun_out/b12/ion_fuzz.txt
timeout 900 python bench.py --steps 20 2> gpurun_out/b12/default.err > gpurun_out/b12/default.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/b12/default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"]); print("strong", d.get("strong"))
print("landed", d["host_landed"]["value"], d["host_landed_gz"]["value"], "e2e", d["end_to_end"]["seconds"], "genome", d["end_to_end_genome"]["seconds"], d["end_to_end_genome"]["stages"])
PY
for gb in 33554432 2130706432; do
  echo "== strong grch38 group-bp $gb"
  python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp $gb 2> gpurun_out/b12/strong_$gb.err | tee gpurun_out/b12/strong_$gb.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['simulate_kernels'])"
  python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp $gb --no-pipeline 2>> gpurun_out/b12/strong_$gb.err | tee gpurun_out/b12/strong_nopipe_$gb.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-pipeline', d['value'], d['ms_per_step'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['simulate_kernels'])"
done
out=gpurun_out/b12/kt; rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -- python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp 2130706432 --no-pipeline > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $out -name '*.db' | head -1)" | head -40 | tee gpurun_out/b12/walk_kernels.txt
rm -rf $out
