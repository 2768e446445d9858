#!/bin/bash
# tools/r05_gpu_batch7.sh -- analysis only (gpurun): the sparse walk (dirty chunks, site scan as one kernel, whole-genome groups): parity subset, the whole-genome strong line
# with 32 Mi groups and with whole-genome groups, kernel times of the walk
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b7
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "walk or again or repeat or count_random or two_hundred or justify or mutation or vcf or bed" > gpurun_out/b7/pytest.log 2>&1; tail -3 gpurun_out/b7/pytest.log
python bench.py --no-legs --no-cpu-baseline --steps 20 2> gpurun_out/b7/default.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['breakdown_ms']['walk_gpu'])"
for gb in 33554432 2130706432; do
  echo "== strong grch38 group-bp $gb"
  python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp $gb 2> gpurun_out/b7/strong_$gb.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['breakdown_ms'])"
done
out=gpurun_out/b7/kt; rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -- python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp 2130706432 --no-pipeline > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $out -name '*.db' | head -1)" | head -30
rm -rf $out
