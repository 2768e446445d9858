#!/bin/bash
# tools/r05_gpu_batch21.sh -- analysis only (gpurun): k_dirty_chunks in both mappings (chosen by group size), the quality line back in registers: parity (whole
# suite), the walk alone on the genome in both groupings, bench lines
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b21; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -3 $o/pytest.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2 3; do $B 2>/dev/null | line "default"; done 2>&1 | tee $o/bench_variants.txt
for a in "--workload ecoli --steps 50" "--workload assembly5k --steps 30" "--workload grch38 --mode strong --steps 3 --warmup 1" "--workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432" "--workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 33554432"; do
  eval "python bench.py $a --no-legs --no-cpu-baseline" 2>/dev/null | line "$a"
done | tee -a $o/bench_variants.txt
for m in word chunk; do DWGSIM_HIP_DIRTY_MAP=$m python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 33554432 --no-legs --no-cpu-baseline 2>/dev/null | line "genome-24-groups,no-pipeline,dirty-map-$m"; done | tee -a $o/bench_variants.txt
for gb in 2130706432 33554432; do
  out=$o/kt; rm -rf $out
  rocprofv3 --kernel-trace --stats -d $out -- python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp $gb --no-pipeline > /dev/null 2>&1
  python tools/rocprof_summary.py "$(find $out -name '*.db' | head -1)" | head -30 > $o/walk_kernels_genome_$gb.txt; head -14 $o/walk_kernels_genome_$gb.txt
  rm -rf $out
done
