#!/bin/bash
# tools/r05_gpu_batch20.sh -- analysis only (gpurun): the walk's two heavy kernels re-mapped (k_site_scan_list: 1024 lanes, four tiles at a time; k_dirty_chunks:
# a lane per chunk): parity (whole suite), the walk alone on the genome, bench lines, timeline
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b20; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -3 $o/pytest.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2 3; do $B 2>/dev/null | line "default"; $B --depth 1 2>/dev/null | line "default,depth-1"; done 2>&1 | tee $o/bench_variants.txt
for a in "--workload ecoli --steps 50" "--workload assembly5k --steps 30" "--workload chr20 --ion --steps 10" "--workload ecoli --ion" "--workload grch38 --mode strong --steps 3 --warmup 1" "--workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432" "--workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 33554432"; do
  eval "python bench.py $a --no-legs --no-cpu-baseline" 2>/dev/null | line "$a"
done | tee -a $o/bench_variants.txt
out=$o/kt; rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -- python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp 2130706432 --no-pipeline > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $out -name '*.db' | head -1)" | head -30 > $o/walk_kernels_genome.txt; head -14 $o/walk_kernels_genome.txt
rm -rf $out
out=$o/tl; rm -rf $out
rocprofv3 --kernel-trace --memory-copy-trace -d $out -- python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
python tools/step_timeline.py "$(find $out -name '*.db' | head -1)" 2 > $o/timeline_depth2.txt 2>&1; tail -3 $o/timeline_depth2.txt
rm -rf $out
PROBE_TRACE=0 PROBE_VARIANTS="default" timeout 600 python tools/r05_genome_probe.py 2>&1 | tee $o/genome_probe.txt
