#!/bin/bash
# tools/r05_gpu_batch25.sh -- analysis only (gpurun): bench.py with the walks three steps ahead on depth + 2 resident copies and no wait between a step's launches and
# the next step's preparation (library unchanged: 4a5ba67b): lines at --depth 3 / 2 / 1, gaps between launches, two ranks sharing the GPU
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b25; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'], (d.get('strong') or {}).get('value'))"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2 3; do for d in 3 2 1; do $B --depth $d 2>$o/err.txt | line "default,depth-$d"; done; done 2>&1 | tee $o/bench_variants.txt; tail -2 $o/err.txt
for a in "--workload ecoli --steps 50" "--workload assembly5k --steps 30" "--workload chr20 --ion --steps 10" "--workload grch38 --mode strong --steps 3 --warmup 1"; do
  for d in 3 2; do eval "python bench.py $a --depth $d --no-legs --no-cpu-baseline" 2>/dev/null | line "$a --depth $d"; done
done | tee -a $o/bench_variants.txt
for d in 3 2; do python bench.py --gpus 2 --share-gpu --depth $d --no-legs --no-cpu-baseline --steps 20 2>/dev/null | grep -v Gloo | line "two-ranks-sharing-the-gpu,depth-$d"; done | tee -a $o/bench_variants.txt
out=$o/tl; rm -rf $out
rocprofv3 --kernel-trace --memory-copy-trace -d $out -- python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
python tools/step_timeline.py "$(find $out -name '*.db' | head -1)" 3 > $o/timeline_depth3.txt 2>&1; tail -2 $o/timeline_depth3.txt
rm -rf $out
( time python bench.py ) > $o/default_line.json 2> $o/default_line.err; tail -4 $o/default_line.err; cut -c1-200 $o/default_line.json
