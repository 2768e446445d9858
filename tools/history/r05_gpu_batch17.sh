#!/bin/bash
# tools/r05_gpu_batch17.sh -- analysis only (gpurun): does a walk beside a device-filling k_simulate have to take as long as that kernel?  Timelines and bench lines
# with the walk stream at / above the batches' priority
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b17; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2; do
  for p in low high above; do for d in 2 1; do DWGSIM_HIP_WALK_PRIO=$p $B --depth $d 2>/dev/null | line "walk-$p,depth-$d"; done; done
done 2>&1 | tee $o/bench_variants.txt
for p in high above; do
  out=$o/tl; rm -rf $out
  DWGSIM_HIP_WALK_PRIO=$p rocprofv3 --kernel-trace --memory-copy-trace -d $out -- python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
  python tools/step_timeline.py "$(find $out -name '*.db' | head -1)" 2 > $o/timeline_$p.txt 2>&1; tail -2 $o/timeline_$p.txt
  rm -rf $out
done
for p in low above; do
  DWGSIM_HIP_WALK_PRIO=$p python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "grch38-strong,walk-$p"
  DWGSIM_HIP_WALK_PRIO=$p python bench.py --workload assembly5k --steps 30 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | line "assembly5k,walk-$p"
  DWGSIM_HIP_WALK_PRIO=$p python bench.py --workload ecoli --steps 50 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,walk-$p"
done 2>&1 | tee -a $o/bench_variants.txt
PROBE_TRACE=0 PROBE_VARIANTS="default;DWGSIM_HIP_WALK_PRIO=above" timeout 600 python tools/r05_genome_probe.py 2>&1 | tee $o/genome_probe.txt
