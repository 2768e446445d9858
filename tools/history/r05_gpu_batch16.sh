#!/bin/bash
# tools/r05_gpu_batch16.sh -- analysis only (gpurun): bench.py with the walks two steps ahead (--depth 2) against one (rounds 3-4), timeline of the steps; the rare
# paths of k_simulate called against inlined (-DDW_INLINE_RARE) on one box; scratch-slot sweep for 2 000-base reads; whole-genome strong line, walk kernels
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b16; mkdir -p $o
bash tools/variant_build.sh inl "-DDW_INLINE_RARE=1" > $o/variant.log 2>&1; grep built $o/variant.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2 3; do
  $B 2>$o/err.txt | line "depth-2"
  $B --depth 1 2>/dev/null | line "depth-1"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_inl.so $B 2>/dev/null | line "depth-2,rare-paths-inlined"
done 2>&1 | tee $o/bench_variants.txt
tail -3 $o/err.txt
for wl in ecoli assembly5k; do for d in 2 1; do python bench.py --workload $wl --depth $d --steps 50 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | line "$wl,depth-$d"; done; done 2>&1 | tee -a $o/bench_variants.txt
for d in 2 1; do python bench.py --workload chr20 --ion --depth $d --no-legs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | line "ion,depth-$d"; done 2>&1 | tee -a $o/bench_variants.txt
for d in 2 1; do python bench.py --workload ecoli --ion --depth $d --no-legs --no-cpu-baseline --steps 20 --warmup 2 2>/dev/null | line "ion-ecoli,depth-$d"; done 2>&1 | tee -a $o/bench_variants.txt
for d in 2 1; do python bench.py --workload grch38 --mode strong --depth $d --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "grch38-strong,depth-$d"; done 2>&1 | tee -a $o/bench_variants.txt
for gb in 33554432 2130706432; do python bench.py --workload grch38 --mode strong --no-pipeline --group-bp $gb --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "grch38-strong,no-pipeline,group-bp-$gb"; done 2>&1 | tee -a $o/bench_variants.txt
out=$o/tl; rm -rf $out
rocprofv3 --kernel-trace --memory-copy-trace -d $out -- python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
python tools/step_timeline.py "$(find $out -name '*.db' | head -1)" 2 > $o/timeline_depth2.txt 2>&1; tail -4 $o/timeline_depth2.txt
rm -rf $out
out=$o/kt; rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -- python bench.py --workload grch38 --mode strong --no-legs --no-cpu-baseline --steps 3 --warmup 1 --group-bp 2130706432 --no-pipeline > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $out -name '*.db' | head -1)" | head -40 > $o/walk_kernels_genome.txt; head -12 $o/walk_kernels_genome.txt
rm -rf $out
for sl in 0 64 128 256 384 512; do FLOW_SLOTS=$sl timeout 300 python tools/time_probe.py "-z 13 -1 2000 -2 0 -C 30 -o 1" 2>&1 | tail -1 | sed "s/^/slots-per-xcd $sl /"; done | tee $o/long_slots.txt
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0"; do
  for v in product inl product inl; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; DWGSIM_HIP_LIB=$lib timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$v /"; done
done | tee $o/probe.txt
