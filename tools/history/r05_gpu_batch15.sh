#!/bin/bash
# tools/r05_gpu_batch15.sh -- analysis only (gpurun): k_simulate with its rare paths called instead of inlined (94 -> 57 KB of code) and the name line's numbers
# without a loop per digit: parity, then A/B against variants built on the box (inlined ragged stores; priority dropped at the look-backs), the timeline of a step
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b15; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -3 $o/pytest.log
( bash tools/variant_build.sh inl "-DDW_DEV_NOINLINE=__device__\ __forceinline__" ; bash tools/variant_build.sh prio0 "-DDW_PRIO_DROP=0" ) > $o/variant.log 2>&1; grep built $o/variant.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2 3; do
  $B 2>/dev/null | line "product"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_inl.so $B 2>/dev/null | line "ragged-stores-inlined"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_prio0.so $B 2>/dev/null | line "prio-drop-at-look-backs"
done 2>&1 | tee $o/bench_variants.txt
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 250 -2 250 -C 30 -o 0" "-z 13 -1 100 -2 100 -C 30 -o 1" "-z 13 -1 50 -2 50 -C 30 -o 1"; do
  for v in product inl prio0 product; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; DWGSIM_HIP_LIB=$lib timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$v /"; done
done | tee $o/probe.txt
out=$o/tl; rm -rf $out
rocprofv3 --kernel-trace --memory-copy-trace -d $out -- python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
python tools/step_timeline.py "$(find $out -name '*.db' | head -1)" 3 > $o/timeline.txt 2>&1; tail -60 $o/timeline.txt
rm -rf $out
ONLY="chr20" bash tools/r05_final_profiles.sh > $o/final.log 2>&1; grep -m2 "k_simulate" gpurun_out/final/r05_chr20_kernel_stats_pmc.txt | cut -c1-200
