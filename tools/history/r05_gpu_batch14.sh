#!/bin/bash
# tools/r05_gpu_batch14.sh -- analysis only (gpurun): both output families from one FIFO image (-o 0), the priority dropped after the name line, launches carried
# across the steps of bench.py, the walk stream's priority: parity first, then the timings that decide what is adopted
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b14; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -3 $o/pytest.log
bash tools/variant_build.sh prio0 "-DDW_PRIO_DROP=0" > $o/variant.log 2>&1; tail -1 $o/variant.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2; do
  $B 2>/dev/null | line "carry,walk-low"
  $B --no-carry 2>/dev/null | line "no-carry,walk-low"
  DWGSIM_HIP_WALK_PRIO=high $B 2>/dev/null | line "carry,walk-high"
  DWGSIM_HIP_WALK_PRIO=high $B --no-carry 2>/dev/null | line "no-carry,walk-high"
  DWGSIM_HIP_WALK_PRIO=mid $B 2>/dev/null | line "carry,walk-mid"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_prio0.so $B 2>/dev/null | line "carry,walk-low,prio-drop-0"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_prio0.so DWGSIM_HIP_WALK_PRIO=high $B 2>/dev/null | line "carry,walk-high,prio-drop-0"
done 2>&1 | tee $o/bench_variants.txt
for wl in ecoli grch38; do
  for p in low high; do DWGSIM_HIP_WALK_PRIO=$p python bench.py --workload $wl $([ $wl = grch38 ] && echo "--mode strong --steps 3 --warmup 1" || echo "--steps 50 --warmup 3") --no-legs --no-cpu-baseline 2>/dev/null | line "$wl,walk-$p"; done
done 2>&1 | tee -a $o/bench_variants.txt
for p in low high; do DWGSIM_HIP_WALK_PRIO=$p python bench.py --workload chr20 --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | line "ion,walk-$p"; done 2>&1 | tee -a $o/bench_variants.txt
# -o 0 and friends as lone launches (tools/time_probe.py), then the bench line
for fl in "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 100 -2 100 -C 30 -o 0" "-z 13 -1 150 -2 0 -C 30 -o 0" "-z 13 -1 200 -2 200 -C 30 -o 0" "-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.01 -o 0"; do
  timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1
done | tee $o/o0_probe.txt
for w in 0 1; do WRITER=$w timeout 300 python tools/time_probe.py "-z 13 -1 250 -2 250 -C 30 -o 0" 2>&1 | tail -1; WRITER=$w timeout 300 python tools/time_probe.py "-z 13 -1 200 -2 200 -C 30 -o 0" 2>&1 | tail -1; done | tee -a $o/o0_probe.txt
ONLY="o0" bash tools/r05_final_profiles.sh > $o/final_o0.log 2>&1; grep -m1 "k_simulate" gpurun_out/final/r05_o0_kernel_stats_pmc.txt | cut -c1-200
PROBE_TRACE=0 PROBE_VARIANTS="default;DWGSIM_HIP_WALK_PRIO=high" timeout 600 python tools/r05_genome_probe.py 2>&1 | tee $o/genome_probe.txt
