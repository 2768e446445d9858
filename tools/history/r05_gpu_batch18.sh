#!/bin/bash
# tools/r05_gpu_batch18.sh -- analysis only (gpurun): the quality line laid down by two-byte LDS stores (quality_line_fifo): parity, then A/B on one box against the
# same library with the pairs compacted in registers (-DDW_QUAL_FIFO=0) and with the rare paths called (-DDW_CALL_RARE)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b18; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -3 $o/pytest.log
( bash tools/variant_build.sh qreg "-DDW_QUAL_FIFO=0"; bash tools/variant_build.sh call "-DDW_CALL_RARE=1" ) > $o/variant.log 2>&1; grep built $o/variant.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2 3; do
  $B 2>/dev/null | line "product"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_qreg.so $B 2>/dev/null | line "quality-pairs-compacted-in-registers"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_call.so $B 2>/dev/null | line "rare-paths-called"
done 2>&1 | tee $o/bench_variants.txt
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 250 -2 250 -C 30 -o 1"; do
  for v in product qreg call product; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; DWGSIM_HIP_LIB=$lib timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$v /"; done
done | tee $o/probe.txt
for a in "--workload ecoli" "--workload chr20 --ion --steps 10" "--workload ecoli --ion" "--workload chr20 --flags='-z 13 -1 150 -2 150 -C 30 -o 0'" "--workload chr20 --flags='-z 13 -1 50 -2 50 -C 30 -o 1'"; do eval "python bench.py $a --no-legs --no-cpu-baseline" 2>/dev/null | line "$a"; done | tee -a $o/bench_variants.txt
ONLY="chr20" bash tools/r05_final_profiles.sh > $o/final.log 2>&1; grep -m2 "k_simulate" gpurun_out/final/r05_chr20_kernel_stats_pmc.txt | cut -c1-200
