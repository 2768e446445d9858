#!/bin/bash
# tools/r05_gpu_batch19.sh -- analysis only (gpurun): quality line, second form (base qualities loaded before the draws, picked by v_perm_b32; pairs laid down by
# two-byte LDS stores): parity, A/B against -DDW_QUAL_FIFO=0 on one box; the default bench line with every leg; two ranks sharing the GPU
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b19; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -3 $o/pytest.log
bash tools/variant_build.sh qreg "-DDW_QUAL_FIFO=0" > $o/variant.log 2>&1; grep built $o/variant.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['breakdown_ms']['host_wait_for_walks'])"; }
B="python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3"
for rep in 1 2 3; do
  $B 2>/dev/null | line "product"
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_qreg.so $B 2>/dev/null | line "quality-pairs-compacted-in-registers"
done 2>&1 | tee $o/bench_variants.txt
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 250 -2 250 -C 30 -o 1" "-z 13 -1 100 -2 100 -C 30 -o 1"; do
  for v in product qreg product qreg; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; DWGSIM_HIP_LIB=$lib timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$v /"; done
done | tee $o/probe.txt
for a in "--workload ecoli" "--workload chr20 --ion --steps 10" "--workload ecoli --ion"; do
  eval "python bench.py $a --no-legs --no-cpu-baseline" 2>/dev/null | line "product $a"; eval "DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_qreg.so python bench.py $a --no-legs --no-cpu-baseline" 2>/dev/null | line "qreg $a"
done | tee -a $o/bench_variants.txt
( time python bench.py ) > $o/default_line.json 2> $o/default_line.err; tail -4 $o/default_line.err; cut -c1-300 $o/default_line.json
( time python bench.py --gpus 2 --share-gpu --no-legs --no-cpu-baseline --steps 20 ) > $o/two_ranks.json 2> $o/two_ranks.err; tail -4 $o/two_ranks.err; cut -c1-300 $o/two_ranks.json
ONLY="chr20" bash tools/r05_final_profiles.sh > $o/final.log 2>&1; grep -m2 "k_simulate" gpurun_out/final/r05_chr20_kernel_stats_pmc.txt | cut -c1-200
