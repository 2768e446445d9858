#!/bin/bash
# tools/r05_gpu_batch28.sh -- analysis only (gpurun): k_site_scan_list with 8 / 32 tiles per block instead of 16 (scratch copies of the sources, sed-ed): the walk alone on
# the genome (two groups) and the default bench line, against the product on one box
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/b28; mkdir -p $o
for T in 8 32; do
  rm -rf /tmp/vs$T && mkdir -p /tmp/vs$T/dwgsim_amd && cp -r dwgsim_amd/csrc /tmp/vs$T/dwgsim_amd/ && cp -r include /tmp/vs$T/ && rm -rf /tmp/vs$T/dwgsim_amd/csrc/build
  sed -i "s/constexpr int SITE_TILES = 16,/constexpr int SITE_TILES = $T,/" /tmp/vs$T/dwgsim_amd/csrc/dw_walk.hip
  ( cd /tmp/vs$T/dwgsim_amd/csrc && make -s -j16 ../libdwgsim_hip.so ) > $o/build_$T.log 2>&1
  cp /tmp/vs$T/dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_st$T.so && echo built st$T || tail -5 $o/build_$T.log
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_st$T.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_option_surface or contigs_resident or again" > $o/pytest_st$T.log 2>&1; tail -1 $o/pytest_st$T.log
done
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'])"; }
for v in product st8 st32 product st8 st32; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so
  DWGSIM_HIP_LIB=$lib python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432 --no-legs --no-cpu-baseline 2>/dev/null | line "genome-two-groups,no-pipeline,$v"
  DWGSIM_HIP_LIB=$lib python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 33554432 --no-legs --no-cpu-baseline 2>/dev/null | line "genome-24-groups,no-pipeline,$v"
  DWGSIM_HIP_LIB=$lib python bench.py --no-pipeline --steps 30 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,no-pipeline,$v"
  DWGSIM_HIP_LIB=$lib python bench.py --steps 50 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,default,$v"
done | tee $o/bench_variants.txt
