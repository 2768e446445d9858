#!/bin/bash
# tools/r05_gpu_batch27.sh -- analysis only (gpurun): k_simulate in 320- / 384- / 512-lane blocks (the whole library built with -DDW_SIM_THREADS=... in scratch copies; 128 lanes:
# tools/r05_gpu_batch26.sh): fewer, wider blocks = a shorter look-back chain; parity of a subset, then the bench line against the product on one box
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/b27; mkdir -p $o
for T in 320 384 512; do
  rm -rf /tmp/v$T && mkdir -p /tmp/v$T/dwgsim_amd && cp -r dwgsim_amd/csrc /tmp/v$T/dwgsim_amd/ && cp -r include /tmp/v$T/ && rm -rf /tmp/v$T/dwgsim_amd/csrc/build
  ( cd /tmp/v$T/dwgsim_amd/csrc && make -s -j16 FLAGS="--offload-arch=gfx950 -I. -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result -Wno-unused-value -DDW_SIM_THREADS=$T" ../libdwgsim_hip.so ) > $o/build_$T.log 2>&1
  cp /tmp/v$T/dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_t$T.so && echo built t$T || tail -5 $o/build_$T.log
  DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_t$T.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_option_surface or contigs_resident or both_forms or writers or names" > $o/pytest_t$T.log 2>&1; tail -1 $o/pytest_t$T.log
done
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'])"; }
for rep in 1 2 3; do for v in product t320 t384 t512; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; [ -f $lib ] && DWGSIM_HIP_LIB=$lib python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3 2>/dev/null | line "default,$v"; done; done | tee $o/bench_variants.txt
for fl in "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 250 -2 250 -C 30 -o 1" "-z 13 -1 100 -2 100 -C 30 -o 1"; do
  for v in product t320 t384 t512; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; [ -f $lib ] && DWGSIM_HIP_LIB=$lib timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$v /"; done
done | tee $o/probe.txt
for v in product t320; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; [ -f $lib ] && DWGSIM_HIP_LIB=$lib python bench.py --workload ecoli --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,$v"; done | tee -a $o/bench_variants.txt
