#!/bin/bash
# tools/r05_gpu_batch26.sh -- analysis only (gpurun): k_simulate in 128-lane blocks (the whole library built with -DDW_SIM_THREADS=128 in a scratch copy): parity of a
# subset, then lone launches against the product on one box
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/b26; mkdir -p $o
rm -rf /tmp/v128 && mkdir -p /tmp/v128/dwgsim_amd && cp -r dwgsim_amd/csrc /tmp/v128/dwgsim_amd/ && cp -r include /tmp/v128/ && rm -rf /tmp/v128/dwgsim_amd/csrc/build
( cd /tmp/v128/dwgsim_amd/csrc && make -s -j16 FLAGS="--offload-arch=gfx950 -I. -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result -Wno-unused-value -DDW_SIM_THREADS=128" ../libdwgsim_hip.so ) > $o/build.log 2>&1
cp /tmp/v128/dwgsim_amd/libdwgsim_hip.so dwgsim_amd/libdwgsim_hip_var_t128.so && echo built t128
DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_t128.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_option_surface or contigs_resident or both_forms or writers or names" > $o/pytest_t128.log 2>&1; tail -3 $o/pytest_t128.log
for wl in chr20 ecoli; do
for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 100 -2 100 -C 30 -o 1" "-z 13 -1 50 -2 50 -C 30 -o 1" "-z 13 -1 250 -2 250 -C 30 -o 1"; do
  for v in product t128 product t128; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; WL=$wl DWGSIM_HIP_LIB=$lib timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | sed "s/^/$wl $v /"; done
done; done | tee $o/probe.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'])"; }
for rep in 1 2; do for v in product t128; do lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_var_$v.so; DWGSIM_HIP_LIB=$lib python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 3 2>/dev/null | line "default,$v"; done; done | tee $o/bench_variants.txt
