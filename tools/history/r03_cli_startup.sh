#!/bin/bash
# tools/r03_cli_startup.sh -- analysis only (gpurun): what dwgsim-hip costs around its work (process start, runtime, contexts, exit) on a job of 1000 pairs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import sys; sys.path.insert(0, '.')
from dwgsim_amd import synth
synth.write_fasta('/tmp/tiny.fa', [('c1', synth.random_contig(200000, 3, []))])
PY
B=dwgsim_amd/dwgsim-hip
for i in 1 2 3; do
  ( TIMEFORMAT="wall %R s  user %U  sys %S"; time DWGSIM_HIP_TIMING=1 $B -z 5 -N 1000 -1 100 -2 100 /tmp/tiny.fa /tmp/tiny_out ) 2>&1 | grep -E "wall|dwgsim-hip\]"
done
echo "-- python -c pass (process start of an interpreter, for scale)"; ( TIMEFORMAT="wall %R s"; time python -c pass ) 2>&1
echo "-- a C program that only calls hipGetDeviceCount + hipFree(0)"
cat > /tmp/hipinit.cpp <<'CPP'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
int main(){ auto t0=std::chrono::steady_clock::now(); int n=0; hipGetDeviceCount(&n); auto t1=std::chrono::steady_clock::now(); hipSetDevice(0); hipFree(0); auto t2=std::chrono::steady_clock::now();
 void*p; hipMalloc(&p, 1<<20); auto t3=std::chrono::steady_clock::now(); void*h; hipHostMalloc(&h, 256<<20, 0); auto t4=std::chrono::steady_clock::now();
 auto d=[](auto a, auto b){return std::chrono::duration<double>(b-a).count();};
 printf("count %.3f  setdevice+free(0) %.3f  malloc %.3f  hostmalloc256M %.3f\n", d(t0,t1), d(t1,t2), d(t2,t3), d(t3,t4)); if (getenv("FAST_EXIT")) _exit(0); return 0; }
CPP
/opt/rocm/bin/hipcc -O2 /tmp/hipinit.cpp -o /tmp/hipinit 2>/dev/null
for i in 1 2; do ( TIMEFORMAT="wall %R s"; time /tmp/hipinit ) 2>&1; done
( TIMEFORMAT="wall %R s (fast exit)"; time FAST_EXIT=1 /tmp/hipinit ) 2>&1
echo "-- the chr20-sized job (64 Mb, 6.78 M pairs, .gz members to /dev/shm)"
python - <<'PY'
import sys; sys.path.insert(0, '.')
from dwgsim_amd import synth
synth.write_fasta('/dev/shm/chr20.fa', synth.workload_contigs('chr20'))
PY
for i in 1 2 3; do
  ( TIMEFORMAT="wall %R s  user %U  sys %S"; time DWGSIM_HIP_TIMING=1 $B -z 13 -1 150 -2 150 -C 30 -o 1 /dev/shm/chr20.fa /dev/shm/chr20_out ) 2>&1 | grep -E "wall|dwgsim-hip\]"
done
for bp in 524288 262144 131072; do
  ( TIMEFORMAT="wall %R s  user %U  sys %S (DWGSIM_HIP_BATCH=$bp)"; time DWGSIM_HIP_BATCH=$bp DWGSIM_HIP_TIMING=1 $B -z 13 -1 150 -2 150 -C 30 -o 1 /dev/shm/chr20.fa /dev/shm/chr20_out ) 2>&1 | grep -E "wall|dwgsim-hip\]"
done
( TIMEFORMAT="wall %R s  user %U  sys %S (null sink)"; time DWGSIM_HIP_SINK=null DWGSIM_HIP_TIMING=1 $B -z 13 -1 150 -2 150 -C 30 -o 1 /dev/shm/chr20.fa /dev/shm/chr20_out ) 2>&1 | grep -E "wall|dwgsim-hip\]"
( TIMEFORMAT="wall %R s (with DWGSIM_HIP_TEARDOWN)"; time DWGSIM_HIP_TEARDOWN=1 $B -z 13 -1 150 -2 150 -C 30 -o 1 /dev/shm/chr20.fa /dev/shm/chr20_out ) 2>&1 | grep -E "wall"
rm -f /dev/shm/chr20.fa /dev/shm/chr20_out*
