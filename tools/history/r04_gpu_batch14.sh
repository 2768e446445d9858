#!/bin/bash
# tools/r04_gpu_batch14.sh -- analysis only (gpurun): bench lines of the final round-4 library: default with all legs, E. coli, 2 / 8 ranks sharing the GPU,
# S4 strong, assembly5k, Ion Torrent, smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b14; mkdir -p $o
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $o/build_smoke.log 2>&1; tail -1 $o/build_smoke.log
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
run() { name=$1; shift; timeout 900 python bench.py "$@" > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run default --steps 20 --warmup 5
run n1 --steps 50 --no-legs --no-cpu-baseline
run n1_ecoli --workload ecoli --steps 100 --no-legs --no-cpu-baseline
run n1_assembly5k --workload assembly5k --steps 10 --no-legs --no-cpu-baseline
run n2_weak_share --gpus 2 --share-gpu --steps 20 --no-legs --no-cpu-baseline
run n8_weak_share --gpus 8 --share-gpu --steps 5 --warmup 2 --no-legs --no-cpu-baseline
run n1_strong_grch38 --mode strong --workload grch38 --steps 2 --warmup 1 --no-legs --no-cpu-baseline
run n8_strong_grch38_share --gpus 8 --share-gpu --mode strong --workload grch38 --steps 1 --warmup 1 --no-legs --no-cpu-baseline
run ion_chr20 --ion --steps 10 --no-legs --no-cpu-baseline
run ion_ecoli --ion --workload ecoli --steps 20 --no-legs --no-cpu-baseline
cat $o/default.json | tail -1
