#!/bin/bash
# tools/r06_gpu_batch6.sh -- analysis only (gpurun): where the headline kernel's time goes now (flags that switch parts of the model off; knock-out builds)
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/r06b6; mkdir -p $o
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['roofline']['frac'])"; }
B="-z 13 -1 150 -2 150 -C 30 -o 1"
{
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline 2>/dev/null | line "default"
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=$B -Q 0" 2>/dev/null | line "-Q0(no quality normals)"
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=$B -q I" 2>/dev/null | line "-qI(fixed quality)"
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=$B -e 0 -E 0" 2>/dev/null | line "-e0(no errors)"
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=$B -y 0" 2>/dev/null | line "-y0(no random reads)"
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=$B -r 0" 2>/dev/null | line "-r0(no mutations)"
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=-z 13 -1 50 -2 50 -C 30 -o 1 -Q 0 -e 0 -E 0" 2>/dev/null | line "2x50,-Q0,-e0"
python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=-z 13 -1 50 -2 50 -C 30 -o 1" 2>/dev/null | line "2x50"
for k in 128 2048 1 16; do DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_knock$k.so python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline 2>/dev/null | line "knock$k"; done
DWGSIM_HIP_LIB=dwgsim_amd/libdwgsim_hip_var_knock1.so python bench.py --steps 50 --no-legs --no-cpu-baseline --no-pipeline "--flags=$B -Q 0" 2>/dev/null | line "knock1,-Q0"
} | tee $o/where.txt
