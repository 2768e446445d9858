#!/bin/bash
# tools/r06_gpu_batch5.sh -- analysis only (gpurun): the error sites of a read end as a gap chain (headline kernel): lines, instruction counters, the -m gpu suite
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/r06b5; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['roofline']['frac'])"; }
O0="--flags=-z 13 -1 150 -2 150 -C 30 -o 0"
{
python bench.py --workload chr20 --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150"
python bench.py --workload chr20 --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150"
python bench.py --workload ecoli --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,2x150"
python bench.py --workload chr20_like --steps 50 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20_like,2x150"
python bench.py --workload assembly5k --steps 20 --no-legs --no-cpu-baseline 2>/dev/null | line "assembly5k,2x150"
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline "$O0" 2>/dev/null | line "chr20,2x150-o0"
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline "--flags=-z 13 -1 50 -2 50 -C 30 -o 1" 2>/dev/null | line "chr20,2x50"
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline "--flags=-z 13 -c 1 -1 50 -2 50 -C 30 -o 0" 2>/dev/null | line "chr20,solid2x50-o0"
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline "--flags=-z 13 -1 150 -2 150 -C 30 -o 1 -e 0.001-0.05 -E 0.02" 2>/dev/null | line "chr20,2x150,ramp"
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "chr20,ion400"
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "genome,strong,N=1"
} | tee $o/lines.txt
rocprofv3 --kernel-trace --stats -d $o/kt -- python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $o/kt -name '*.db' | head -1)" > $o/chr20_kernel_stats.txt 2>&1; rm -rf $o/kt; head -6 $o/chr20_kernel_stats.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES -d $o/pmc -- python bench.py --no-legs --no-cpu-baseline --steps 4 --warmup 1 > $o/pmc.log 2>&1
python tools/pmc_summary.py $(find $o/pmc -name '*.db') > $o/chr20_pmc.txt 2>&1; rm -rf $o/pmc; cat $o/chr20_pmc.txt
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; tail -3 $o/pytest_gpu.log; fi
