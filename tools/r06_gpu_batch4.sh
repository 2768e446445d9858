#!/bin/bash
# tools/r06_gpu_batch4.sh -- analysis only (gpurun): with the walk's site draws as gap chains too: (1) lines, (2) the walk kernel by kernel, (3) the solo-rank sweep,
# (4) the -m gpu suite
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/r06b4; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['roofline']['frac'])"; }
{
python bench.py --workload chr20 --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150"
python bench.py --workload chr20 --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150"
python bench.py --workload ecoli --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,2x150"
python bench.py --workload assembly5k --steps 20 --no-legs --no-cpu-baseline 2>/dev/null | line "assembly5k,2x150"
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "chr20,ion400"
python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "ecoli,ion400"
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432 --no-legs --no-cpu-baseline 2>/dev/null | line "genome-two-groups,no-pipeline(walk alone)"
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 33554432 --no-legs --no-cpu-baseline 2>/dev/null | line "genome-24-groups,no-pipeline(walk alone)"
python bench.py --workload chr20 --steps 10 --no-legs --no-cpu-baseline --no-pipeline 2>/dev/null | line "chr20,2x150,no-pipeline(walk alone)"
python bench.py --workload chr20_like --steps 10 --no-legs --no-cpu-baseline --no-pipeline 2>/dev/null | line "chr20_like,2x150,no-pipeline(walk alone)"
} | tee $o/lines.txt
rocprofv3 --kernel-trace --stats -d $o/kt -- python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432 --no-legs --no-cpu-baseline > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $o/kt -name '*.db' | head -1)" > $o/genome_walk_kernel_stats.txt 2>&1; rm -rf $o/kt; head -32 $o/genome_walk_kernel_stats.txt
timeout 900 python bench.py --solo-sweep 2,4,8 --no-cpu-baseline --no-legs > $o/solo_sweep.json 2> $o/solo_sweep.err; python - <<PY
import json; d=json.load(open("$o/solo_sweep.json"))
for mode in ("weak","strong"):
    for W,v in d[mode].items():
        if W=="job": continue
        print(mode, "W", W, "max", v["max_ms_per_step"], "min", v["min_ms_per_step"], "eff", v["efficiency"])
PY
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; tail -3 $o/pytest_gpu.log; fi
