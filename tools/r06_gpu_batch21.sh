#!/bin/bash
# tools/r06_gpu_batch21.sh -- (gpurun) the round's final library (one look-back): the bench lines of every workload the documents quote, the solo-rank sweep
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b21; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['roofline']['frac'])"; }
{
python bench.py --workload chr20 --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150"
python bench.py --workload chr20 --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150"
python bench.py --workload chr20_like --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20_like,2x150"
python bench.py --workload ecoli --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,2x150"
python bench.py --workload ecoli_like --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli_like,2x150"
python bench.py --workload assembly5k --steps 20 --no-legs --no-cpu-baseline 2>/dev/null | line "assembly5k,2x150"
python bench.py --workload chr20 --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -1 50 -2 50 -C 30 -o 1" 2>/dev/null | line "chr20,2x50"
python bench.py --workload chr20 --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -1 250 -2 250 -C 30 -o 1" 2>/dev/null | line "chr20,2x250"
python bench.py --workload chr20 --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -1 150 -2 150 -C 30 -o 0" 2>/dev/null | line "chr20,2x150,-o0"
python bench.py --workload chr20 --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -1 150 -2 150 -C 30 -o 0" 2>/dev/null | line "chr20,2x150,-o0"
python bench.py --workload chr20 --steps 40 --no-legs --no-cpu-baseline "--flags=-z 13 -c 1 -1 50 -2 50 -C 30 -o 0" 2>/dev/null | line "chr20,solid2x50,-o0"
python bench.py --workload chr20 --steps 40 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "chr20,ion400"
python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "ecoli,ion400"
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "genome,strong,N=1"
} | tee $o/lines.txt
timeout 900 python bench.py --solo-sweep 2,4,8 --no-cpu-baseline --no-legs > $o/solo_sweep.json 2> $o/solo_sweep.err; python - <<PY | tee $o/solo_sweep.txt
import json; d=json.load(open("$o/solo_sweep.json"))
for mode in ("weak","strong"):
    print("##", mode, d[mode].get("job"))
    for W,v in d[mode].items():
        if W=="job": continue
        print(mode, "W", W, "max", v["max_ms_per_step"], "min", v["min_ms_per_step"], "eff", v["efficiency"])
PY
