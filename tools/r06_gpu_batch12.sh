#!/bin/bash
# tools/r06_gpu_batch12.sh -- (gpurun) the round's final library: the fuzz campaign (new seeds), the solo-rank sweep, the remaining bench lines
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b12; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['roofline']['frac'])"; }
{
python bench.py --workload chr20_like --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20_like,2x150"
python bench.py --workload ecoli --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,2x150"
python bench.py --workload assembly5k --steps 20 --no-legs --no-cpu-baseline 2>/dev/null | line "assembly5k,2x150"
python bench.py --workload chr20 --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -1 50 -2 50 -C 30 -o 1" 2>/dev/null | line "chr20,2x50"
python bench.py --workload chr20 --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -1 250 -2 250 -C 30 -o 1" 2>/dev/null | line "chr20,2x250"
python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "ecoli,ion400"
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-legs --no-cpu-baseline 2>/dev/null | line "genome,strong,N=1"
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432 --no-legs --no-cpu-baseline 2>/dev/null | line "genome-two-groups,no-pipeline(walk alone)"
} | tee $o/lines.txt
timeout 900 python bench.py --solo-sweep 2,4,8 --no-cpu-baseline --no-legs > $o/solo_sweep.json 2> $o/solo_sweep.err; python - <<PY | tee $o/solo_sweep.txt
import json; d=json.load(open("$o/solo_sweep.json"))
for mode in ("weak","strong"):
    print("##", mode, d[mode].get("job"))
    for W,v in d[mode].items():
        if W=="job": continue
        print(mode, "W", W, "max", v["max_ms_per_step"], "min", v["min_ms_per_step"], "eff", v["efficiency"], "by fastest", v.get("efficiency_fastest"))
PY
bash tools/r06_fuzz_campaign.sh > $o/fuzz_campaign.log 2>&1; cp gpurun_out/r06_fuzz/fuzz.txt $o/fuzz.txt; cat $o/fuzz.txt
