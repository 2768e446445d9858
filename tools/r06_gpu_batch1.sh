#!/bin/bash
# tools/r06_gpu_batch1.sh -- analysis only (gpurun): the round's first GPU pass.  (1) the -m gpu suite with the round's new cases, (2) the default bench line
# on this box, (3) the solo-rank sweep (every rank of W = 2, 4, 8 alone on this GPU: weak line + whole-genome strong job), (4) the genome-like workloads
# beside the uniform ones, (5) the job level as device r of 8 (DWGSIM_HIP_SOLO) and the delivery threads' rate with a memcpy sink at 8 contexts
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/r06b1; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; tail -3 $o/pytest_gpu.log; fi
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err; python - <<PY
import json; d=json.load(open("$o/bench_default.json")); print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("strong",{}).get("value"), d.get("host_landed_gz",{}).get("value"), d.get("end_to_end_genome",{}).get("seconds"))
PY
timeout 900 python bench.py --solo-sweep 2,4,8 --no-cpu-baseline --no-legs > $o/solo_sweep.json 2> $o/solo_sweep.err; python - <<PY
import json; d=json.load(open("$o/solo_sweep.json"))
for mode in ("weak","strong"):
    for W,v in d[mode].items():
        if W=="job": print(mode, v); continue
        print(mode, "W", W, "max", v["max_ms_per_step"], "min", v["min_ms_per_step"], "eff", v["efficiency"])
PY
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['roofline']['frac'])"; }
ION="--ion"; O0="--flags=-z 13 -1 150 -2 150 -C 30 -o 0"
for wl in chr20 chr20_like ecoli ecoli_like; do
  python bench.py --workload $wl --steps 50 --no-legs --no-cpu-baseline 2>/dev/null | line "$wl,2x150"
  python bench.py --workload $wl --steps 30 --no-legs --no-cpu-baseline "$O0" 2>/dev/null | line "$wl,2x150-o0"
  python bench.py --workload $wl --steps 30 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "$wl,ion400"
  python bench.py --workload $wl --steps 10 --no-legs --no-cpu-baseline --no-pipeline 2>/dev/null | line "$wl,2x150,no-pipeline(walk alone)"
done | tee $o/genome_like_lines.txt
# (5) the job level
python - <<PY
import sys, time; sys.path.insert(0, "$R")
from dwgsim_amd import synth
t=time.time(); c = synth.workload_contigs("grch38"); synth.write_fasta("/dev/shm/g38.fa", c)
with open("/dev/shm/g38.fa.fai", "w") as f:
    off = 0
    for name, arr in c:
        off += len(name) + 2; f.write(f"{name}\t{len(arr)}\t{off}\t60\t61\n"); off += len(arr) + (len(arr) + 59) // 60
print("genome written", time.time() - t)
PY
FL="-z 13 -1 150 -2 150 -C 30 -o 1"
{
for rep in 1 2; do
  echo "== whole job on this device (W = 1), null sink"; DWGSIM_HIP_SINK=null DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
done
for W in 2 4 8; do for r in $(seq 0 $((W-1))); do
  echo "== solo $r/$W, null sink"; DWGSIM_HIP_SOLO=$r/$W DWGSIM_HIP_SINK=null DWGSIM_HIP_TIMING=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip\]"
done; done
echo "== memcpy sink, 1 context"; DWGSIM_HIP_SINK=memcpy DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
echo "== memcpy sink, 8 contexts on this GPU"; DWGSIM_HIP_SINK=memcpy DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=0,0,0,0,0,0,0,0 timeout 300 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
echo "== memcpy sink, 8 contexts, raw text (DWGSIM_HIP_GZIP=cpu would deflate: not this) -- skipped"
} | tee $o/job_level_solo.txt
rm -f /dev/shm/g38.fa /dev/shm/g38.fa.fai /dev/shm/out.*
