#!/bin/bash
# tools/r05_gpu_batch22.sh -- analysis only (gpurun): the round's final library: GPU suite, rocprofv3 passes of every profiled workload (profiles/r05_*_kernel_stats_pmc.txt,
# r05_counters.json), the default bench line with every leg, the other workloads' lines, two ranks sharing the GPU (the 8-rank readiness line: tools/r05_gpu_batch24.sh)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/b22; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so | tee $o/library_sha256.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; tail -3 $o/pytest.log
bash tools/r05_final_profiles.sh > $o/final.log 2>&1; tail -3 $o/final.log | cut -c1-200
{
echo "## python bench.py (every leg)"; python bench.py 2> $o/default.err
for a in "--workload ecoli" "--workload assembly5k --steps 30" "--workload chr20 --ion --steps 10 --warmup 2" "--workload ecoli --ion --steps 20 --warmup 2" "--workload grch38 --mode strong --steps 3 --warmup 1" "--workload chr20 --flags='-z 13 -1 150 -2 150 -C 30 -o 0'" "--workload chr20 --flags='-z 13 -1 50 -2 50 -C 30 -o 1'" "--workload chr20 --flags='-z 13 -c 1 -1 50 -2 50 -C 30 -o 0'" "--depth 1" "--no-pipeline"; do
  echo "## python bench.py $a --no-legs --no-cpu-baseline"; eval "python bench.py $a --no-legs --no-cpu-baseline" 2>/dev/null
done
echo "## python bench.py --gpus 2 --share-gpu --no-legs --no-cpu-baseline --steps 20   (two ranks on the ONE GPU: what N > 1 adds, not a speed-up)"; python bench.py --gpus 2 --share-gpu --no-legs --no-cpu-baseline --steps 20 2>/dev/null
} > $o/bench_lines.txt
grep -c metric $o/bench_lines.txt
python - <<'PY'
import json
for ln in open("gpurun_out/b22/bench_lines.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); print(d["n_gpus"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["workload"][:60], (d.get("strong") or {}).get("value"))
PY
PROBE_VARIANTS="default" timeout 900 python tools/r05_genome_probe.py > $o/genome_probe.txt 2>&1; grep "wall\|busy" $o/genome_probe.txt | cut -c1-200
