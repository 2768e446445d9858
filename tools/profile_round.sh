#!/bin/bash
# tools/profile_round.sh <tag> -- analysis only: the rocprofv3 passes behind profiles/<tag>_*.txt (run on the GPU box through gpurun).
# Pass 1: kernel trace + stats.  Passes 2-5: PMC counters, each in its own run (FETCH_SIZE | WRITE_SIZE | SQ issue counters | GRBM_GUI_ACTIVE).
set -u
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/$tag; mkdir -p "$out"
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
python bench.py --steps 20 --warmup 3 > "$out/bench_line.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats -d "$out/kt" -- $BENCH > /dev/null 2>&1
python tools/rocprof_summary.py "$(find "$out/kt" -name '*.db' | head -1)" > "$out/kernel_stats.txt" 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $pmc -d "$out/pmc$i" -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python tools/pmc_summary.py $(find "$out/pmc$i" -name '*.db') >> "$out/pmc.txt" 2>&1
done
cat "$out/bench_line.json"; head -12 "$out/kernel_stats.txt"; cat "$out/pmc.txt"
