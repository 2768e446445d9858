#!/bin/bash
# tools/profile_round.sh <tag> [workload] [extra bench flags] -- analysis only: the rocprofv3 passes behind profiles/<tag>_*.txt (run on the
# GPU box through gpurun).  Pass 1: kernel trace + stats.  Further passes: PMC counters, each in its own run (never together with a
# trace domain other than --kernel-trace): FETCH_SIZE | WRITE_SIZE | SQ issue counters | SQ instruction mix | GRBM_GUI_ACTIVE.
set -u
tag=${1:-prof}; wl=${2:-chr20}; extra=${3:-}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
out=gpurun_out/$tag; mkdir -p "$out"
BENCH="python bench.py --workload $wl --no-legs --no-cpu-baseline $extra"
eval "$BENCH --steps 20 --warmup 3" > "$out/bench_line.json" 2> "$out/bench.err"      # (eval: extra may carry a quoted --flags='...')
eval "rocprofv3 --kernel-trace --stats -d $out/kt -- $BENCH --steps 10 --warmup 2" > /dev/null 2>&1
python tools/rocprof_summary.py "$(find "$out/kt" -name '*.db' | head -1)" > "$out/kernel_stats.txt" 2>&1
rm -rf "$out/kt"      # (the databases are tens of MB each; gpurun brings back at most 64 MiB)
i=0
: > "$out/pmc.txt"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    eval "rocprofv3 --kernel-trace --pmc $pmc -d $out/pmc$i -- $BENCH --steps 4 --warmup 1" > "$out/pmc$i.log" 2>&1
    python tools/pmc_summary.py $(find "$out/pmc$i" -name '*.db') >> "$out/pmc.txt" 2>&1
    rm -rf "$out/pmc$i"
done
cat "$out/bench_line.json"; head -12 "$out/kernel_stats.txt"; cat "$out/pmc.txt"
