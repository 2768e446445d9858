#!/bin/bash
# tools/knockout.sh -- analysis only: k_simulate time of the chr20 job with parts of the per-read work switched off by dwgsim flags
# (the kernel is VALU-issue bound, so the time differences are the parts' VALU shares)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="-z 13 -1 150 -2 150 -C 30 -o 1"
for v in "$B" "$B -Q 0" "$B -q I" "$B -y 0" "$B -y 0.999" "$B -e 0 -E 0" "-z 13 -1 150 -2 0 -C 15 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 150 -2 150 -C 30 -o 2" "-z 13 -1 50 -2 50 -C 10 -o 1"; do
    python bench.py --workload chr20 --no-legs --no-cpu-baseline --steps 10 --warmup 2 --flags "$v" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-48s pairs %9d  kernel %8.3f ms  %7.1f M pairs/s  text %6.1f B/pair' % ('$v', d['config']['pairs_per_gpu_per_step'], d['breakdown_ms']['simulate_kernels'], d['value'], d['config']['fastq_bytes_per_step_per_gpu']/d['config']['pairs_per_gpu_per_step']))"
done
