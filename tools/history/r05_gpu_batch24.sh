#!/bin/bash
# tools/r05_gpu_batch24.sh -- analysis only (gpurun): tools/r05_gpu_batch22.sh again on the library with the SOLiD occupancy hint (the round's final library), with the
# 8-rank readiness line last and on one resident copy of the genome
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"

bash tools/r05_gpu_batch22.sh
o=gpurun_out/b22
{ echo "## python bench.py --gpus 8 --share-gpu --no-legs --no-cpu-baseline --steps 5 --warmup 1   (eight ranks on the ONE GPU: the 8-rank path runs, weak line + strong object; the genome resident once per rank)"; timeout 1500 python bench.py --gpus 8 --share-gpu --no-legs --no-cpu-baseline --steps 5 --warmup 1 2>$o/n8.err | grep -v Gloo; } >> $o/bench_lines.txt; tail -1 $o/bench_lines.txt | cut -c1-300; grep -i "error" $o/n8.err | grep -v "elastic\|error_file" | head -3
