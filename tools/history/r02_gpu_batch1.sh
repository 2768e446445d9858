#!/bin/bash
# analysis only: the round-2 measurement batch (run through gpurun)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MASTER_ADDR=127.0.0.1
echo "== 2 ranks on one GPU, weak (ecoli) =="
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 2 --share-gpu --workload ecoli 2>&1 | tail -1
echo "== 2 ranks on one GPU, strong (grch38_mini) =="
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 1 --share-gpu --workload grch38_mini --mode strong 2>&1 | tail -1
echo "== 1 rank grch38_mini =="
timeout 300 python bench.py --steps 5 --warmup 1 --workload grch38_mini --no-legs --no-cpu-baseline 2>&1 | tail -1
echo "== ion (ecoli contig) =="
timeout 300 python bench.py --steps 20 --warmup 2 --workload ecoli --ion --no-legs --no-cpu-baseline 2>&1 | tail -1
echo "== ion (chr20 contig) =="
timeout 300 python bench.py --steps 5 --warmup 1 --workload chr20 --ion --no-legs --no-cpu-baseline 2>&1 | tail -1
