#!/bin/bash
# tools/r06_gpu_batch30.sh -- (gpurun) readiness of the N-rank paths on the one GPU with the final library: the driver's own launch line for N = 2 and N = 8 (ranks share the device: not a measurement),
# the dwgsim-hip executable end to end, smoke()
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b30; mkdir -p $o
for n in 2 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) bench.py --gpus $n --steps 5 --warmup 1 --share-gpu > $o/n$n.json 2> $o/n$n.err
  echo "N=$n rc=$?"; tail -1 $o/n$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('strong',{}).get('value'))"
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
