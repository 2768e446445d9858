#!/bin/bash
# tools/r05_gpu_batch5.sh -- analysis only (gpurun): Ion Torrent with the buffers in scratch slots: fewer slots per XCD (do they stay in the 4 MB L2?) -- speed and traffic
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b5
for sl in 96 112 128 160; do
  echo "== flow_slots=$sl"
  DWGSIM_HIP_DEBUG="ion_lds=0,flow_slots=$sl" timeout 600 python bench.py --workload chr20 --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['roofline']['launch_ms'])"
  for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
    out=gpurun_out/b5/pmc; rm -rf $out
    DWGSIM_HIP_DEBUG="ion_lds=0,flow_slots=$sl" rocprofv3 --kernel-trace --pmc $pmc -d $out -- python bench.py --workload chr20 --ion --no-legs --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
    python tools/pmc_summary.py $(find $out -name '*.db') | grep k_simulate
    rm -rf $out
  done
done
