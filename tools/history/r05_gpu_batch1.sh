#!/bin/bash
# tools/r05_gpu_batch1.sh -- analysis only (gpurun): first look at the round-5 Ion Torrent kernels (read buffers in LDS, in place, 2 bits per base):
# parity subset, then bench lines of the three buffer homes on both launch sizes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "TACG or TCGA or ion_torrent" > gpurun_out/b1/pytest.log 2>&1; tail -3 gpurun_out/b1/pytest.log
for wl in chr20 ecoli; do
  for m in 1 2 0; do
    echo "== $wl ion_lds=$m"
    DWGSIM_HIP_DEBUG="ion_lds=$m" timeout 600 python bench.py --workload $wl --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 2> gpurun_out/b1/bench_${wl}_$m.err | tee gpurun_out/b1/bench_${wl}_$m.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d['roofline']['launch_ms'])"
  done
done
python bench.py --no-legs --no-cpu-baseline --steps 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
