#!/bin/bash
# tools/r05_gpu_batch6.sh -- analysis only (gpurun): Ion Torrent, buffers in LDS, as one kernel (split=0) and as two (default): parity subset, bench lines, traffic
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b6
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "TACG or TCGA or ion_torrent" > gpurun_out/b6/pytest.log 2>&1; tail -2 gpurun_out/b6/pytest.log
for wl in chr20 ecoli; do
  for dbg in "ion_lds=1" "ion_lds=1,split=0" "ion_lds=0"; do
    echo "== $wl $dbg"
    DWGSIM_HIP_DEBUG="$dbg" timeout 600 python bench.py --workload $wl --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 2> gpurun_out/b6/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'])"
  done
done
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  out=gpurun_out/b6/pmc; rm -rf $out
  DWGSIM_HIP_DEBUG="ion_lds=1" rocprofv3 --kernel-trace --pmc $pmc -d $out -- python bench.py --workload chr20 --ion --no-legs --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/pmc_summary.py $(find $out -name '*.db') | grep -E "k_simulate|k_split"
  rm -rf $out
done
