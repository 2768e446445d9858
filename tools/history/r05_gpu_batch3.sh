#!/bin/bash
# tools/r05_gpu_batch3.sh -- analysis only (gpurun): Ion Torrent bench lines of the three buffer homes + VALU / SALU per wave of each (one pmc pass)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "TACG or TCGA or ion_torrent" > gpurun_out/b3/pytest.log 2>&1; tail -2 gpurun_out/b3/pytest.log
for wl in chr20 ecoli; do
  for m in ${MODES:-1 2 0}; do
    echo "== $wl ion_lds=$m"
    DWGSIM_HIP_DEBUG="ion_lds=$m" timeout 600 python bench.py --workload $wl --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 2> gpurun_out/b3/bench_${wl}_$m.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d['roofline']['launch_ms'])"
  done
done
for m in ${MODES:-1 2 0}; do
  out=gpurun_out/b3/pmc_$m; rm -rf $out
  DWGSIM_HIP_DEBUG="ion_lds=$m" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out -- python bench.py --workload chr20 --ion --no-legs --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  echo "== pmc ion_lds=$m"; python tools/pmc_summary.py $(find $out -name '*.db') | grep k_simulate
  rm -rf $out
done
