#!/bin/bash
# tools/r05_gpu_batch4.sh -- analysis only (gpurun): Ion Torrent after the capacity change + non-temporal text stores: bench lines and HBM traffic of both buffer homes
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/b4
for wl in chr20 ecoli; do
  for m in ${MODES:-1 0}; do
    echo "== $wl ion_lds=$m"
    DWGSIM_HIP_DEBUG="ion_lds=$m" timeout 600 python bench.py --workload $wl --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 2> gpurun_out/b4/bench_${wl}_$m.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d['roofline']['launch_ms'])"
  done
done
for m in ${MODES:-1 0}; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    out=gpurun_out/b4/pmc_$m; rm -rf $out
    DWGSIM_HIP_DEBUG="ion_lds=$m" rocprofv3 --kernel-trace --pmc $pmc -d $out -- python bench.py --workload chr20 --ion --no-legs --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
    echo "== pmc ion_lds=$m"; python tools/pmc_summary.py $(find $out -name '*.db') | grep k_simulate
    rm -rf $out
  done
done
python bench.py --no-legs --no-cpu-baseline --steps 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
