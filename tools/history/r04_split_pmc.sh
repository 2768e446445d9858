#!/bin/bash
# analysis only (gpurun): counters of the two halves (and of the single kernel) on the chr20-sized launch
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_split_pmc; rm -rf $o; mkdir -p $o
for sp in ${SPLITS:-1 0}; do
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  SPLIT=$sp rocprofv3 --kernel-trace --pmc $pmc -d $o/s${sp}_pmc$i -- python tools/time_probe.py "${FLAGS:--z 13 -1 150 -2 150 -C 30 -o 1}" > $o/s${sp}_pmc$i.log 2>&1
  python tools/pmc_summary.py $(find $o/s${sp}_pmc$i -name '*.db') 2>&1 | grep "k_simulate"
done
done
