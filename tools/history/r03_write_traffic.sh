#!/bin/bash
# tools/r03_write_traffic.sh -- analysis only (gpurun): WRITE_SIZE of the chr20-sized k_simulate launch for both record writers, with and
# without the quality line / header, 2 x 50 bp; FASTQ bytes of the launch are printed by time_probe (GB/s x ms)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
o=gpurun_out/wtraffic; mkdir -p $o
i=0
for spec in "1|-z 13 -1 150 -2 150 -C 30 -o 1" "0|-z 13 -1 150 -2 150 -C 30 -o 1" "1|-z 13 -1 150 -2 150 -C 30 -o 1 -Q 0" "1|-z 13 -1 150 -2 150 -C 30 -o 1 -q I" "1|-z 13 -1 50 -2 50 -C 10 -o 1" "1|-z 13 -1 150 -2 150 -C 30 -o 2" "1|-z 13 -1 150 -2 0 -C 15 -o 1"; do
  i=$((i + 1)); w=${spec%%|*}; f=${spec#*|}
  echo "== writer $w  $f"
  WRITER=$w python tools/time_probe.py "$f" 2>&1 | tail -1
  WRITER=$w rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $o/p$i -- python tools/time_probe.py "$f" > $o/p$i.log 2>&1
  python tools/pmc_summary.py $(find $o/p$i -name '*.db') 2>&1 | grep simulate
done
