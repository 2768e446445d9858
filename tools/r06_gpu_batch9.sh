#!/bin/bash
# tools/r06_gpu_batch9.sh -- analysis only (gpurun): VALU / SALU instructions per wave of the 2 x 150 launch with parts switched off (flags, knock-out builds): the instruction budget by phase
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b9; mkdir -p $o; : > $o/valu_by_phase.txt
B="-z 13 -1 150 -2 150 -C 30 -o 1"
one() { name=$1; lib=$2; shift 2
  DWGSIM_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $o/pmc -- python bench.py --no-legs --no-cpu-baseline --no-pipeline --steps 3 --warmup 1 "$@" > $o/log.txt 2>&1
  python - "$name" $(find $o/pmc -name '*.db') >> $o/valu_by_phase.txt <<'PY'
import sqlite3, sys
name=sys.argv[1]; tot={}
for f in sys.argv[2:]:
    db=sqlite3.connect(f)
    for kn, cn, v, n, dur in db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if "k_simulate" in kn: tot[cn]=v; tot["dur_us"]=dur/1e3
w=tot.get("SQ_WAVES",1)
print(f"{name:34s} VALU/wave {tot.get('SQ_INSTS_VALU',0)/w:8.0f}  SALU/wave {tot.get('SQ_INSTS_SALU',0)/w:8.0f}  LDS/wave {tot.get('SQ_INSTS_LDS',0)/w:7.0f}  VMEM wr/wave {tot.get('SQ_INSTS_VMEM_WR',0)/w:6.0f}  rd/wave {tot.get('SQ_INSTS_VMEM_RD',0)/w:6.0f}  kernel {tot.get('dur_us',0):8.1f} us")
PY
  rm -rf $o/pmc; }
P=dwgsim_amd/libdwgsim_hip.so
one "default" $P
one "-Q 0 (no quality normals)" $P "--flags=$B -Q 0"
one "-q I (constant quality line)" $P "--flags=$B -q I"
one "-e 0 -E 0 (no errors)" $P "--flags=$B -e 0 -E 0"
one "-r 0 (no mutations)" $P "--flags=$B -r 0"
one "-y 0 (no random reads)" $P "--flags=$B -y 0"
one "knock 1 (no text assembly)" dwgsim_amd/libdwgsim_hip_var_knock1.so
one "knock 1, -Q 0" dwgsim_amd/libdwgsim_hip_var_knock1.so "--flags=$B -Q 0"
one "knock 16 (no header)" dwgsim_amd/libdwgsim_hip_var_knock16.so
one "knock 8 (no extraction)" dwgsim_amd/libdwgsim_hip_var_knock8.so
one "2 x 50" $P "--flags=-z 13 -1 50 -2 50 -C 30 -o 1"
one "2 x 250" $P "--flags=-z 13 -1 250 -2 250 -C 30 -o 1"
cat $o/valu_by_phase.txt
