#!/bin/bash
# tools/r06_gpu_batch13.sh -- (gpurun) analysis: what a VALU instruction costs on this device and how the SQ's VALU counters relate to it (tools/probe/valu_rate.hip)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b13; mkdir -p $o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe/valu_rate.hip -o /tmp/valu_rate 2>/dev/null
/tmp/valu_rate | tee $o/valu_rate.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $o/pmc -- /tmp/valu_rate > /dev/null 2>&1
python - $(find $o/pmc -name '*.db') <<'PY' | tee $o/valu_rate_pmc.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
t = "counters_collection"
rows = list(db.execute(f"select dispatch_id, kernel_name, counter_name, value, duration from {t} order by dispatch_id"))
by = {}
for d, k, c, v, dur in rows:
    e = by.setdefault(d, {"k": k, "dur": dur}); e[c] = e.get(c, 0) + v
for d in sorted(by):
    e = by[d]
    w = e.get("SQ_WAVES", 1)
    print(f"{e['k'][:28]:28s} waves {w:6.0f} dur {e['dur']/1e3:8.1f} us  INSTS_VALU/wave {e.get('SQ_INSTS_VALU',0)/w:9.0f}  ACTIVE_INST_VALU/INSTS_VALU {e.get('SQ_ACTIVE_INST_VALU',0)/max(1,e.get('SQ_INSTS_VALU',1)):5.2f}  WAVE_CYCLES/wave {e.get('SQ_WAVE_CYCLES',0)/w:9.0f}  BUSY_CYCLES {e.get('SQ_BUSY_CYCLES',0):12.0f}  GUI_ACTIVE {e.get('GRBM_GUI_ACTIVE',0):10.0f}")
PY
rm -rf $o/pmc
