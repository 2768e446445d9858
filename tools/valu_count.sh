#!/bin/bash
# tools/valu_count.sh "<dwgsim flags>" -- analysis only: VALU / SALU instructions per wave and kernel time of k_simulate for a flag set
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/valu_$$; rm -rf $out
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU -d $out -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --flags "$1" > /dev/null 2>&1
python - "$out" "$1" <<'PY'
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
rows = dict((r[0], (r[1], r[2])) for r in db.execute("select counter_name, avg(value), avg(duration) from counters_collection where kernel_name like '%k_simulate%' group by counter_name"))
w = rows["SQ_WAVES"][0]
print(f"{sys.argv[2]:60s} VALU/wave {rows['SQ_INSTS_VALU'][0]/w:8.0f}  SALU/wave {rows['SQ_INSTS_SALU'][0]/w:7.0f}  kernel {rows['SQ_WAVES'][1]/1e3:7.1f} us")
PY
