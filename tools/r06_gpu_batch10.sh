#!/bin/bash
# tools/r06_gpu_batch10.sh -- (gpurun) the substitution draws merged into the error-site chain: the -m gpu suite and the bench lines on that library
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b10; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha
for i in 1 2; do python bench.py --steps 100 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 >> $o/lines.txt; done
python bench.py --steps 100 --no-legs --no-cpu-baseline --workload ecoli_like 2>/dev/null | tail -1 >> $o/lines.txt
python bench.py --steps 60 --no-legs --no-cpu-baseline --ion 2>/dev/null | tail -1 >> $o/lines.txt
python bench.py --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -c 2 -1 200 -2 200 -C 30 -o 1" 2>/dev/null | tail -1 >> $o/lines.txt
python bench.py --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -c 1 -1 50 -2 50 -C 30 -o 1" 2>/dev/null | tail -1 >> $o/lines.txt
python bench.py --steps 60 --no-legs --no-cpu-baseline "--flags=-z 13 -1 150 -2 150 -C 30 -o 0" 2>/dev/null | tail -1 >> $o/lines.txt
cat $o/lines.txt | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print('??', l[:200]); continue
    print(d['config'].get('workload'), d['config'].get('flags',''), d['value'], d['unit'], d['roofline']['frac'])
"
timeout 2400 python -m pytest tests -x -q -m gpu > $o/gputest.txt 2>&1; tail -3 $o/gputest.txt
