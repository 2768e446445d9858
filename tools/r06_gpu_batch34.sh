#!/bin/bash
# tools/r06_gpu_batch34.sh -- (gpurun) the round's final library, last pass again after the episode change: final profiles + counters, the bench line with them installed, the fuzz campaign (new seeds), the -m gpu suite
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06b34; mkdir -p $o gpurun_out/final gpurun_out/r06_fuzz
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
bash tools/r06_final_profiles.sh > $o/final.log 2>&1
cp gpurun_out/final/r06_counters.json profiles/r06_counters.json
python bench.py > gpurun_out/final/bench_line_n1.json 2> gpurun_out/final/bench_line_n1.err; python -c "import json; d=json.loads(open('gpurun_out/final/bench_line_n1.json').read().strip().splitlines()[-1]); print('bench line', d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['strong']['value'], d['cpu_baseline']['value'])"
sed -i "s/952\([0-9]\)/954\1/g; s/962\([0-9]\)/964\1/g; s/972\([0-9]\)/974\1/g" tools/r06_fuzz_campaign.sh
bash tools/r06_fuzz_campaign.sh > $o/fuzz_campaign.log 2>&1; cp gpurun_out/r06_fuzz/fuzz.txt $o/fuzz.txt; cat $o/fuzz.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $o/gputest.txt 2>&1; tail -3 $o/gputest.txt
