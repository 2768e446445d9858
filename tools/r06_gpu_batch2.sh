#!/bin/bash
# tools/r06_gpu_batch2.sh -- analysis only (gpurun): round 6's second pass.  The flow model with gap-drawn first draws and event rounds, the site scan in
# slots, the count stream, the offset sink: (1) the -m gpu suite, (2) Ion Torrent lines + instruction counters, (3) the walk of the genome kernel by kernel,
# (4) the solo-rank sweep again, (5) the delivery threads with the offset sink
cd /tmp && export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
o=gpurun_out/r06b2; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/library_sha256.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['breakdown_ms']['simulate_kernels'], d['breakdown_ms']['walk_gpu'], d['roofline']['frac'])"; }
{
for wl in chr20 ecoli chr20_like; do
  python bench.py --workload $wl --steps 30 --no-legs --no-cpu-baseline --ion 2>/dev/null | line "$wl,ion400"
done
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline "--flags=-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.000000001 -o 1" 2>/dev/null | line "chr20,ion400,e=1e-9(no events)"
python bench.py --workload chr20 --steps 30 --no-legs --no-cpu-baseline "--flags=-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 200 -2 0 -C 50 -e 0.01 -o 1" 2>/dev/null | line "chr20,ion200"
python bench.py --workload chr20 --steps 50 --no-legs --no-cpu-baseline 2>/dev/null | line "chr20,2x150"
python bench.py --workload ecoli --steps 50 --no-legs --no-cpu-baseline 2>/dev/null | line "ecoli,2x150"
python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432 --no-legs --no-cpu-baseline 2>/dev/null | line "genome-two-groups,no-pipeline(walk alone)"
python bench.py --workload chr20 --steps 10 --no-legs --no-cpu-baseline --no-pipeline 2>/dev/null | line "chr20,2x150,no-pipeline(walk alone)"
} | tee $o/lines.txt
# Ion Torrent: kernel stats + instruction counters
for wl in chr20; do
  rocprofv3 --kernel-trace --stats -d $o/kt -- python bench.py --workload $wl --ion --no-legs --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
  python tools/rocprof_summary.py "$(find $o/kt -name '*.db' | head -1)" > $o/ion_${wl}_kernel_stats.txt 2>&1; rm -rf $o/kt
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES -d $o/pmc -- python bench.py --workload $wl --ion --no-legs --no-cpu-baseline --steps 4 --warmup 1 > $o/pmc.log 2>&1
  python tools/pmc_summary.py $(find $o/pmc -name '*.db') > $o/ion_${wl}_pmc.txt 2>&1; rm -rf $o/pmc
  head -8 $o/ion_${wl}_kernel_stats.txt; cat $o/ion_${wl}_pmc.txt
done
# the walk of the genome, kernel by kernel
rocprofv3 --kernel-trace --stats -d $o/kt -- python bench.py --workload grch38 --mode strong --steps 3 --warmup 1 --no-pipeline --group-bp 2130706432 --no-legs --no-cpu-baseline > /dev/null 2>&1
python tools/rocprof_summary.py "$(find $o/kt -name '*.db' | head -1)" > $o/genome_walk_kernel_stats.txt 2>&1; rm -rf $o/kt; head -40 $o/genome_walk_kernel_stats.txt
# solo-rank sweep
timeout 900 python bench.py --solo-sweep 2,4,8 --no-cpu-baseline --no-legs > $o/solo_sweep.json 2> $o/solo_sweep.err; python - <<PY
import json; d=json.load(open("$o/solo_sweep.json"))
for mode in ("weak","strong"):
    for W,v in d[mode].items():
        if W=="job": continue
        print(mode, "W", W, "max", v["max_ms_per_step"], "min", v["min_ms_per_step"], "eff", v["efficiency"])
PY
# the job level with the offset sink
python - <<PY
import sys, time; sys.path.insert(0, "$R")
from dwgsim_amd import synth
c = synth.workload_contigs("grch38"); synth.write_fasta("/dev/shm/g38.fa", c)
with open("/dev/shm/g38.fa.fai", "w") as f:
    off = 0
    for name, arr in c:
        off += len(name) + 2; f.write(f"{name}\t{len(arr)}\t{off}\t60\t61\n"); off += len(arr) + (len(arr) + 59) // 60
PY
FL="-z 13 -1 150 -2 150 -C 30 -o 1"
{
echo "== null sink, offset sink (default), 1 context"; DWGSIM_HIP_SINK=null DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
echo "== memcpy sink, ORDERED (one delivery thread per stream), 1 context"; DWGSIM_HIP_SINK_ORDERED=1 DWGSIM_HIP_SINK=memcpy DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
echo "== memcpy sink, offset sink, 1 context"; DWGSIM_HIP_SINK=memcpy DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
echo "== memcpy sink, ORDERED, 8 contexts on this GPU"; DWGSIM_HIP_SINK_ORDERED=1 DWGSIM_HIP_SINK=memcpy DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=0,0,0,0,0,0,0,0 timeout 300 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
echo "== memcpy sink, offset sink, 8 contexts on this GPU"; DWGSIM_HIP_SINK=memcpy DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=0,0,0,0,0,0,0,0 timeout 300 dwgsim_amd/dwgsim-hip $FL /dev/shm/g38.fa /dev/shm/out 2>&1 | grep "^\[dwgsim-hip"
echo "== files on tmpfs (pwrite), chr20-sized job, 1 context, and the same through the ordered sink"; python - <<PY
import sys; sys.path.insert(0, "$R")
from dwgsim_amd import synth
synth.write_fasta("/dev/shm/c20.fa", synth.workload_contigs("chr20"))
PY
DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/c20.fa /dev/shm/o_at 2>&1 | grep "^\[dwgsim-hip\]"
DWGSIM_HIP_SINK_ORDERED=1 DWGSIM_HIP_TIMING=1 DWGSIM_HIP_DEVICES=1 dwgsim_amd/dwgsim-hip $FL /dev/shm/c20.fa /dev/shm/o_ord 2>&1 | grep "^\[dwgsim-hip\]"
for f in bwa.read1.fastq.gz bwa.read2.fastq.gz mutations.txt mutations.vcf; do cmp /dev/shm/o_at.$f /dev/shm/o_ord.$f && echo "same $f"; done
} | tee $o/job_level_sinks.txt
rm -f /dev/shm/g38.fa /dev/shm/g38.fa.fai /dev/shm/out.* /dev/shm/c20.fa /dev/shm/o_at.* /dev/shm/o_ord.*
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; tail -3 $o/pytest_gpu.log; fi
