#!/bin/bash
# tools/r04_gpu_batch7.sh -- analysis only (gpurun): scratch slots (Ion Torrent buffers, long reads of the one-wave blocks), a batch that runs again when
# an Ion Torrent read outgrows its buffers; read-length sweep: 256-lane blocks staging in LDS against one-wave blocks staging in scratch slots.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_b7; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
sha256sum dwgsim_amd/libdwgsim_hip.so > $o/lib.sha256
timeout 1500 python -m pytest tests -x -q -m gpu -k "ion or flow or scratch or staging_limit or outgrows or kernel_parity" > $o/pytest.log 2>&1; tail -5 $o/pytest.log
{
for L in 300 400 600 800 1000 1150; do
  for simt in "" 64; do SIMT=$simt timeout 300 python tools/time_probe.py "-z 13 -1 $L -2 0 -C 30 -o 1" 2>&1 | tail -1 | sed "s/^/SIMT=${simt:-256} /"; done
done
for L in 2000 4000 10000 30000; do timeout 300 python tools/time_probe.py "-z 13 -1 $L -2 0 -C 30 -o 1" 2>&1 | tail -1 | sed "s/^/SIMT=auto /"; done
SIMT=64 timeout 300 python tools/time_probe.py "-z 13 -1 150 -2 150 -C 30 -o 1" 2>&1 | tail -1 | sed "s/^/SIMT=64 /"
timeout 300 python tools/time_probe.py "-z 13 -c 1 -1 2000 -2 0 -C 30 -o 0" 2>&1 | tail -1 | sed "s/^/SIMT=auto /"
} > $o/length_sweep.txt 2>&1
cat $o/length_sweep.txt
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > $o/$name.json 2> $o/$name.err; tail -c 300 $o/$name.err | grep -v "amdgpu.ids\|socket.cpp" | tail -3; }
run ion_chr20 --ion --steps 10
run ion_ecoli --ion --workload ecoli --steps 20
run n1 --steps 50
for f in $o/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b=d["breakdown_ms"]; print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, {k:b[k] for k in b if k!="note"}, d["roofline"]["frac"])
except Exception as e: print("ERR",e)
PY
done
timeout 900 bash tools/profile_round.sh r04_b7/long2000 chr20 "--flags='-z 13 -1 2000 -2 0 -C 30 -o 1'" > $o/profile_long2000.log 2>&1; tail -45 $o/profile_long2000.log | cut -c1-200
