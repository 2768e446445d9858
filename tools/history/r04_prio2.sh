#!/bin/bash
# tools/r04_prio2.sh -- analysis only (gpurun): WHERE k_simulate drops its raised priority: after the error tests (before the look-backs are awaited), after the
# look-backs (the product), after the name line, after the base line.  Variants are sed-ed copies built on the box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_prio2; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
cd dwgsim_amd/csrc; mkdir -p build/knock
mk() { name=$1; mark=$2; sed "s|    if (SPLIT == 0) wave_priority(0);|    ;|; s|$mark|if (SPLIT == 0) wave_priority(0); $mark|" dw_simulate.hip > build/knock/dw_simulate_$name.hip; for P in 1 3; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -I. -O3 -std=c++17 -ffp-contract=off -fPIC -DDW_PART=$P -c build/knock/dw_simulate_$name.hip -o build/knock/${name}_$P.o & done; }
mk early "DW_PROBE_MARK(a, 2);     // error tests + substitutions"
mk name "DW_PROBE_MARK(a, 4); // header line"
mk bases "DW_PROBE_MARK(a, 5); // sequence line"
wait
for name in early name bases; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/walk.o build/gzip.o build/host.o build/mutin.o build/job.o build/s0.o build/knock/${name}_1.o build/s2.o build/knock/${name}_3.o build/s[4-9].o build/s10.o -lpthread -o ../libdwgsim_hip_knock_p$name.so; done
cd ../..
for v in product early name bases product; do
  lib=dwgsim_amd/libdwgsim_hip.so; [ $v != product ] && lib=dwgsim_amd/libdwgsim_hip_knock_p$v.so
  for fl in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -C 50 -e 0.01 -o 1"; do DWGSIM_HIP_LIB=$lib SPLIT=0 timeout 300 python tools/time_probe.py "$fl" 2>&1 | tail -1 | cut -c1-20,60-140 | sed "s/^/$v /"; done
done | tee $o/prio2.txt
