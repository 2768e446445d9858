#!/bin/bash
# tools/r02_final_profiles.sh -- analysis only (run through gpurun): the rocprofv3 passes behind profiles/r02_*_kernel_stats_pmc.txt and
# profiles/r02_counters.json for this round's final build: chr20 (the bench workload), E. coli, Ion Torrent; plus the knock-out timings.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/final
for spec in "chr20:chr20:" "ecoli:ecoli:" "ion:ecoli:--ion"; do
  IFS=: read -r name wl extra <<< "$spec"
  timeout 500 bash tools/profile_round.sh final_$name $wl "$extra" > gpurun_out/final/$name.log 2>&1
  d=gpurun_out/final_$name
  {
    echo "# profiles/r02_${name}_kernel_stats_pmc.txt -- rocprofv3 passes of tools/profile_round.sh (MI355X, this round's final build; tools/r02_final_profiles.sh)"
    echo "## bench line of the same build"
    cat $d/bench_line.json
    echo
    echo "## rocprofv3 --kernel-trace --stats (10 timed steps + 2 warm-up)"
    head -22 $d/kernel_stats.txt
    echo "## rocprofv3 --kernel-trace --pmc <one group per pass> (4 timed steps + 1 warm-up); FETCH_SIZE / WRITE_SIZE in KiB"
    cat $d/pmc.txt
  } > gpurun_out/final/r02_${name}_kernel_stats_pmc.txt
  key=$name; [ $name = ion ] && key=ecoli_ion
  python tools/make_counters_json.py $key $d gpurun_out/final/r02_counters.json > /dev/null
done
cat gpurun_out/final/r02_counters.json
echo "== knock-outs by flags (chr20) =="
for f in "-z 13 -1 150 -2 150 -C 30 -o 1" "-z 13 -1 150 -2 150 -C 30 -o 1 -Q 0" "-z 13 -1 150 -2 150 -C 30 -o 1 -q I" "-z 13 -1 150 -2 150 -C 30 -o 1 -y 0" "-z 13 -1 150 -2 150 -C 30 -o 1 -e 0 -E 0" "-z 13 -1 150 -2 150 -C 30 -o 0" "-z 13 -1 150 -2 150 -C 30 -o 2"; do
  for w in 1 0; do WRITER=$w timeout 100 python tools/time_probe.py "$f" 2>/dev/null; done
done | tee gpurun_out/final/knock_flags.txt
