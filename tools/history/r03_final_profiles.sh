#!/bin/bash
# tools/r03_final_profiles.sh -- analysis only (run through gpurun): the rocprofv3 passes behind profiles/r03_*_kernel_stats_pmc.txt and
# profiles/r03_counters.json (keyed by the library's sha256) for the current build: chr20 (the bench workload), E. coli, Ion Torrent.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
mkdir -p gpurun_out/final
for spec in "chr20:chr20:" "ecoli:ecoli:" "ion:ecoli:--ion"; do
  IFS=: read -r name wl extra <<< "$spec"
  timeout 700 bash tools/profile_round.sh final_$name $wl "$extra --no-genome-leg" > gpurun_out/final/$name.log 2>&1
  d=gpurun_out/final_$name
  echo "${GIT_HEAD:-unknown}" > $d/git_head.txt
  {
    echo "# profiles/r03_${name}_kernel_stats_pmc.txt -- rocprofv3 passes of tools/profile_round.sh (MI355X; tools/r03_final_profiles.sh); library sha256 $(sha256sum dwgsim_amd/libdwgsim_hip.so | cut -c1-16)..."
    echo "## bench line of the same build"
    cat $d/bench_line.json
    echo
    echo "## rocprofv3 --kernel-trace --stats (10 timed steps + 2 warm-up)"
    head -24 $d/kernel_stats.txt
    echo "## rocprofv3 --kernel-trace --pmc <one group per pass> (4 timed steps + 1 warm-up); FETCH_SIZE / WRITE_SIZE in KiB"
    cat $d/pmc.txt
  } > gpurun_out/final/r03_${name}_kernel_stats_pmc.txt
  key=$name; [ $name = ion ] && key=ecoli_ion
  python tools/make_counters_json.py $key $d gpurun_out/final/r03_counters.json > /dev/null
done
cat gpurun_out/final/r03_counters.json
