#!/bin/bash
# tools/r04_final_profiles.sh -- analysis only (run through gpurun): the rocprofv3 passes behind profiles/r04_*_kernel_stats_pmc.txt and
# profiles/r04_counters.json (keyed by the library's sha256) for the current build: chr20 (the bench workload), E. coli, Ion Torrent (both sizes),
# -o 0, SOLiD 2 x 50, 2 000-base reads.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
mkdir -p gpurun_out/final
one() {
  name=$1; wl=$2; extra=$3; key=$4
  timeout 900 bash tools/profile_round.sh final_$name $wl "$extra --no-genome-leg" > gpurun_out/final/$name.log 2>&1
  d=gpurun_out/final_$name
  echo "${GIT_HEAD:-unknown}" > $d/git_head.txt
  {
    echo "# profiles/r04_${name}_kernel_stats_pmc.txt -- rocprofv3 passes of tools/profile_round.sh (MI355X; tools/r04_final_profiles.sh); library sha256 $(sha256sum dwgsim_amd/libdwgsim_hip.so | cut -c1-16)..."
    echo "## bench line of the same build"
    cat $d/bench_line.json
    echo
    echo "## rocprofv3 --kernel-trace --stats (10 timed steps + 2 warm-up)"
    head -24 $d/kernel_stats.txt
    echo "## rocprofv3 --kernel-trace --pmc <one group per pass> (4 timed steps + 1 warm-up); FETCH_SIZE / WRITE_SIZE in KiB"
    cat $d/pmc.txt
  } > gpurun_out/final/r04_${name}_kernel_stats_pmc.txt
  [ -n "$key" ] && python tools/make_counters_json.py $key $d gpurun_out/final/r04_counters.json > /dev/null
}
one chr20 chr20 "" chr20
one ecoli ecoli "" ecoli
one ion ecoli "--ion" ecoli_ion
one ion_chr20 chr20 "--ion" chr20_ion
one o0 chr20 "--flags='-z 13 -1 150 -2 150 -C 30 -o 0'" ""
one solid50 chr20 "--flags='-z 13 -c 1 -1 50 -2 50 -C 30 -o 0'" ""
one long2000 chr20 "--flags='-z 13 -1 2000 -2 0 -C 30 -o 1'" ""
cat gpurun_out/final/r04_counters.json
