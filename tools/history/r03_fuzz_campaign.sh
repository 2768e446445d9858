#!/bin/bash
# tools/r03_fuzz_campaign.sh [seed0] [count] [dense] -- analysis only (gpurun): a long run of tests/fuzz_flags.py on the GPU (every mode), results under
# gpurun_out/; "dense": every option set with a mutation rate of 0.05 - 0.5 (DWGSIM_FUZZ_MUT)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
make -s -C oracle oracle > /dev/null 2>&1
s=${1:-9300}; n=${2:-400}
[ "$3" = dense ] && export DWGSIM_FUZZ_MUT=1
i=0
for mode in "" "inputs" "cli" "inputs cli" "shards" "inputs shards"; do
  i=$((i + 1))
  timeout 1500 python tests/fuzz_flags.py $((s + i)) $n $mode > gpurun_out/fuzz_r03_$((s + i)).txt 2>&1
  tail -1 gpurun_out/fuzz_r03_$((s + i)).txt
done
