#!/bin/bash
# tools/r04_fuzz_campaign.sh -- analysis only (gpurun): random option sets against the oracle beyond what the suite holds -- long reads (scratch slots),
# Ion Torrent flow orders (capacity re-runs), plain / inputs / shards / cli with high mutation rates
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04_fuzz; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
{
for sd in 9101 9102 9103; do DWGSIM_FUZZ_LONG=1 timeout 900 python tests/fuzz_flags.py $sd 60 | tail -3; done
DWGSIM_FUZZ_LONG=1 timeout 900 python tests/fuzz_flags.py 9104 40 shards | tail -3
DWGSIM_FUZZ_LONG=1 timeout 900 python tests/fuzz_flags.py 9105 30 cli | tail -3
for sd in 9201 9202; do timeout 1200 python tests/fuzz_ion_flows.py $sd 150 | tail -3; done
timeout 900 python tests/fuzz_flags.py 9301 150 | tail -3
DWGSIM_FUZZ_MUT=1 timeout 900 python tests/fuzz_flags.py 9302 100 | tail -3
timeout 900 python tests/fuzz_flags.py 9303 60 cli | tail -3
timeout 900 python tests/fuzz_flags.py 9304 60 inputs shards | tail -3
} 2>&1 | tee $o/fuzz.txt
