#!/bin/bash
# tools/r06_fuzz_campaign.sh -- analysis only (gpurun), round 6, new seeds, the final library: random option sets against the oracle beyond what the suite holds -- long reads (scratch slots),
# Ion Torrent flow orders (capacity re-runs), plain / inputs / shards / cli with high mutation rates
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r06_fuzz; mkdir -p $o
sha256sum dwgsim_amd/libdwgsim_hip.so | tee $o/library_sha256.txt
{
for sd in 9521 9522 9523; do DWGSIM_FUZZ_LONG=1 timeout 900 python tests/fuzz_flags.py $sd 60 | tail -3; done
DWGSIM_FUZZ_LONG=1 timeout 900 python tests/fuzz_flags.py 9524 40 shards | tail -3
DWGSIM_FUZZ_LONG=1 timeout 900 python tests/fuzz_flags.py 9525 30 cli | tail -3
for sd in 9621 9622; do timeout 1200 python tests/fuzz_ion_flows.py $sd 150 | tail -3; done
timeout 900 python tests/fuzz_flags.py 9721 150 | tail -3
DWGSIM_FUZZ_MUT=1 timeout 900 python tests/fuzz_flags.py 9722 100 | tail -3
timeout 900 python tests/fuzz_flags.py 9723 60 cli | tail -3
timeout 900 python tests/fuzz_flags.py 9724 60 inputs shards | tail -3
} 2>&1 | tee $o/fuzz.txt
