"""Replay parity on the CPU (tests/replay_common.py): the unmodified reference, its drand48() replaying the Philox stream of mode B, must
write the files the oracle writes in mode B, from exactly the draws that were dumped.  Retires "mode A == reference and mode B == same code,
other provider" as the only link between the reference and the stream the HIP kernels draw from, for every configuration without a normal."""
import os, random
import pytest

from replay_common import REPLAY_CASES, SUFFIXES, have_reference, run_replay, random_replay_flags

pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref/dwgsim (the unmodified reference, built where /root/reference is mounted) is not here")


def check(oracle_bin, golden_dir, fasta, flags, tmp):
    orc, rrc, want, got, served, avail = run_replay(oracle_bin, os.path.join(golden_dir, fasta), flags, str(tmp))
    if orc != 0 or rrc != 0:      # an option set both reject (the reference's own parse errors), or on which both abort the same way
        assert orc == rrc or (orc != 0 and rrc != 0), f"oracle rc {orc}, replayed reference rc {rrc}: {flags}"
        return False
    for suf in SUFFIXES:
        assert got[suf] == want[suf], f"{suf} differs between the replayed reference ({len(got[suf])} bytes) and mode B ({len(want[suf])}): {fasta} {flags}"
    assert served == avail and avail > 0, f"the reference consumed {served} of the {avail} dumped draws: {fasta} {flags}"
    return True


@pytest.mark.parametrize("fasta,flags", REPLAY_CASES, ids=[f"{f}:{fl}" for f, fl in REPLAY_CASES])
def test_reference_fed_the_philox_stream_writes_mode_b(oracle_bin, golden_dir, tmp_path, fasta, flags):
    assert check(oracle_bin, golden_dir, fasta, flags, tmp_path)


@pytest.mark.parametrize("seed", [20260929, 31337])
def test_replay_fuzz(oracle_bin, golden_dir, tmp_path, seed):
    """25 random single-end option sets per seed (Illumina / SOLiD / Ion Torrent, mutation inputs, regions)."""
    rng = random.Random(seed); ran = 0
    for k in range(25):
        fasta, flags = random_replay_flags(rng)
        d = tmp_path / f"c{k}"; d.mkdir()
        ran += bool(check(oracle_bin, golden_dir, fasta, flags, d))
    assert ran >= 15
