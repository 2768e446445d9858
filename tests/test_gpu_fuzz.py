"""-m gpu: random option combinations (tests/fuzz_flags.py) through the HIP path (C-ABI, or the dwgsim-hip executable with "cli") and the
oracle, byte for byte.
Every case runs in its own process with a time limit; option sets that are skipped are counted by reason in the summary line ("oracle-rejects": the
reference's own option checks refuse them; "oracle-timeout": the oracle did not finish -- the generator no longer draws the Ion Torrent error rates at
which the unmodified reference does not terminate either, see tests/fuzz_flags.py -- ; "limit": a documented limit).  The source-derived seeds change
with dwgsim_amd/csrc and include/ only: a round that touches only tests/ or oracle/ re-runs the previous sample."""
import os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_seed():
    """A seed that changes with the product: the first digits of the sha256 over the kernel and host sources and the C-ABI header (the GPU box
    gets no .git, so `git rev-parse HEAD` is not available there; this is what a commit that touches the product changes).  Every round's GPU
    run therefore fuzzes a sample nobody has seen before -- and anyone can reproduce it from the same sources."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "dwgsim_amd", "csrc")
    for name in sorted(os.listdir(src)) + ["../../include/dwgsim_hip.h"]:
        p = os.path.join(src, name)
        if os.path.isfile(p) and name.endswith((".hip", ".hpp", ".cpp", ".h", "Makefile")):
            h.update(open(p, "rb").read())
    return 300000 + int(h.hexdigest()[:8], 16) % 600000


SRC_SEED = source_seed()


@pytest.mark.parametrize("seed,count,mode", [(101, 40, ""), (102, 25, "inputs"), (103, 20, "cli"), (104, 10, "inputs cli"), (105, 25, "shards"), (106, 15, "inputs shards")])
def test_random_option_sets_bit_exact(oracle_bin, seed, count, mode):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "fuzz_flags.py"), str(seed), str(count)] + mode.split()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), r.stdout[-2000:]


@pytest.mark.parametrize("seed,count,mode,mut", [(SRC_SEED, 30, "", 0), (SRC_SEED + 1, 15, "inputs", 0), (SRC_SEED + 2, 10, "cli", 0), (SRC_SEED + 3, 15, "shards", 0), (SRC_SEED + 4, 20, "", 1),
                                                 (SRC_SEED + 5, 10, "inputs shards", 1)],
                         ids=lambda v: str(v))
def test_fresh_random_option_sets_bit_exact(oracle_bin, seed, count, mode, mut):
    """The same fuzzer with seeds derived from the sources under test (the seed is in the test id and in the failure text): what the fixed seeds
    above found once they can only find again.  mut = 1: mutation rates of 0.05 .. 0.5 (the walk under stress, where round 3's campaign found the
    reach bug of the parallel left-justification)."""
    env = dict(os.environ)
    if mut:
        env["DWGSIM_FUZZ_MUT"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "tests", "fuzz_flags.py"), str(seed), str(count)] + mode.split()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), f"seed {seed} mode '{mode}' mut {mut}: " + r.stdout[-2000:]


@pytest.mark.parametrize("seed,count,mode", [(SRC_SEED + 7, 20, ""), (SRC_SEED + 8, 8, "shards"), (SRC_SEED + 9, 6, "cli")], ids=lambda v: str(v))
def test_long_reads_random_option_sets_bit_exact(oracle_bin, seed, count, mode):
    """The same fuzzer with read lengths of 640 .. 5 000 bases (DWGSIM_FUZZ_LONG): the one-wave blocks whose reads are staged in scratch slots, all
    read models and outputs, sharded and through the executable."""
    env = dict(os.environ, DWGSIM_FUZZ_LONG="1")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "fuzz_flags.py"), str(seed), str(count)] + mode.split()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), f"seed {seed} mode '{mode}': " + r.stdout[-2000:]


def test_ion_torrent_random_flow_orders_bit_exact(oracle_bin):
    """tests/fuzz_ion_flows.py: the flow model under flow orders of 4 .. 64 flows with long gaps, read lengths 1 .. 400, per-flow error rates up to
    0.2, -B; the buffers of a read that outgrows them are doubled up to 2^20 bases (rounds 3-4: up to 16 x the estimate, which 1 % of such samples met)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_ion_flows.py"), "31", "60"], capture_output=True, text=True, timeout=1200)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), r.stdout[-2000:]
    # a batch whose read outgrew its buffers runs again with twice the room: what may still fail is a read that degenerates (the run stack of pass 2,
    # 2^14 errors in one event) -- none in this sample; a regression of the capacity logic would show here
    assert int(last.split(" outgrew")[0].split()[-1]) == 0, last
