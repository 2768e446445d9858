"""-m gpu: random option combinations (tests/fuzz_flags.py) through the HIP path (C-ABI, or the dwgsim-hip executable with "cli") and the
oracle, byte for byte.
Every case runs in its own process with a time limit; option sets the oracle itself rejects or gives up on are skipped."""
import os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,count,mode", [(101, 40, ""), (102, 25, "inputs"), (103, 20, "cli"), (104, 10, "inputs cli"), (105, 25, "shards"), (106, 15, "inputs shards")])
def test_random_option_sets_bit_exact(oracle_bin, seed, count, mode):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "fuzz_flags.py"), str(seed), str(count)] + mode.split()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), r.stdout[-2000:]


def test_ion_torrent_random_flow_orders_bit_exact(oracle_bin):
    """tests/fuzz_ion_flows.py: the flow model under flow orders of 4 .. 64 flows with long gaps, read lengths 1 .. 400, per-flow error rates up to
    0.2, -B; reads that outgrow their buffer (the documented limit) are counted, not failed."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_ion_flows.py"), "31", "100"], capture_output=True, text=True, timeout=1200)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), r.stdout[-2000:]
