"""-m gpu: the HIP path (libdwgsim_hip.so through the C-ABI) against the oracle in Philox mode,
bit-exact on FASTQ text (names, bases, qualities) and on mutations.txt / .vcf."""
import os
import pytest

from dwgsim_amd import api
from parity_common import CASES, compare_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return api.load()


@pytest.mark.parametrize("fasta,flags", CASES, ids=[f"{f}:{fl}" for f, fl in CASES])
def test_bit_exact_vs_oracle(lib, oracle_bin, golden_dir, fasta, flags):
    compare_case(lib, oracle_bin, os.path.join(golden_dir, fasta), flags)


def test_batches_and_shards_are_order_independent(lib, oracle_bin, golden_dir):
    """Read-index ranges are independent: tiny batches (many simulate() calls, rand_base chained by the
    host) give the same bytes as one call -- the property multi-GPU sharding relies on."""
    compare_case(lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 9 -N 3000 -y 0.2", batch_pairs=257)
