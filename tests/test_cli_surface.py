"""The command-line surface of dwgsim-hip against the unmodified reference (oracle/_ref/dwgsim): the usage text line for line (behind the
program's own banner) and what both say -- and return -- on twenty-odd bad command lines (dwgsim_opt.c:93-160, :204-391, dwgsim.c:225).
Runs the CPU emulation build of the same CLI source (tests/emu/dwgsim-emu): no GPU needed for arguments."""
import os, subprocess
import pytest

from replay_common import have_reference, REF_BIN

HERE = os.path.dirname(os.path.abspath(__file__))
FA = os.path.join(HERE, "golden", "tiny.fa")
pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref/dwgsim is not here")

BAD = [[], ["-h"], ["-e", "abc", FA, "{P}"], ["-N", "x", FA, "{P}"], ["-c", "3", FA, "{P}"], ["-c", "2", FA, "{P}"], ["-1", "0", "-2", "0", FA, "{P}"], ["-y", "2", FA, "{P}"],
       ["-r", "-1", FA, "{P}"], ["-S", "3", FA, "{P}"], ["-A", "5", FA, "{P}"], ["-o", "4", FA, "{P}"], ["-o", "4", "-P", "x", FA, "{P}"], ["-q", "ab", FA, "{P}"], ["-q", "ab", "-Q", "-1", FA, "{P}"],
       ["-Q", "-1", FA, "{P}"], ["-c", "2", "-f", "TACG", "-e", "0.1-0.2", FA, "{P}"], ["-N", "10", "/nonexistent.fa", "{P}"], ["-N", "10", FA], ["-d", "-5", FA, "{P}"],
       ["-n", "-2", FA, "{P}"], ["-I", "0", FA, "{P}"], ["-X", "2", FA, "{P}"], ["-Z", FA, "{P}"], ["-C", "-1", "-N", "-1", FA, "{P}"], ["-m", "a", "-b", "b", FA, "{P}"],
       ["-x", "/nonexistent.bed", FA, "{P}"], ["-i", "-B", "-H", "-a", "-P", "pp", "-q", "I", "-f", "TACG", "-x", "r.bed", "-v", "m.vcf", "-z", "5", "-M", "2", "-h"], ["-E", "1.5", FA, "{P}"], ["-e", "2", FA, "{P}"]]


@pytest.fixture(scope="module")
def cli():
    subprocess.run([os.path.join(HERE, "emu", "build.sh")], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(HERE, "emu", "dwgsim-emu")


def said(binary, args, tmp):
    r = subprocess.run([binary] + [a.replace("{P}", os.path.join(tmp, "out")) for a in args], capture_output=True, text=True, timeout=120)
    lines = [l.replace(binary, "PROG") for l in r.stderr.splitlines() if not l.startswith(("Program:", "Version:", "Contact:", "Usage:"))]
    while lines and lines[0] == "": lines.pop(0)          # (the banner is three lines in the reference, two here)
    out = []
    for l in lines:                                       # blank lines around the banner collapse
        if l == "" and out and out[-1] == "": continue
        out.append(l)
    return r.returncode, out


@pytest.mark.parametrize("args", BAD, ids=[" ".join(a).replace(FA, "ref.fa") or "(none)" for a in BAD])
def test_bad_command_lines_are_answered_like_the_reference(cli, tmp_path, args):
    for d in ("r", "c"): os.makedirs(str(tmp_path / d), exist_ok=True)
    a = said(REF_BIN, args, str(tmp_path / "r"))
    b = said(cli, args, str(tmp_path / "c"))
    assert a[0] == b[0], (a[0], b[0], a[1][:5], b[1][:5])
    if a[0] == 0: return          # (a command line both accept -- "-e abc" is 0.0 to atof: the progress lines of a run are not compared here)
    assert a[1] == [l.replace(str(tmp_path / "c"), str(tmp_path / "r")) for l in b[1]], "\n".join(["reference:"] + a[1][:60] + ["dwgsim-hip:"] + b[1][:60])
