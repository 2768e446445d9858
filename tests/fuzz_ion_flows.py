"""tests/fuzz_ion_flows.py <seed> <count> -- test helper: Ion Torrent option sets with RANDOM FLOW ORDERS (4 .. 64 flows, long gaps included) through
the HIP path (or, with DWGSIM_HIP_LIB=tests/emu/libdwgsim_emu.so, the emulated kernels) against the oracle, byte for byte.  A read that outgrows its
buffer (documented limit: the flow model is a branching process) is counted separately, not as a mismatch."""
import os, sys, random, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seed, count = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed); bad = rej = outgrew = 0
for k in range(count):
    F = rng.choice([4, 4, 5, 8, 13, 32, 32, 40, 64])
    while True:
        fl = "".join(rng.choice("ACGT") for _ in range(F))
        if set(fl) == set("ACGT"): break
    if rng.random() < 0.15: fl = rng.choice(["TACG", "TACGTACGTCTGAGCATCGATCGATGTACAGC", "TCG" + "A" * 37, "TACG" * 4 + "A" * 34])
    l1 = rng.choice([1, 2, 5, 8, 9, 17, 40, 100, 150, 251, 400]); pe = rng.random() < 0.3
    l2 = rng.choice([1, 8, 50, 120]) if pe else 0
    e = rng.choice(["0", "0.0001", "0.005", "0.01", "0.02", "0.05", "0.1", "0.2"])
    f = [f"-z {rng.randrange(1, 10000)}", "-c 2", f"-f {fl}", f"-1 {l1}", f"-2 {l2}", f"-e {e}"]
    if pe: f += [f"-E {rng.choice(['0', '0.01', '0.1'])}", f"-d {max(l1 + l2, rng.choice([200, 500]))}", f"-s {rng.choice([0, 10, 50])}"]
    f.append(rng.choice([f"-N {rng.choice([1, 63, 64, 65, 500, 2000])}", f"-C {rng.choice([0.5, 3])}"]))
    if rng.random() < 0.5: f.append(f"-r {rng.choice([0, 0.001, 0.05])}")
    if rng.random() < 0.3: f.append(f"-R {rng.choice([0.1, 0.9])}")
    if rng.random() < 0.4: f.append(f"-y {rng.choice([0, 0.1, 1.0])}")
    if rng.random() < 0.4: f.append(f"-n {rng.choice([0, 3, 1000])}")
    if rng.random() < 0.3: f.append(f"-A {rng.choice([1, 2])}")
    if rng.random() < 0.3: f.append(f"-o {rng.choice([0, 1, 2])}")
    if rng.random() < 0.2: f.append(f"-Q {rng.choice([0, 10])}")
    if rng.random() < 0.1 and not os.environ.get("DWGSIM_FUZZ_NO_B"): f.append("-B")      # (the draw is made either way)
    flags = " ".join(f); fasta = os.path.join(ROOT, "tests", "golden", rng.choice(["tiny.fa", "odd.fa", "ex1.fa"]))
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_flags.py"), "--one", flags, fasta], capture_output=True, text=True, timeout=150)
        rc, out = r.returncode, (r.stdout + r.stderr[-300:]).strip()
    except subprocess.TimeoutExpired:
        rc, out = 5, "TIMEOUT"
    if rc == 3: rej += 1
    if rc == 4 and ("outgrew its buffer" in out or "dwgsim_hip_create failed with code -5" in out): outgrew += 1; continue      # (-B: the calibration runs the same model)
    if rc not in (0, 3): bad += 1; print(f"[{k}] rc={rc} {os.path.basename(fasta)} {flags}\n   {out[-300:]}", flush=True)
print(f"ion flow fuzz seed {seed}: {count} cases, {rej} rejected by the oracle, {outgrew} outgrew their buffer, {bad} bad", flush=True)
