#!/bin/bash
# tools/r05_exit_probe.sh -- analysis only (gpurun): tools/ubench_exit.hip for a few shapes of held memory
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
hipcc -O2 --offload-arch=gfx950 -o /tmp/ubench_exit tools/ubench_exit.hip || exit 1
head -c 3000000000 /dev/urandom > /dev/shm/big.bin
run() { echo "== $*"; python3 - "$@" <<'PY'
import subprocess, sys, time
r = subprocess.run(["/tmp/ubench_exit"] + sys.argv[1:], stdout=subprocess.PIPE); t1 = time.time()
out = r.stdout.decode(); print(out.strip())
e = [float(l.split()[1]) for l in out.splitlines() if l.startswith("EXIT")]
if e: print(f"_exit -> parent's wait returns: {t1 - e[0]:.3f} s")
PY
}
run 0 0 0
run 3 300 0
run 11 100 0
run 0 0 10000
run 0 0 0 /dev/shm/big.bin
run 11 150 10000 /dev/shm/big.bin
rm -f /dev/shm/big.bin
