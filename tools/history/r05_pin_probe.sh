#!/bin/bash
# tools/r05_pin_probe.sh -- analysis only (gpurun): tools/ubench_pin.hip
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
hipcc -O2 --offload-arch=gfx950 -o /tmp/ubench_pin tools/ubench_pin.hip -lpthread 2>/dev/null || exit 1
run() { echo "== $*"; python3 - "$@" <<'PY'
import subprocess, sys, time
r = subprocess.run(["/tmp/ubench_pin"] + sys.argv[1:], stdout=subprocess.PIPE); t1 = time.time()
out = r.stdout.decode(); print(out.strip())
e = [float(l.split()[1]) for l in out.splitlines() if l.startswith("EXIT")]
if e: print(f"_exit -> parent's wait returns: {t1 - e[0]:.3f} s")
PY
}
run malloc 100 8 1
run malloc 100 8 4
run register 100 8 1
run register_thp 100 8 1
run register_thp 100 8 4
run register_thp 300 3 1
