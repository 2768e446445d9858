cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06b8
for mp in 8388608 4194304 2097152; do
  DWGSIM_BENCH_MAX_LAUNCH_PAIRS=$mp timeout 600 python bench.py --solo-sweep 8 --no-cpu-baseline --no-legs --steps 20 > gpurun_out/r06b8/sweep_$mp.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/r06b8/sweep_$mp.json"))
for mode in ("weak","strong"):
    for W,v in d[mode].items():
        if W=="job": continue
        print("$mp", mode, "W", W, "max", v["max_ms_per_step"], "min", v["min_ms_per_step"], "eff", v["efficiency"], "launches", v["ranks"][0]["launches"])
PY
done
