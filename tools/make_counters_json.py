#!/usr/bin/env python3
"""tools/make_counters_json.py <workload-key> <gpurun_out/tag> [profiles/r04_counters.json] [round tag, default r04] -- analysis only: fold the PMC passes of
tools/profile_round.sh into the JSON that bench.py quotes (roofline.traffic, roofline.valu).  Units and corrections as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as
64 bytes, so it is doubled; separate --pmc passes."""
import hashlib, json, os, re, subprocess, sys

key, d = sys.argv[1], sys.argv[2]
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_counters.json")
rtag = sys.argv[4] if len(sys.argv) > 4 else "r04"
vals, per_kernel = {}, {}
for line in open(os.path.join(d, "pmc.txt")):
    m = re.match(r"^(.*k_simulate.*?)\s([A-Z][A-Z0-9_]+)\s+([0-9.]+)\s+n=", line)
    if m:
        per_kernel.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
# a launch that ran as TWO kernels (the two-kernel form of small launches / short reads): bytes, instructions and busy time add up; both halves have the same waves
for name, kv in per_kernel.items():
    for k, v in kv.items():
        vals[k] = v if k == "SQ_WAVES" else vals.get(k, 0.0) + v
bench = json.load(open(os.path.join(d, "bench_line.json")))
pairs = bench["config"]["pairs_per_gpu_per_step"]
waves = vals.get("SQ_WAVES")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(open(os.path.join(root, "dwgsim_amd", "libdwgsim_hip.so"), "rb").read()).hexdigest()      # the build the counters were taken on: bench.py quotes them for this library only
try:
    head = open(os.path.join(d, "git_head.txt")).read().strip()
except Exception:
    head = None
pairs = int(pairs / max(bench["config"].get("launches_per_gpu_per_step", 1), 1))
out = {"source": f"profiles/{rtag}_{os.path.basename(d).replace('final_', '')}_kernel_stats_pmc.txt (rocprofv3 --kernel-trace --pmc, one counter group per pass, tools/profile_round.sh; MI355X)",
       "kernel": bench["roofline"]["kernel"] + (" (as two kernels: counters summed over both halves)" if len(per_kernel) > 1 else ""), "pairs_per_launch": pairs, "lib_sha256": sha, "git_head": head}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    out.update({"fetch_size_kib": vals["FETCH_SIZE"], "fetch_correction": 2.0, "write_size_kib": vals["WRITE_SIZE"],
                "traffic_bytes_per_launch": int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)})
if waves:
    out.update({"valu_instr_per_wave": round(vals["SQ_INSTS_VALU"] / waves), "salu_instr_per_wave": round(vals["SQ_INSTS_SALU"] / waves),
                "valu_instr_per_pair": round(vals["SQ_INSTS_VALU"] * 64 / pairs)})
    if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        # SQ_ACTIVE_INST_VALU counts quad-cycles per SE-level SQ; GRBM_GUI_ACTIVE is summed over the 8 XCDs (profiles/r01: same formula)
        out["valu_issue_active_pct"] = round(100.0 * vals["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * vals["GRBM_GUI_ACTIVE"] / 8), 1)
allj = json.load(open(dst)) if os.path.exists(dst) else {}
allj[key] = out
json.dump(allj, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out))
