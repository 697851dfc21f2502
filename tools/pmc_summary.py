#!/usr/bin/env python3
"""Summaries of a tools/profile_round.sh run for profiles/: HBM traffic of k_step from the FETCH_SIZE / WRITE_SIZE passes
(FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and per-kernel averages of the SQ pass.
usage: pmc_summary.py gpurun_out/<tag> profiles/<prefix>"""
import collections, csv, glob, json, os, re, sys
src, dst = sys.argv[1], sys.argv[2]
N = 65536


def rows(d):
    f = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def kname(n):
    m = re.search(r"(k_\w+)", n)
    return m.group(1) if m else None


tr = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for r in rows("pmc_" + c) if kname(r["Kernel_Name"]) == "k_step" and r["Counter_Name"] == c]
    v = v[len(v) // 5:]                      # skip warm-up launches
    tr[c] = sum(v) / max(1, len(v))
if tr.get("FETCH_SIZE") and tr.get("WRITE_SIZE"):
    fetch, write = tr["FETCH_SIZE"] * 1024 / N, tr["WRITE_SIZE"] * 1024 / N
    out = {str(N): {"FETCH_SIZE_KB_per_launch": tr["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": tr["WRITE_SIZE"],
                    "fetch_B_per_env_step_raw": fetch, "write_B_per_env_step": write,
                    "hbm_B_per_env_step_corrected": 2 * fetch + write},
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --workload env --graph-steps 1), "
                   "k_step launches after warm-up; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests "
                   "at 64 B); WRITE_SIZE as is; expected from the layout: reads 112 (tile) + 4 (episode no.) = 116 B, writes 104 "
                   "(tile) + 92 (obs) + 4 + 1 = 201 B (+ ~2 B of in-kernel resets)",
           "traffic_B_per_env_step": 2 * fetch + write}
    json.dump(out, open(dst + "_pmc_traffic.json", "w"), indent=1)
    print("traffic B/env-step", out["traffic_B_per_env_step"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows("pmc_sq"):
    k = kname(r["Kernel_Name"])
    if k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    out = {"counters_avg_per_launch": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in agg.items()}}
    sp = out["counters_avg_per_launch"].get("k_mlp_split")
    if sp and sp.get("SQ_VALU_MFMA_BUSY_CYCLES") and sp.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy cycles over all SIMDs (256 CUs x 4)
        sp["mfma_busy_fraction"] = sp["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * sp["GRBM_GUI_ACTIVE"] / 8)
    ks = out["counters_avg_per_launch"].get("k_step")
    if ks and ks.get("SQ_WAVE_CYCLES"):
        ks["wait_any_fraction"] = ks["SQ_WAIT_ANY"] / ks["SQ_WAVE_CYCLES"]
    out["note"] = ("rocprofv3 --pmc (one pass, 8 SQ/GRBM counters) on bench.py --serial --step-graph 0 (eager launches, serial order: "
                   "kernels do not overlap); averages over the launches of each kernel.  mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES "
                   "/ (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles")
    json.dump(out, open(dst + "_pmc_sq_counters.json", "w"), indent=1)
    print({k: round(v.get("mfma_busy_fraction", v.get("wait_any_fraction", 0)), 3) for k, v in out["counters_avg_per_launch"].items()})
