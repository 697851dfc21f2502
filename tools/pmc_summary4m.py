#!/usr/bin/env python3
"""Per-launch averages of every counter the round-3 passes of tools/r03_diag.sh collected for k_step at N = 4 M envs, plus
the kernel's average duration from the kernel trace.  usage: pmc_summary4m.py gpurun_out/<tag>"""
import collections, csv, glob, json, os, re, sys
src = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4194304
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc4m_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {c: sum(v[len(v) // 5:]) / max(1, len(v[len(v) // 5:])) for c, v in sorted(agg.items())}
dur = []
for f in glob.glob(os.path.join(src, "env4m", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if dur:
    d = dur[len(dur) // 5:]
    out["kernel_us_avg"] = sum(d) / len(d)
    out["algorithmic_GBps"] = 313.0 * N / (out["kernel_us_avg"] * 1e-6) / 1e9
    out["frac_of_8TBps"] = out["algorithmic_GBps"] / 8000.0
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_B_per_env_step_corrected"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024 / N
w = out.get("SQ_WAVE_CYCLES")
if w:
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
        if c in out:
            out[c + "_over_WAVE_CYCLES"] = out[c] / w
if "SQ_INSTS_VALU" in out and "SQ_WAVES" in out:
    out["valu_insts_per_wave"] = out["SQ_INSTS_VALU"] / out["SQ_WAVES"]
print(json.dumps(out, indent=1))
