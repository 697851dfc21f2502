#!/bin/bash
# one box: updates-per-step scaling (N = 65536 and 4096), soak of the final loop, driver forms
set -o pipefail
out=gpurun_out/${1:-r03l}
mkdir -p $out
timeout -k 10 600 python3 tools/updates_scaling.py 65536 > $out/updates_scaling_65536.txt 2>&1; cat $out/updates_scaling_65536.txt
timeout -k 10 600 python3 tools/updates_scaling.py 4096 > $out/updates_scaling_4096.txt 2>&1; cat $out/updates_scaling_4096.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver2.json 2> $out/bench_driver2.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],5), "%.3e"%d["value"], d["timing"]["median_ms_per_step"], d["config"].get("setup_vector_steps"))
    except Exception as e: print(f, "failed", e)
PY
timeout -k 10 600 python3 tools/soak.py 65536 200000 > $out/soak.txt 2>&1; tail -8 $out/soak.txt
