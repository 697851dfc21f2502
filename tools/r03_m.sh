#!/bin/bash
# one box: full GPU tests, then several updates per step with all draws in the opening launch (TT_MULTI_DRAW=1, default) or one per update (0)
set -o pipefail
out=gpurun_out/${1:-r03m}
mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -5 $out/tests.log
grep -q "tests rc=0" $out/tests.log || exit 1
for rep in 1 2; do
for md in 1 0; do
  TT_MULTI_DRAW=$md timeout -k 10 300 python3 bench.py --no-cpu-baseline --updates-per-step 64 --steps 200 --warmup 40 > $out/bench_u64_md${md}_$rep.json 2> $out/bench_u64_md$md.err
  TT_MULTI_DRAW=$md timeout -k 10 300 python3 bench.py --no-cpu-baseline --updates-per-step 8 --steps 400 --warmup 40 --n-envs 4096 > $out/bench_u8_4096_md${md}_$rep.json 2> $out/bench_u8_md$md.err
done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],5), "%.3e"%d["value"], d["timing"]["median_ms_per_step"], round(d["ms_per_step"]*1e3/d["config"]["updates_per_step"],2), "us per update")
    except Exception as e: print(f, "failed", e)
PY
